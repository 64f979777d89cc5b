"""Time of TE_CONV_3X3W6 with and without the residual + mask epilogue stages at the discriminator's conv1 shapes (joint batch of 32).
    python tools/wino6_epilogue_time.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transeditor_amd import _lib
from tools.exp_time import timeit
DEV='cuda'
for (B,K,M,H) in [(32,128,128,256),(32,256,256,128),(16,128,128,256)]:
    x=torch.randn(B,K,H,H,device=DEV); w=torch.randn(M,K,3,3,device=DEV)/(3*math.sqrt(K))
    res=torch.randn(B,M,H,H,device=DEV); mref=torch.randn(B,M,H,H,device=DEV); bias=torch.randn(M,device=DEV)
    u6=_lib.conv_pack(w,_lib.PACK_W6FWD)
    f=lambda: _lib.conv(x,u6,_lib.CONV_3X3W6,M,H,H,None,None,bias,3,res=res,mask_ref=mref,mask_gain=1.3)
    g=lambda: _lib.conv(x,u6,_lib.CONV_3X3W6,M,H,H,None,None,bias,3)
    fl=2.0*9*K*M*H*H*B
    tf,tg=timeit(f),timeit(g)
    print(f'B{B} {K}->{M} @{H}: res+mask {tf*1e3:8.1f} us {fl/tf/1e9:6.1f} TF/s | plain {tg*1e3:8.1f} us {fl/tg/1e9:6.1f}')
