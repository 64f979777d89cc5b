"""time te_rgb_wgrad_f32 at the ToRGB shapes of FFHQ-1024 (batch 4) / FFHQ-256 (batch 16) and the stem shape of the discriminator"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transeditor_amd import _lib


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for B, K, H in ((4, 32, 1024), (4, 64, 512), (16, 128, 256), (16, 256, 128), (16, 512, 64), (32, 128, 256)):
    x = torch.randn(B, K, H, H, device='cuda'); g = torch.randn(B, 3, H, H, device='cuda')
    ref = torch.einsum('bop,bkp->bok', g.flatten(2).double(), x.flatten(2).double())
    got = _lib.rgb_wgrad_slabs(g, x).sum(1).squeeze(-1).double()
    err = float((got - ref).abs().max() / ref.abs().max())
    ms = t(lambda: _lib.rgb_wgrad_slabs(g, x))
    print(f'rgb_wgrad B{B} K{K} @{H}: {ms * 1e3:7.1f} us  {(x.numel() + g.numel()) * 4 / ms / 1e9:5.2f} TB/s  err {err:.1e}')
