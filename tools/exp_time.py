"""Time conv-family kernels of a library variant at the FFHQ-256 batch-16 layer shapes (HIP events) and check them
against the product library.   python tools/exp_time.py [variant ...]      (variants built by tools/exp_build.py)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402

DEV = 'cuda'
SHAPES = [  # kind, K, M, H (low-res), label
    ('T2', 512, 256, 64), ('T2', 256, 128, 128), ('T2', 512, 512, 32), ('T2', 512, 512, 16), ('T2', 512, 512, 8), ('T2', 512, 512, 4), ('3X3', 512, 512, 8), ('3X3', 512, 512, 4), ('S2', 512, 512, 8), ('S2', 512, 512, 4), ('S2', 256, 512, 64), ('S2', 128, 256, 128),
    ('1X1', 256, 512, 64), ('1X1', 128, 256, 128), ('1X1', 512, 512, 32), ('1X1', 512, 512, 16), ('1X1', 512, 256, 64), ('1X1', 256, 128, 128), ('1X1', 512, 512, 8), ('3X3', 256, 256, 128), ('3X3', 128, 128, 256), ('3X3', 512, 512, 64), ('3X3', 512, 512, 16), ('R3X3', 128, 128, 256), ('R3X3', 256, 256, 128), ('R1X1', 128, 256, 128), ('WT2', 512, 256, 64), ('WT2', 256, 128, 128), ('W3X3', 128, 128, 256)]


if os.environ.get('SMALL'):      # the layers <= 16^2 only (split-K plan experiments)
    SHAPES = [s for s in SHAPES if s[3] <= 16 and s[0] in ('T2', 'S2', '3X3', '1X1')] + [
        ('3X3', 512, 512, 16), ('S2', 512, 512, 16), ('W3X3', 512, 512, 16), ('W3X3', 512, 512, 8), ('W3X3', 512, 512, 4),
        ('WT2', 512, 512, 16), ('WT2', 512, 512, 8), ('WT2', 512, 512, 4)]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def run(label):
    B = 16
    res = {}
    for kind, K, M, H in SHAPES:
        flops = 2.0 * (1 if kind in ('1X1', 'R1X1') else 9) * K * M * H * H * B
        torch.manual_seed(0)
        if kind in ('R3X3', 'R1X1'):
            pass
        elif kind in ('T2', '3X3', '1X1'):
            x = torch.randn(B, K, H, H, device=DEV)
        elif kind == 'S2':
            x = torch.randn(B, K, 2 * H + 1, 2 * H + 1, device=DEV)
        if kind in ('R3X3', 'R1X1'):       # unmodulated launch with the residual (+ mask) epilogue stages (discriminator ResBlock node)
            ks = 1 if kind == 'R1X1' else 3
            x = torch.randn(B, K, H, H, device=DEV)
            w = torch.randn(M, K, ks, ks, device=DEV) / (ks * K ** 0.5)
            wp = _lib.conv_pack(w, _lib.PACK_FWD, 1.0)
            resid = torch.randn(B, M, H, H, device=DEV)
            mref = torch.randn(B, M, H, H, device=DEV) if ks == 3 else None
            code = _lib.CONV_3X3 if ks == 3 else _lib.CONV_1X1
            fn = lambda: _lib.conv(x, wp, code, M, H, H, None, None, None, 0, res=resid, mask_ref=mref, mask_gain=2 ** 0.5)
        elif kind in ('T2', 'S2', '3X3', '1X1'):
            ks = 1 if kind == '1X1' else 3
            w = torch.randn(M, K, ks, ks, device=DEV) / (ks * K ** 0.5)
            wp = _lib.conv_pack(w, _lib.PACK_FWD, 1.0)
            isc = None if os.environ.get('NO_ISC') else 1 + 0.1 * torch.randn(B, K, device=DEV)
            code = {'T2': _lib.CONV_T2, 'S2': _lib.CONV_S2, '3X3': _lib.CONV_3X3, '1X1': _lib.CONV_1X1}[kind]
            fn = lambda: _lib.conv(x, wp, code, M, H, H, isc, None, None, 0)
        elif kind in ('WT2', 'W3X3'):
            code = _lib.CONV_T2 if kind == 'WT2' else _lib.CONV_3X3
            g = torch.randn(B, M, 2 * H + 1, 2 * H + 1, device=DEV) if kind == 'WT2' else torch.randn(B, M, H, H, device=DEV)
            x = torch.randn(B, K, H, H, device=DEV)
            fn = lambda: _lib.wgrad_slabs(g, x, code, H, H)
        out = fn()
        ms = timeit(fn)
        res[(kind, K, M, H)] = (ms, flops / ms / 1e9, out.double().sum().item(), out.abs().double().sum().item())
        print(f'[{label}] {kind:5s} {K:4d}->{M:4d} @{H:4d}: {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TF/s   checksum {res[(kind, K, M, H)][2]:.6e} / {res[(kind, K, M, H)][3]:.6e}',
              flush=True)
        del out
    return res


if __name__ == '__main__' and 'fir' not in sys.argv:
    name = sys.argv[1] if len(sys.argv) > 1 else None
    if name and name != 'product':
        _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', f'libte_{name}.so')
    run(name or 'product')


def fir_bench(label):
    """HBM-bound tail at the FFHQ-256 batch-16 top shapes: blur (+bias+lrelu) 257^2 -> 256^2, adjoint blur 256^2 -> 257^2,
    backward of blur+bias+lrelu in one pass (AG), bias_act backward; TB/s of ALGORITHMIC bytes (each operand once)."""
    from oracle import te_oracle as O       # taps only
    B, C, H = 16, 128, 256
    k = O.fir_kernel((1, 3, 3, 1), 4.0).to(DEV)
    kf = torch.flip(k, [0, 1]).contiguous()
    x = torch.randn(B, C, 2 * (H // 2) + 1, H + 1, device=DEV)
    bias = torch.randn(C, device=DEV)
    f1 = lambda: _lib.upfirdn2d_raw(x, k, (1, 1), (1, 1), (1, 1, 1, 1), bias=bias, act=3, alpha=0.2, scale=2 ** 0.5)
    y = f1()
    g = torch.randn_like(y)
    f2 = lambda: _lib.upfirdn2d_raw(g, kf, (1, 1), (1, 1), (2, 2, 2, 2))
    f3 = lambda: _lib.blur_actgrad(g, y, kf, (2, 2, 2, 2), 0.2, 2 ** 0.5)
    f4 = lambda: _lib.bias_act_bwd(g, y, 0.2, 2 ** 0.5, want_bias=True)
    # the same adjoint blur into rows of 260 floats (16-byte aligned rows): does the 257-wide output cost its row alignment?
    f2a = lambda: _lib.upfirdn2d_raw(g, kf, (1, 1), (1, 1), (2, 5, 2, 2))
    # which property of the 256 -> 257 direction costs 35 %?  255 -> 256 (aligned output rows, no extra row / column) and
    # 252 -> 253 (unaligned output rows, no extra row / column), same taps and pads
    g255, g252 = torch.randn(B, C, 255, 255, device=DEV), torch.randn(B, C, 252, 252, device=DEV)
    f2b = lambda: _lib.upfirdn2d_raw(g255, kf, (1, 1), (1, 1), (2, 2, 2, 2))
    f2c = lambda: _lib.upfirdn2d_raw(g252, kf, (1, 1), (1, 1), (2, 2, 2, 2))
    xs = torch.randn(B, C, H, H, device=DEV)
    f5 = lambda: _lib.upfirdn2d_raw(xs, k, (1, 1), (2, 2), (2, 2, 2, 2))       # D skip branch: blur + keep every 2nd pixel
    for name, fn, nbytes in (('blur+bias+lrelu 257->256', f1, 4 * (x.numel() + y.numel())),
                             ('adjoint blur 256->257', f2, 4 * (g.numel() + x.numel())),
                             ('adjoint blur 256->257x260 (aligned rows)', f2a, 4 * (g.numel() + B * C * 257 * 260)),
                             ('adjoint blur 255->256 (aligned out, no EXT)', f2b, 4 * B * C * (255 * 255 + 256 * 256)),
                             ('adjoint blur 252->253 (unaligned out, no EXT)', f2c, 4 * B * C * (252 * 252 + 253 * 253)),
                             ('blur_actgrad (AG) 256->257', f3, 4 * (2 * g.numel() + x.numel())),
                             ('bias_act_bwd', f4, 4 * 3 * g.numel()),
                             ('blur-down2 256->128', f5, 4 * (xs.numel() + xs.numel() // 4))):
        ms = timeit(fn, 20)
        print(f'[{label}] FIR {name:46s}: {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.2f} TB/s algorithmic', flush=True)


if __name__ == '__main__' and 'fir' in sys.argv:
    name = sys.argv[1] if sys.argv[1] != 'fir' else 'product'
    if name != 'product':
        _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', f'libte_{name}.so')
    fir_bench(name)
