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
    ('T2', 512, 256, 64), ('T2', 256, 128, 128), ('T2', 512, 512, 32), ('S2', 256, 512, 64), ('S2', 128, 256, 128),
    ('3X3', 256, 256, 128), ('WT2', 512, 256, 64), ('WT2', 256, 128, 128), ('W3X3', 128, 128, 256)]


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
        flops = 2.0 * 9 * K * M * H * H * B
        torch.manual_seed(0)
        if kind in ('T2', '3X3'):
            x = torch.randn(B, K, H, H, device=DEV)
        elif kind == 'S2':
            x = torch.randn(B, K, 2 * H + 1, 2 * H + 1, device=DEV)
        if kind in ('T2', 'S2', '3X3'):
            w = torch.randn(M, K, 3, 3, device=DEV) / (3 * K ** 0.5)
            wp = _lib.conv_pack(w, _lib.PACK_FWD, 1.0)
            isc = 1 + 0.1 * torch.randn(B, K, device=DEV)
            code = {'T2': _lib.CONV_T2, 'S2': _lib.CONV_S2, '3X3': _lib.CONV_3X3}[kind]
            fn = lambda: _lib.conv(x, wp, code, M, H, H, isc, None, None, 0)
        else:
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


if __name__ == '__main__':
    name = sys.argv[1] if len(sys.argv) > 1 else None
    if name and name != 'product':
        _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', f'libte_{name}.so')
    run(name or 'product')
