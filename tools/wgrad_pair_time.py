"""A/B of the 3x3 weight-gradient kernel: pair form (Winograd F(3,2), default) against the direct form (TE_WGRAD_DIRECT=1,
read once per process), at the FFHQ-256 batch-16 layer shapes.  Prints time, algorithmic TFLOP/s and the deviation of the
reduced dW from an fp64 torch reference at the small shapes.      python tools/wgrad_pair_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402
from tools.exp_time import timeit     # noqa: E402

DEV = 'cuda'
SHAPES = [(128, 128, 256, 16), (256, 256, 128, 16), (512, 512, 64, 16), (512, 512, 32, 16), (512, 512, 16, 16), (512, 512, 8, 16),
          (128, 128, 256, 32), (513, 512, 4, 16), (96, 160, 40, 3)]

if __name__ == '__main__':
    label = 'direct' if os.environ.get('TE_WGRAD_DIRECT') else 'pair'
    for K, M, H, B in SHAPES:
        torch.manual_seed(1)
        g = torch.randn(B, M, H, H, device=DEV)
        x = torch.randn(B, K, H, H, device=DEV)
        fn = lambda: _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, H)
        out = fn()
        ms = timeit(fn)
        flops = 2.0 * 9 * K * M * H * H * B
        dw = out.double().sum(dim=tuple(range(out.dim() - 3)))          # slabs [..., M, K, 9] -> [M, K, 9]
        err = ''
        if H <= 64:
            ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (M, K, 3, 3), g.double().cpu(), padding=1).reshape(M, K, 9)
            err = f'  rel err vs fp64 {float((dw.cpu() - ref).norm() / ref.norm()):.2e}'
        print(f'[{label}] W3X3 {K:4d}->{M:4d} @{H:4d} B{B:2d}: {ms:8.3f} ms {flops / ms / 1e9:7.1f} TF/s  checksum {float(dw.sum()):.8e}{err}', flush=True)
