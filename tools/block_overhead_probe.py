"""Per-block overhead of the split-bf16 kernels: the same spatial problem with 128 and 512 input channels (8 and 32 stages per block);
time = blocks/CU x (stages x t_stage + overhead).   python tools/block_overhead_probe.py"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402
from tools.exp_time import timeit     # noqa: E402

DEV = 'cuda'


def run(name, kind, pack, B, M, H, W, hin):
    t = {}
    for K in (128, 512):
        x = torch.randn(B, K, *hin(H, W), device=DEV)
        w = torch.randn(M, K, 3, 3, device=DEV) / (3 * math.sqrt(K))
        wp = _lib.conv_pack(w, pack)
        f = lambda: _lib.conv(x, wp, kind, M, H, W)
        t[K] = min(timeit(f, n=20), timeit(f, n=20)) * 1e3      # us
    per_stage = (t[512] - t[128]) / 24.0          # per 1 of the 8 / 32 stages, all blocks of the launch
    over = t[128] - 8 * per_stage
    print(f'{name}: K=128 {t[128]:8.1f} us, K=512 {t[512]:8.1f} us -> stages {8 * per_stage:8.1f} us of the K=128 launch, '
          f'outside the stages {over:7.1f} us = {100 * over / t[128]:4.1f} % (K=128), {100 * over / t[512]:4.1f} % (K=512)', flush=True)


if __name__ == '__main__':
    run('wino6p 3x3  M=128 @256x256 b16', _lib.CONV_3X3W6, _lib.PACK_W6FWD, 16, 128, 256, 256, lambda h, w: (h, w))
    run('wino6p 3x3  M=256 @128x128 b16', _lib.CONV_3X3W6, _lib.PACK_W6FWD, 16, 256, 128, 128, lambda h, w: (h, w))
    run('s2s6   S2   M=256 @128x128 b16', _lib.CONV_S2S6, _lib.PACK_S6FWD, 16, 256, 128, 128, lambda h, w: (2 * h + 1, 2 * w + 1))
    run('t2s6   T2   M=128 @128x128 b16', _lib.CONV_T2S6, _lib.PACK_T6FWD, 16, 128, 128, 128, lambda h, w: (h, w))
