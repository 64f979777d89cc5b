"""Differential fuzz of the GROUPED plain weight gradient (te_wgrad_group_f32) against the per-sample form (te_wgrad_f32) on
random shapes where the plan may group: channel counts that are multiples of 128, batches with many divisors, images from 1 pixel
to 40 (scalar and 16-byte staging paths, partial cell tiles, the transposed kind's odd-sized operand).
    python tools/wgrad_group_fuzz.py [cases] [seed]"""
import sys

import torch

sys.path.insert(0, '.')
from transeditor_amd import _lib

DEV, TOL = 'cuda', 2e-5
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
pick = lambda xs: xs[int(torch.randint(len(xs), (1,), generator=g))]
bad, grouped, worst = 0, 0, (0.0, '')
for i in range(n):
    kind = ('3X3', '1X1', 'T2')[i % 3]
    code = {'3X3': _lib.CONV_3X3, '1X1': _lib.CONV_1X1, 'T2': _lib.CONV_T2}[kind]
    B, Co, Ci = pick([8, 12, 16, 24, 32, 48, 64]), pick([128, 256, 384, 512]), pick([128, 256, 512])
    H, W = pick([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 32]), pick([1, 2, 4, 5, 8, 11, 16, 31, 32, 33, 40])
    x = torch.randn(B, Ci, H, W, device=DEV)
    gy = torch.randn(B, Co, 2 * H + 1, 2 * W + 1, device=DEV) if kind == 'T2' else torch.randn(B, Co, H, W, device=DEV)
    per = _lib.wgrad_slabs(gy, x, code, H, W)
    grp = _lib.wgrad_slabs(gy, x, code, H, W, group=True)
    a, b = grp.sum(dim=(0, 1)), per.sum(dim=(0, 1))
    err = float((a - b).norm() / b.norm())
    grouped += grp.shape[0] < B
    desc = f'{kind} B={B} Co={Co} Ci={Ci} H={H} W={W} slabs {tuple(per.shape[:2])} -> {tuple(grp.shape[:2])}'
    if err > worst[0]:
        worst = (err, desc)
    if err > TOL:
        bad += 1
        print(f'[{i}] MISMATCH {err:.2e}: {desc}', flush=True)
print(f'wgrad_group_fuzz: {n} cases ({grouped} grouped), {bad} above {TOL:g}; worst {worst[0]:.2e} ({worst[1]})')
sys.exit(1 if bad else 0)
