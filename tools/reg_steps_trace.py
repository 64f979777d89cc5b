"""run only the lazy-regulariser sub-steps (path length x4, R1 x2) of the FFHQ-256 train step: for a kernel trace of the
twice-differentiated paths (rocprofv3 --kernel-trace --stats -- python tools/reg_steps_trace.py [path|r1])"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transeditor_amd.train_step import TrainStep, default_args      # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'path'
if 'torchscale' in sys.argv:          # A/B: the framework's broadcast multiply in the any-order composite
    from transeditor_amd.op import chanscale
    chanscale.USE_KERNELS = False
if 'oldcomposite' in sys.argv:        # A/B: chan_scale -> conv trio -> chan_scale instead of the closed five-linear family (round 4)
    from transeditor_amd.op import modconv
    modconv.USE_CLOSED_MODCONV = False
dev = 'cuda'
ts = TrainStep(default_args(size=256, batch=16), dev)
real = torch.randn(16, 3, 256, 256, device=dev).clamp(-1, 1)
from transeditor_amd.op.modconv import packed_weights_cache      # noqa: E402
step = ts.path_step if which == 'path' else (lambda: ts.r1_step(real))


def fn():                                  # as inside TrainStep.iteration: packed weight layouts cached between optimiser steps
    with packed_weights_cache(ts._packs):
        step()


fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(4):
    fn()
e.record()
torch.cuda.synchronize()
print(f'{which}: {s.elapsed_time(e) / 4:.1f} ms per step')
