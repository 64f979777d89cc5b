"""Time of the plain weight gradient (slabs + their reduction) per-sample vs grouped at the discriminator's batch-32 shapes.
    python tools/wgrad_group_time.py"""
import sys

import torch

sys.path.insert(0, '.')
from transeditor_amd import _lib

DEV = 'cuda'


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for kind, code in (('3X3', _lib.CONV_3X3), ('T2', _lib.CONV_T2), ('1X1', _lib.CONV_1X1)):
    for B, C, H in [(32, 512, 4), (32, 512, 8), (32, 512, 16), (32, 512, 32), (32, 512, 64), (16, 512, 16)]:
        x = torch.randn(B, C, H, H, device=DEV)
        g = torch.randn(B, C, 2 * H + 1, 2 * H + 1, device=DEV) if kind == 'T2' else torch.randn(B, C, H, H, device=DEV)

        def run(group):
            sl = _lib.wgrad_slabs(g, x, code, H, H, group)
            if kind == 'T2':
                return sl.sum(dim=(0, 1))
            return _lib.wgrad_reduce(sl, sl[0, 0], 1.0, None, None, want_w=True)[0]
        a, b = timeit(lambda: run(False)), timeit(lambda: run(True))
        shp = tuple(_lib.wgrad_slabs(g, x, code, H, H, True).shape[:2])
        print(f'{kind} B={B} {C}x{C} @{H}: per-sample {a:7.1f} us   grouped {b:7.1f} us   (slabs {B} -> {shp[0]} x {shp[1]})', flush=True)
