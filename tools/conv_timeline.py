"""Whole-block timeline of a convolution launch (experimental build with -DTE_CONV_PROF2):
    python tools/exp_build.py prof2 -DTE_CONV_PROF2 && python tools/conv_timeline.py
Every block's wave 0 stamps s_memrealtime (100 MHz) at kernel entry, K-loop start, K-loop end and exit: where the time of a
SMALL launch goes (dispatch ramp, prologue, stage loop, epilogue), which the per-stage profiler of the FAST kernels cannot say."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', 'libte_prof2.so')
DEV = 'cuda'
CODE = {'T2': _lib.CONV_T2, 'S2': _lib.CONV_S2, '3X3': _lib.CONV_3X3, '1X1': _lib.CONV_1X1}


def run(kind, K, M, H, B=16, isc=False):
    torch.manual_seed(0)
    x = torch.randn(B, K, 2 * H + 1, 2 * H + 1, device=DEV) if kind == 'S2' else torch.randn(B, K, H, H, device=DEV)
    ks = 1 if kind == '1X1' else 3
    w = torch.randn(M, K, ks, ks, device=DEV) / (ks * K ** 0.5)
    wp = _lib.conv_pack(w, _lib.PACK_FWD, 1.0)
    sc = (1 + 0.1 * torch.randn(B, K, device=DEV)) if isc else None
    for _ in range(5):
        _lib.conv(x, wp, CODE[kind], M, H, H, sc, None, None, 0)
    torch.cuda.synchronize()
    L = _lib.lib()
    L.te_debug_conv_prof.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    zero = np.zeros(8192 * 8, dtype=np.uint64)
    L.te_debug_conv_prof_clear()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _lib.conv(x, wp, CODE[kind], M, H, H, sc, None, None, 0)
    e1.record()
    torch.cuda.synchronize()
    wall = e0.elapsed_time(e1) / 20 * 1e3
    buf = zero.copy()
    assert L.te_debug_conv_prof(buf.ctypes.data, buf.nbytes) == 0
    raw = buf.reshape(-1, 8)
    raw = raw[raw[:, 7] == 1]
    d = raw[:, :5].astype(np.float64)
    lo, hi = (lambda v: (v & np.uint64(0xFFFFFFFF)).astype(np.float64) / 100.0), (lambda v: (v >> np.uint64(32)).astype(np.float64) / 100.0)
    sub = (lo(raw[:, 5]), hi(raw[:, 5]), lo(raw[:, 6]), hi(raw[:, 6]))      # sync+commit, of it vmcnt wait, issue, compute  (generic form only)
    t0 = d[:, 0].min()
    us = (d[:, :4] - t0) / 100.0           # 100 MHz ticks -> us
    pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    f = lambda v: f'mean {v.mean():6.1f}  p10 {np.percentile(v, 10):6.1f}  p90 {np.percentile(v, 90):6.1f}  max {v.max():6.1f}'
    print(f'{kind} {K}->{M} @{H} B={B} isc={isc}: {wall:.1f} us per call by events (main kernel + finalize); {len(d)} blocks, '
          f'{d[0, 4]:.0f} stages each')
    print(f'   block entry after the first block: {f(us[:, 0])}')
    print(f'   prologue (entry -> K loop)       : {f(pro)}')
    print(f'   K loop                           : {f(loop)}   ({(loop / d[:, 4]).mean():.2f} us per stage)')
    if sub[3].max() > 0:
        print(f'      generic form, per block: barriers + commit {sub[0].mean():.1f} us (of it waiting for the stage loads {sub[1].mean():.1f}), '
              f'issue {sub[2].mean():.1f}, MFMA phase {sub[3].mean():.1f}')
    print(f'   epilogue                         : {f(epi)}')
    print(f'   first entry -> last exit         : {us[:, 3].max():.1f} us')


if __name__ == '__main__':
    for kind, K, M, H in [('3X3', 512, 512, 4), ('3X3', 512, 512, 8), ('T2', 512, 512, 4), ('T2', 512, 512, 8),
                          ('S2', 512, 512, 4), ('1X1', 512, 512, 8), ('3X3', 512, 512, 16), ('T2', 512, 512, 16)]:
        run(kind, K, M, H)
    run('3X3', 512, 512, 4, isc=True)
    run('3X3', 512, 512, 4, B=32)
