"""Per-phase cycle counts inside the ping-pong split-bf16 Winograd kernel (experimental build with -DW6P_PROF):
    python tools/exp_build.py w6p_prof -DW6P_PROF && python tools/w6p_phase_prof.py [variant]
Wave-level s_memtime stamps: multiplying role (first half / mid barrier / second half), staging role (part 1 / mid barrier / part 2),
end-of-phase barrier."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', f'libte_{sys.argv[1] if len(sys.argv) > 1 else "w6p_prof"}.so')
DEV = 'cuda'


def run(B, K, M, H, W):
    torch.manual_seed(0)
    x = torch.randn(B, K, H, W, device=DEV)
    w = torch.randn(M, K, 3, 3, device=DEV) / (3 * K ** 0.5)
    u6 = _lib.conv_pack(w, _lib.PACK_W6FWD, 1.0)
    isc = 1 + 0.1 * torch.randn(B, K, device=DEV)
    _lib.wino6_form(1)
    for _ in range(3):
        _lib.conv(x, u6, _lib.CONV_3X3W6, M, H, W, isc, None, None, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.conv(x, u6, _lib.CONV_3X3W6, M, H, W, isc, None, None, 0)
    e1.record()
    torch.cuda.synchronize()
    wall_us = e0.elapsed_time(e1) * 1e3
    buf = np.zeros(2048 * 8 * 8, dtype=np.uint64)
    L = _lib.lib()
    L.te_debug_w6p_prof.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    assert L.te_debug_w6p_prof(buf.ctypes.data, buf.nbytes) == 0
    raw = buf.reshape(-1, 8, 8)
    nst = (raw[:, :, 7] >> np.uint64(48)).astype(np.float64)
    tot = (raw[:, :, 7] & np.uint64(0xFFFFFFFFFFFF)).astype(np.float64)
    real = (raw[:, :, 6] >> np.uint64(40)).astype(np.float64)
    raw[:, :, 6] &= np.uint64((1 << 40) - 1)
    d = raw.astype(np.float64)
    keep = nst[:, 0] > 0
    d, nst, tot, real = d[keep], nst[keep], tot[keep], real[keep]
    print(f'   shader clock: {(tot / (real / 100e6)).mean() / 1e9:.3f} GHz (cycle counter per second of the 100 MHz real-time counter)')
    print(f'B{B} {K}->{M} @{H}x{W}: {wall_us:.0f} us, {len(d)} blocks recorded, {nst[0, 0]:.0f} stages per block; '
          f'loop span per block (cycle counter) mean {tot.mean():.0f} = {tot.mean() / nst[0, 0]:.0f} per stage')
    names = ('mult first half', 'mult mid barrier', 'mult second half', 'stage part 1', 'stage mid barrier', 'stage part 2', 'end barriers (2/stage)')
    for g in (0, 1):
        sel = d[:, 4 * g:4 * g + 4, :]
        st = nst[:, 4 * g:4 * g + 4]
        print(f'  group {g}:')
        for i, n in enumerate(names):
            per = sel[:, :, i] / st
            print(f'     {n:24s} mean {per.mean():7.0f}  p10 {np.percentile(per, 10):7.0f}  p90 {np.percentile(per, 90):7.0f}  cycles per stage')
        print(f'     sum {(sel[:, :, :7].sum(2) / st).mean():7.0f}')


if __name__ == '__main__':
    run(16, 128, 128, 256, 256)
    run(16, 512, 512, 64, 64)
