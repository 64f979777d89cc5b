"""Per-phase cycle counts inside the split-bf16 strided / transposed kernels (experimental build with -DS2_PROF -DT2_PROF):
    python tools/exp_build.py st_prof -DS2_PROF -DT2_PROF && python tools/s2s6_phase_prof.py [variant]
Wave-level s_memtime stamps, as tools/w6p_phase_prof.py: multiplying role (up to the mid barrier / the wait there / the rest), staging
role (part 1 / mid barrier / part 2), end-of-phase barrier."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', f'libte_{sys.argv[1] if len(sys.argv) > 1 else "st_prof"}.so')
DEV = 'cuda'


def report(sym, wall_us, label, mfma_per_phase, nblocks):
    buf = np.zeros(2048 * 8 * 8, dtype=np.uint64)
    L = _lib.lib()
    fn = getattr(L, sym)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    raw = buf.reshape(-1, 8, 8)[:min(2048, nblocks)]       # (the device buffer keeps the records of earlier launches)
    nst = (raw[:, :, 7] >> np.uint64(48)).astype(np.float64)
    tot = (raw[:, :, 7] & np.uint64(0xFFFFFFFFFFFF)).astype(np.float64)
    real = (raw[:, :, 6] >> np.uint64(40)).astype(np.float64)
    raw[:, :, 6] &= np.uint64((1 << 40) - 1)
    d = raw.astype(np.float64)
    keep = nst[:, 0] > 0
    d, nst, tot, real = d[keep], nst[keep], tot[keep], real[keep]
    per_stage = tot.mean() / nst[0, 0]
    print(f'{label}: {wall_us:.0f} us, {len(d)} blocks recorded, {nst[0, 0]:.0f} stages per block; shader clock '
          f'{(tot / (real / 100e6)).mean() / 1e9:.3f} GHz; loop span per block mean {tot.mean():.0f} = {per_stage:.0f} per stage '
          f'(matrix pipe needs {2 * mfma_per_phase * 32} per stage and SIMD = {2 * mfma_per_phase * 32 / per_stage:.2f})')
    names = ('mult to mid barrier', 'mult mid barrier', 'mult rest', 'stage part 1', 'stage mid barrier', 'stage part 2', 'end barriers (2/stage)')
    for g in (0, 1):
        sel = d[:, 4 * g:4 * g + 4, :]
        st = nst[:, 4 * g:4 * g + 4]
        print(f'  group {g}:')
        for i, n in enumerate(names):
            per = sel[:, :, i] / st
            print(f'     {n:24s} mean {per.mean():7.0f}  p10 {np.percentile(per, 10):7.0f}  p90 {np.percentile(per, 90):7.0f}  cycles per stage')
        print(f'     sum {(sel[:, :, :7].sum(2) / st).mean():7.0f}')


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def run_s2(B, K, M, H, W):
    torch.manual_seed(0)
    x = torch.randn(B, K, 2 * H + 1, 2 * W + 1, device=DEV)
    w = torch.randn(M, K, 3, 3, device=DEV) / (3 * K ** 0.5)
    u = _lib.conv_pack(w, _lib.PACK_S6FWD, 1.0)
    isc = 1 + 0.1 * torch.randn(B, K, device=DEV)
    us = timed(lambda: _lib.conv(x, u, _lib.CONV_S2S6, M, H, W, isc, None, None, 0))
    report('te_debug_s2s6_prof', us, f's2s6 B{B} {K}->{M} out {H}x{W} ({2 * 9 * K * M * H * W * B / us / 1e6:.1f} TF/s)', 54, B * (H // 8) * (W // 16) * (M // 64))


def run_t2(B, K, M, H, W):
    torch.manual_seed(0)
    x = torch.randn(B, K, H, W, device=DEV)
    w = torch.randn(M, K, 3, 3, device=DEV) / (3 * K ** 0.5)
    u = _lib.conv_pack(w, _lib.PACK_T6FWD, 1.0)
    isc = 1 + 0.1 * torch.randn(B, K, device=DEV)
    us = timed(lambda: _lib.conv(x, u, _lib.CONV_T2S6, M, H, W, isc, None, None, 0))
    report('te_debug_t2s6_prof', us, f't2s6 (body + thin regions) B{B} {K}->{M} in {H}x{W} ({2 * 9 * K * M * H * W * B / us / 1e6:.1f} TF/s)', 54, B * (H // 8) * (W // 16) * (M // 64))


if __name__ == '__main__':
    for shp in ((16, 128, 128, 128, 128), (16, 512, 512, 32, 32)):
        run_s2(*shp)
        run_t2(*shp)
