"""Per-phase cycle counts inside the FAST convolution kernels (experimental build with -DTE_CONV_PROF):
    python tools/exp_build.py prof -DTE_CONV_PROF && python tools/conv_phase_prof.py
Wave-level s_memtime stamps around: first barrier of a stage, commit (registers -> LDS), second barrier, MFMA phase."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', 'libte_prof.so')
DEV = 'cuda'


def run(kind, K, M, H, B=16):
    torch.manual_seed(0)
    x = torch.randn(B, K, 2 * H + 1, 2 * H + 1, device=DEV) if kind == 'S2' else torch.randn(B, K, H, H, device=DEV)
    w = torch.randn(M, K, 3, 3, device=DEV) / (3 * K ** 0.5)
    wp = _lib.conv_pack(w, _lib.PACK_FWD, 1.0)
    isc = 1 + 0.1 * torch.randn(B, K, device=DEV)
    code = {'T2': _lib.CONV_T2, 'S2': _lib.CONV_S2, '3X3': _lib.CONV_3X3}[kind]
    for _ in range(3):
        _lib.conv(x, wp, code, M, H, H, isc, None, None, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.conv(x, wp, code, M, H, H, isc, None, None, 0)
    e1.record()
    torch.cuda.synchronize()
    wall_ms = e0.elapsed_time(e1)
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    L = _lib.lib()
    L.te_debug_conv_prof.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    rc = L.te_debug_conv_prof(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    raw = buf.reshape(-1, 8)
    real = (raw[:, 0] >> np.uint64(32)).astype(np.float64)
    raw[:, 0] &= np.uint64(0xFFFFFFFF)
    d = raw.astype(np.float64)
    keep = d[:, 7] > 0
    d, real = d[keep], real[keep]
    print(f'   counter frequency: {((d[:, 5] - d[:, 4]) / (real / 100e6)).mean() / 1e9:.3f} GHz (s_memtime counts per second of s_memrealtime)')
    st = d[:, 7]
    tot = d[:, 5] - d[:, 4]
    print(f'{kind} {K}->{M} @{H}: {len(d)} wave records, stages/block {st[0]:.0f}')
    names = ('barrier1', 'commit', 'barrier2', 'mfma phase')
    for i, n in enumerate(names):
        per = d[:, i] / st
        print(f'   {n:11s}: mean {per.mean():8.0f}  p10 {np.percentile(per, 10):8.0f}  p90 {np.percentile(per, 90):8.0f}   cycles per stage')
    per = d[:, 6] / st
    print(f'   (of commit: waiting for the stage loads, vmcnt(0): mean {per.mean():8.0f}  p10 {np.percentile(per, 10):8.0f}  p90 {np.percentile(per, 90):8.0f})')
    span = buf.reshape(-1, 8)[:, 5].max() - buf.reshape(-1, 8)[:, 4][buf.reshape(-1, 8)[:, 7] > 0].min()
    print(f'   kernel {wall_ms * 1e3:.0f} us by events; counter span over the recorded blocks {span:.3e} -> >= {span / wall_ms / 1e6:.3f} counts/ns')
    blk = (d[:, 5] - d[:, 4])
    print(f'   block duration (K loop start -> end): mean {blk.mean():.0f} p10 {np.percentile(blk, 10):.0f} p50 {np.percentile(blk, 50):.0f} p90 {np.percentile(blk, 90):.0f}')
    loop = d[:, :4].sum(1)
    print(f'   K loop {loop.mean():.0f} of {tot.mean():.0f} cycles per block ({100 * loop.mean() / tot.mean():.1f} %); per stage {(loop / st).mean():.0f}')


if __name__ == '__main__' and 'conv' in sys.argv:
    run('T2', 512, 256, 64)
    run('T2', 256, 128, 128)
    run('3X3', 256, 256, 128)
    run('S2', 256, 512, 64)


def run_wgrad(kind, K, M, H, B=16):
    torch.manual_seed(0)
    code = _lib.CONV_T2 if kind == 'WT2' else _lib.CONV_3X3
    g = torch.randn(B, M, 2 * H + 1, 2 * H + 1, device=DEV) if kind == 'WT2' else torch.randn(B, M, H, H, device=DEV)
    x = torch.randn(B, K, H, H, device=DEV)
    for _ in range(3):
        _lib.wgrad_slabs(g, x, code, H, H)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    L = _lib.lib()
    L.te_debug_wgrad_prof.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    assert L.te_debug_wgrad_prof(buf.ctypes.data, buf.nbytes) == 0
    d = buf.reshape(-1, 8).astype(np.float64)
    d = d[d[:, 7] > 0]
    nt = d[:, 7]
    print(f'{kind} {K}->{M} @{H}: {len(d)} wave records, tiles/block {nt.mean():.1f}')
    for i, n in enumerate(('barrier', 'commit', '-', 'issue loads', 'mfma steps', 'first operands')):
        per = d[:, i] / nt
        print(f'   {n:11s}: mean {per.mean():8.0f}  p10 {np.percentile(per, 10):8.0f}  p90 {np.percentile(per, 90):8.0f}   cycles per tile')
    print(f'   total per tile {(d[:, 6] / nt).mean():.0f}')


if __name__ == '__main__':
    run_wgrad('W3X3', 128, 128, 256)
    run_wgrad('W3X3', 256, 256, 128)
    run_wgrad('WT2', 512, 256, 64)
