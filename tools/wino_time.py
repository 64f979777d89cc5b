"""EXPERIMENT: the 1-D Winograd F(2,3) kernel of tools/exp/wino3x3.hip against the product's direct fp32-MFMA kernel
(te_conv_f32, kind 3x3) at FFHQ-256 / batch-16 layer shapes: max relative error and time (HIP events)."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402

SRC = os.path.join(ROOT, 'tools/exp/wino3x3.hip')


def load(kc, db, occ, nr=2, mt=2, pipe=1, extra=()):
    so = os.path.join(ROOT, f'tools/exp/libwino_{kc}_{db}_{occ}_{nr}_{mt}_{pipe}{"_" + "".join(extra).replace("-D", "").replace("=", "") if extra else ""}.so')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(SRC):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', f'-DWKC={kc}',
                               f'-DWDB={db}', f'-DWOCC={occ}', f'-DWNR={nr}', f'-DWMT={mt}', f'-DWPIPE={pipe}', *extra, SRC, '-o', so])
    W = C.CDLL(so)
    W.wino3x3_f32.restype = C.c_int
    W.wino3x3_f32.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
    return W


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def transform_weights(w, kc):
    """[M, K, 3, 3] -> U [K/kc][3][4][kc][M]:  U[ky][c] = G w[.., ky, :],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]"""
    M, K = w.shape[:2]
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], device=w.device, dtype=w.dtype)
    u = torch.einsum('cx,mkyx->yckm', G, w)                       # [3][4][K][M]
    return u.reshape(3, 4, K // kc, kc, M).permute(2, 0, 1, 3, 4).contiguous()


def load_pp(src='tools/exp/wino_pp.hip', flags=()):
    so = os.path.join(ROOT, src.replace('.hip', '') + ''.join(flags).replace('-D', '_').replace('=', '') + '.so')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ROOT, src)):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', *flags, os.path.join(ROOT, src), '-o', so])
    W = C.CDLL(so)
    W.wino3x3_f32.restype = C.c_int
    W.wino3x3_f32.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
    return W


PP = [a for a in sys.argv[1:] if a.startswith('pp')]
sys.argv = [a for a in sys.argv if not a.startswith('pp')]
VARIANTS = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(8, 0, 2, 2), (8, 1, 1, 4), (8, 0, 1, 4), (4, 1, 2, 4)]
SHAPES = ((16, 128, 128, 256), (16, 256, 256, 128), (16, 512, 512, 64))
for B, K, M, H in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, K, H, H, device='cuda')
    w = torch.randn(M, K, 3, 3, device='cuda') / (3 * K ** 0.5)
    isc = 1 + 0.1 * torch.randn(B, K, device='cuda')
    wp = _lib.conv_pack(w, _lib.PACK_FWD, 1.0)
    ref = _lib.conv(x, wp, _lib.CONV_3X3, M, H, H, isc, None, None, 0)
    flops = 2.0 * 9 * K * M * H * H * B
    t_d = timeit(lambda: _lib.conv(x, wp, _lib.CONV_3X3, M, H, H, isc, None, None, 0))
    print(f'B{B} {K}->{M} @{H}: direct {t_d * 1e3:8.1f} us {flops / t_d / 1e9:6.1f} TF/s', flush=True)
    st = torch.cuda.current_stream().cuda_stream
    for spec in PP:
        W = load_pp(flags=tuple(f'-D{f}' for f in spec.split(':')[1:]))
        U = transform_weights(w, 8)
        out = torch.empty_like(ref)
        run = lambda: W.wino3x3_f32(out.data_ptr(), x.data_ptr(), U.data_ptr(), isc.data_ptr(), B, K, M, H, H, st)
        rc = run()
        torch.cuda.synchronize()
        err = float((out - ref).abs().max() / ref.abs().max())
        t_w = timeit(run)
        print(f'    winograd ping-pong {spec}: rc {rc} err {err:.2e}  {t_w * 1e3:8.1f} us {flops / t_w / 1e9:6.1f} TF/s (algorithmic), '
              f'MFMA-equivalent {flops / t_w / 1e9 / 1.5:6.1f}', flush=True)
    for var in VARIANTS:
        kc, db, occ, nr = var[:4]
        mt, pipe = (var[4] if len(var) > 4 else 2), (var[5] if len(var) > 5 else 1)
        W = load(kc, db, occ, nr, mt, pipe)
        U = transform_weights(w, kc)
        out = torch.empty_like(ref)
        run = lambda: W.wino3x3_f32(out.data_ptr(), x.data_ptr(), U.data_ptr(), isc.data_ptr(), B, K, M, H, H, st)
        rc = run()
        torch.cuda.synchronize()
        err = float((out - ref).abs().max() / ref.abs().max())
        t_w = timeit(run)
        print(f'    winograd KC={kc} DB={db} occ={occ} rows={2 * nr} Mtiles/wave={mt} pipe={pipe}: rc {rc} err {err:.2e}  {t_w * 1e3:8.1f} us {flops / t_w / 1e9:6.1f} TF/s (algorithmic), '
              f'MFMA-equivalent {flops / t_w / 1e9 / 1.5:6.1f}', flush=True)
