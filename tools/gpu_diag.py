"""GPU-box diagnostic: quick correctness probes (never abort; print every error) + kernel micro-benchmarks at
the FFHQ-256 / batch-16 layer shapes.  Output is meant to be read from gpurun_out/diag.log."""
import math
import os
import sys
import time
import traceback

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transeditor_amd import _lib, synth                      # noqa: E402
from transeditor_amd.op.modconv import conv_core, modconv    # noqa: E402

DEV = 'cuda'


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def probe(name, fn):
    try:
        print(f'[probe] {name}: {fn()}', flush=True)
    except Exception:
        print(f'[probe] {name}: EXCEPTION\n{traceback.format_exc()}', flush=True)


def conv_probe(kind, B, K, M, H, W):
    ks = 1 if kind == '1x1' else 3
    x = synth.normal((B, K, H, W), 'd.x')
    w = synth.normal((M, K, ks, ks), 'd.w') / math.sqrt(K * ks * ks)
    if kind == '3x3':
        ref = F.conv2d(x, w, padding=1)
    elif kind == '1x1':
        ref = F.conv2d(x, w)
    else:
        ref = F.conv_transpose2d(x, w.transpose(0, 1), stride=2)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = conv_core(xd, wd, kind)
    e = (y.detach().cpu().double() - ref.double()).abs()
    msg = f'fwd rel {rel(y, ref):.2e}'
    if rel(y, ref) > 1e-4:
        idx = torch.nonzero(e > 1e-3 * ref.abs().max())
        msg += f' | {idx.shape[0]} bad of {e.numel()}, first {idx[:6].tolist()}'
    gy = synth.normal(tuple(ref.shape), 'd.g')
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    if kind == '3x3':
        r2 = F.conv2d(xr, wr, padding=1)
    elif kind == '1x1':
        r2 = F.conv2d(xr, wr)
    else:
        r2 = F.conv_transpose2d(xr, wr.transpose(0, 1), stride=2)
    gxr, gwr = torch.autograd.grad((r2 * gy).sum(), (xr, wr))
    gx, gw = torch.autograd.grad((y * gy.to(DEV)).sum(), (xd, wd))
    msg += f' | dgrad rel {rel(gx, gxr):.2e} | wgrad rel {rel(gw, gwr):.2e}'
    return msg


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def bench_layers():
    B = 16
    layers = [('3x3', 512, 512, 4), ('up', 512, 512, 4), ('3x3', 512, 512, 8), ('up', 512, 512, 8), ('3x3', 512, 512, 16),
              ('up', 512, 512, 16), ('3x3', 512, 512, 32), ('up', 512, 512, 32), ('3x3', 512, 512, 64), ('up', 512, 256, 64),
              ('3x3', 256, 256, 128), ('up', 256, 128, 128), ('3x3', 128, 128, 256), ('1x1', 128, 3, 256)]
    tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    for kind, K, M, H in layers:
        ks = 1 if kind == '1x1' else 3
        x = torch.randn(B, K, H, H, device=DEV)
        w = torch.randn(M, K, ks, ks, device=DEV) / math.sqrt(K * ks * ks)
        kk = {'3x3': _lib.CONV_3X3, '1x1': _lib.CONV_1X1, 'up': _lib.CONV_T2}[kind]
        wp = _lib.conv_pack(w, _lib.PACK_FWD)
        y = _lib.conv(x, wp, kk, M, H, H)
        flops = 2.0 * ks * ks * K * M * H * H * B
        t_f = timeit(lambda: _lib.conv(x, wp, kk, M, H, H))
        g = torch.randn_like(y)
        if kind == 'up':
            wpd = _lib.conv_pack(w, _lib.PACK_SWAP)
            t_d = timeit(lambda: _lib.conv(g, wpd, _lib.CONV_S2, K, H, H))
        else:
            wpd = _lib.conv_pack(w, _lib.PACK_DGRAD)
            t_d = timeit(lambda: _lib.conv(g, wpd, kk, K, H, H))
        t_w = timeit(lambda: _lib.wgrad_slabs(g, x, kk, H, H))
        S = _lib.lib().te_wgrad_slab_count(kk, B, M, K, H, H)
        sl = _lib.wgrad_slabs(g, x, kk, H, H)
        t_r = timeit(lambda: _lib.wgrad_reduce(sl, w.reshape(M, K, -1), 1.0, None, None, True, True, True))
        t_p = timeit(lambda: _lib.conv_pack(w, _lib.PACK_FWD))
        tot['fwd'] += t_f; tot['dgrad'] += t_d; tot['wgrad'] += t_w + t_r
        print(f'[bench] {kind:4s} {K:3d}->{M:3d} @{H:3d}: fwd {t_f:7.3f} ms {flops / t_f / 1e9:6.1f} TF | dgrad {t_d:7.3f} ms '
              f'{flops / t_d / 1e9:6.1f} TF | wgrad {t_w:7.3f} ms {flops / t_w / 1e9:6.1f} TF (S={S}) | reduce {t_r:6.3f} ms | '
              f'pack {t_p:6.3f} ms', flush=True)
    print(f'[bench] totals ms: {tot}', flush=True)
    # HBM-bound ops at the top resolution
    x = torch.randn(B, 128, 256, 256, device=DEV)
    b = torch.randn(128, device=DEV)
    t = timeit(lambda: _lib.bias_act(x, b, None, 3, 0, 0.2, 2 ** 0.5))
    print(f'[bench] bias_act 16x128x256x256: {t:.3f} ms  {2 * x.numel() * 4 / t / 1e6:.0f} GB/s', flush=True)
    t = timeit(lambda: _lib.bias_act_bwd(x, x, 0.2, 2 ** 0.5))
    print(f'[bench] bias_act_bwd           : {t:.3f} ms  {3 * x.numel() * 4 / t / 1e6:.0f} GB/s', flush=True)
    xt = torch.randn(B, 128, 257, 257, device=DEV)
    k = torch.tensor([1., 3., 3., 1.], device=DEV)
    k = torch.outer(k, k) / 16
    t = timeit(lambda: _lib.upfirdn2d_raw(xt, k, (1, 1), (1, 1), (1, 1, 1, 1)))
    print(f'[bench] blur 16x128x257^2->256^2: {t:.3f} ms  {(xt.numel() + x.numel()) * 4 / t / 1e6:.0f} GB/s', flush=True)
    t = timeit(lambda: _lib.upfirdn2d_raw(xt, k, (1, 1), (1, 1), (1, 1, 1, 1), bias=b, act=3, scale=2 ** 0.5))
    print(f'[bench] blur+bias+act           : {t:.3f} ms  {(xt.numel() + x.numel()) * 4 / t / 1e6:.0f} GB/s', flush=True)
    xs = torch.randn(B, 3, 128, 128, device=DEV)
    t = timeit(lambda: _lib.upfirdn2d_raw(xs, k * 4, (2, 2), (1, 1), (2, 1, 2, 1)))
    print(f'[bench] skip upsample 3x128->256: {t:.3f} ms', flush=True)


def main():
    print(torch.cuda.get_device_name(0), torch.__version__, flush=True)
    for kind in ('3x3', '1x1', 'up'):
        for shp in [(3, 6, 5, 7, 7), (2, 40, 36, 12, 12), (16, 128, 128, 4, 4), (2, 136, 130, 33, 20), (1, 16, 32, 40, 72)]:
            probe(f'conv {kind} {shp}', lambda: conv_probe(kind, *shp))
    probe('layer microbench', bench_layers)


if __name__ == '__main__':
    main()
