"""TE_CONV_3X3W6 (Winograd 3x3 on the bf16 matrix pipe, three-piece split) against fp64 torch, the fp32 Winograd kernel and the
direct kernel: error and time at FFHQ-256 / batch-16 layer shapes and at small shapes.   python tools/wino6_check.py"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402
from tools.exp_time import timeit     # noqa: E402

DEV = 'cuda'
if os.environ.get('VARIANT'):          # a library variant built by tools/exp_build.py
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', f"libte_{os.environ['VARIANT']}.so")


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def rel2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


if __name__ == '__main__':
    small = [(2, 32, 64, 8, 32), (3, 96, 192, 24, 32), (2, 64, 64, 16, 64), (1, 32, 128, 40, 96)]
    print('variant', os.environ.get('VARIANT', 'product'))
    big = [(16, 128, 128, 256, 256), (16, 256, 256, 128, 128), (16, 512, 512, 64, 64), (16, 512, 512, 32, 32)]
    for B, K, M, H, W in small + ([] if os.environ.get('SMALL') else big):
        assert _lib.wino6_ok(B, K, M, H, W), (B, K, M, H, W)
        torch.manual_seed(0)
        x = torch.randn(B, K, H, W, device=DEV)
        w = torch.randn(M, K, 3, 3, device=DEV) / (3 * math.sqrt(K))
        isc = 1 + 0.3 * torch.randn(B, K, device=DEV)
        osc = 1 + 0.3 * torch.randn(B, M, device=DEV)
        bias = torch.randn(M, device=DEV)
        u6 = _lib.conv_pack(w, _lib.PACK_W6FWD, 0.83)
        uw = _lib.conv_pack(w, _lib.PACK_WFWD, 0.83) if _lib.wino_ok(B, K, M, H, W) else None
        ud = _lib.conv_pack(w, _lib.PACK_FWD, 0.83)
        f6 = lambda: _lib.conv(x, u6, _lib.CONV_3X3W6, M, H, W, isc, osc, bias, 3)
        fw = (lambda: _lib.conv(x, uw, _lib.CONV_3X3W, M, H, W, isc, osc, bias, 3)) if uw is not None else None
        fd = lambda: _lib.conv(x, ud, _lib.CONV_3X3, M, H, W, isc, osc, bias, 3)
        y6, yd = f6(), fd()
        msg = f'B{B} {K}->{M} @{H}x{W}:'
        if B * K * M * H * W <= 2 ** 31:
            want = F.leaky_relu(F.conv2d(x.double() * isc.double()[:, :, None, None], w.double() * 0.83, padding=1)
                                * osc.double()[:, :, None, None] + bias.double()[None, :, None, None], 0.2) * math.sqrt(2)
            msg += f' vs fp64: split {rel(y6, want):.2e} / {rel2(y6, want):.2e} (max / L2), direct {rel(yd, want):.2e} / {rel2(yd, want):.2e}'
            if fw is not None:
                yw = fw()
                msg += f', fp32 winograd {rel(yw, want):.2e} / {rel2(yw, want):.2e}'
        else:
            msg += f' vs direct: {rel(y6, yd):.2e} / {rel2(y6, yd):.2e}'
        flops = 2.0 * 9 * K * M * H * W * B
        t6, td = timeit(f6), timeit(fd)
        msg += f' | split {t6 * 1e3:8.1f} us {flops / t6 / 1e9:6.1f} TF/s, direct {td * 1e3:8.1f} us {flops / td / 1e9:6.1f}'
        if fw is not None:
            tw = timeit(fw)
            msg += f', fp32 winograd {tw * 1e3:8.1f} us {flops / tw / 1e9:6.1f}'
        print(msg, flush=True)
    # data-gradient packing against fp64
    B, K, M, H, W = 2, 64, 128, 16, 32
    g = torch.randn(B, M, H, W, device=DEV)
    w = torch.randn(M, K, 3, 3, device=DEV) / (3 * math.sqrt(K))
    want = F.conv_transpose2d(g.double(), w.double(), padding=1)
    got = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_W6DGRAD), _lib.CONV_3X3W6, K, H, W)
    print(f'data gradient {M}->{K} @{H}x{W}: {rel(got, want):.2e} / {rel2(got, want):.2e}')
