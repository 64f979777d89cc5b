"""TE_CONV_T2S6 (the transposed 3x3 / stride 2 convolution on the bf16 matrix pipe, three-piece split; last output row / column
through the fp32 kernel) against fp64 torch and the fp32 kernel (TE_CONV_T2): error and time at the FFHQ-256 / batch-16 layer shapes
and at small shapes, both weight layouts.      python tools/t2s6_check.py"""
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
if os.environ.get('VARIANT'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', f"libte_{os.environ['VARIANT']}.so")


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def rel2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    small = [(2, 32, 64, 8, 16), (3, 96, 192, 24, 32), (1, 48, 64, 16, 48), (2, 160, 128, 8, 16)]
    big = [(16, 256, 128, 128, 128), (16, 512, 256, 64, 64), (16, 512, 512, 32, 32), (16, 512, 512, 16, 16), (32, 256, 128, 128, 128)]
    bad = 0
    for B, K, M, H, W in small + ([] if os.environ.get('SMALL') else big):
        assert _lib.t2s6_ok(B, K, M, H, W), (B, K, M, H, W)
        torch.manual_seed(0)
        x = torch.randn(B, K, H, W, device=DEV)
        w = torch.randn(M, K, 3, 3, device=DEV) / (3 * math.sqrt(K))          # model layout [Co, Ci, 3, 3] of an up-sampling layer
        isc = 1 + 0.3 * torch.randn(B, K, device=DEV)
        osc = 1 + 0.3 * torch.randn(B, M, device=DEV)
        bias = torch.randn(M, device=DEV)
        u6 = _lib.conv_pack(w, _lib.PACK_T6FWD, 0.83)
        ud = _lib.conv_pack(w, _lib.PACK_FWD, 0.83)
        f6 = lambda: _lib.conv(x, u6, _lib.CONV_T2S6, M, H, W, isc, osc, bias, 3)
        fd = lambda: _lib.conv(x, ud, _lib.CONV_T2, M, H, W, isc, osc, bias, 3)
        y6, yd = f6(), fd()
        msg = f'B{B} {K}->{M} @{H}x{W}:'
        if B * K * M * H * W <= 2 ** 30:
            want = F.leaky_relu(F.conv_transpose2d(x.double() * isc.double()[:, :, None, None], (w.double() * 0.83).transpose(0, 1), stride=2)
                                * osc.double()[:, :, None, None] + bias.double()[None, :, None, None], 0.2) * math.sqrt(2)
            msg += f' vs fp64: split {rel(y6, want):.2e} / {rel2(y6, want):.2e} (max / L2), fp32 kernel {rel(yd, want):.2e} / {rel2(yd, want):.2e}'
            bad += 0 if rel(y6, want) < 5e-6 else 1
        else:
            msg += f' vs fp32 kernel: {rel(y6, yd):.2e} / {rel2(y6, yd):.2e}'
            bad += 0 if rel(y6, yd) < 5e-6 else 1
        flops = 2.0 * 9 * K * M * H * W * B
        t6, td = min(timeit(f6, n=20), timeit(f6, n=20)), min(timeit(fd, n=20), timeit(fd, n=20))
        msg += f' | split {t6 * 1e3:8.1f} us {flops / t6 / 1e9:6.1f} TF/s, fp32 kernel {td * 1e3:8.1f} us {flops / td / 1e9:6.1f} TF/s'
        if hasattr(_lib.lib(), 'te_conv_t2s6_form') and M % 128 == 0:       # the ping-pong form of the same library, alternating
            oldf = _lib.t2s6_form(0)
            y0 = f6()
            t0 = min(timeit(f6, n=20), timeit(f6, n=20))
            _lib.t2s6_form(2)
            y2 = f6()
            t2 = min(timeit(f6, n=20), timeit(f6, n=20))
            _lib.t2s6_form(oldf)
            bad += 0 if torch.equal(y0, y2) else 1
            msg += f' | ping-pong {t0 * 1e3:8.1f} us {flops / t0 / 1e9:6.1f}, two-image {t2 * 1e3:8.1f} us {flops / t2 / 1e9:6.1f} TF/s, identical {torch.equal(y0, y2)}'
        print(msg, flush=True)
    # the swapped layout: the launch as data gradient of a strided (down-sampling) convolution with weight [Co, Ci, 3, 3]
    for B, Co, Ci, H, W in [(2, 64, 128, 8, 16), (1, 128, 64, 16, 32)]:
        g = torch.randn(B, Co, H, W, device=DEV)
        w = torch.randn(Co, Ci, 3, 3, device=DEV) / (3 * math.sqrt(Co))
        want = F.conv_transpose2d(g.double(), w.double(), stride=2)               # adjoint of F.conv2d(x, w, stride 2): [B, Ci, 2H+1, 2W+1]
        got = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_T6SWAP), _lib.CONV_T2S6, Ci, H, W)
        ref = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_SWAP), _lib.CONV_T2, Ci, H, W)
        print(f'swap layout B{B} {Co}->{Ci} @{H}x{W}: vs fp64 {rel(got, want):.2e}, fp32 kernel vs fp64 {rel(ref, want):.2e}', flush=True)
        bad += 0 if rel(got, want) < 5e-6 else 1
    print('FAILURES', bad, flush=True)


if __name__ == '__main__':
    main()
