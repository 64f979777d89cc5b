"""TE_CONV_S2S6 (the 3x3 / stride 2 convolution on the bf16 matrix pipe, three-piece split) against fp64 torch and the fp32 kernel
(TE_CONV_S2): error and time at the FFHQ-256 / batch-16 layer shapes and at small shapes, both weight layouts, every epilogue stage.
    python tools/s2s6_check.py            VARIANT=name: tools/exp/libte_<name>.so"""
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
    print('variant', os.environ.get('VARIANT', 'product'), flush=True)
    small = [(2, 32, 64, 8, 16), (3, 96, 192, 24, 32), (1, 48, 64, 16, 48), (2, 160, 128, 8, 16)]
    big = [(16, 128, 256, 128, 128), (16, 256, 512, 64, 64), (16, 512, 512, 32, 32), (16, 512, 512, 16, 16), (32, 128, 256, 128, 128)]
    bad = 0
    for B, K, M, H, W in small + ([] if os.environ.get('SMALL') else big):
        assert _lib.s2s6_ok(B, K, M, H, W), (B, K, M, H, W)
        torch.manual_seed(0)
        x = torch.randn(B, K, 2 * H + 1, 2 * W + 1, device=DEV)
        w = torch.randn(M, K, 3, 3, device=DEV) / (3 * math.sqrt(K))
        isc = 1 + 0.3 * torch.randn(B, K, device=DEV)
        osc = 1 + 0.3 * torch.randn(B, M, device=DEV)
        bias = torch.randn(M, device=DEV)
        u6 = _lib.conv_pack(w, _lib.PACK_S6FWD, 0.83)
        ud = _lib.conv_pack(w, _lib.PACK_FWD, 0.83)
        f6 = lambda: _lib.conv(x, u6, _lib.CONV_S2S6, M, H, W, isc, osc, bias, 3)
        fd = lambda: _lib.conv(x, ud, _lib.CONV_S2, M, H, W, isc, osc, bias, 3)
        y6, yd = f6(), fd()
        msg = f'B{B} {K}->{M} @{H}x{W}:'
        if B * K * M * H * W <= 2 ** 30:
            want = F.leaky_relu(F.conv2d(x.double() * isc.double()[:, :, None, None], w.double() * 0.83, stride=2)
                                * osc.double()[:, :, None, None] + bias.double()[None, :, None, None], 0.2) * math.sqrt(2)
            e6, ed = rel2(y6, want), rel2(yd, want)
            msg += f' vs fp64: split {rel(y6, want):.2e} / {e6:.2e} (max / L2), fp32 kernel {rel(yd, want):.2e} / {ed:.2e}'
            bad += 0 if rel(y6, want) < 5e-6 else 1
        else:
            msg += f' vs fp32 kernel: {rel(y6, yd):.2e} / {rel2(y6, yd):.2e}'
            bad += 0 if rel(y6, yd) < 5e-6 else 1
        flops = 2.0 * 9 * K * M * H * W * B
        t6, td = min(timeit(f6, n=20), timeit(f6, n=20)), min(timeit(fd, n=20), timeit(fd, n=20))
        msg += f' | split {t6 * 1e3:8.1f} us {flops / t6 / 1e9:6.1f} TF/s, fp32 kernel {td * 1e3:8.1f} us {flops / td / 1e9:6.1f} TF/s'
        if hasattr(_lib.lib(), 'te_conv_s2s6_form') and M % 128 == 0:       # the ping-pong form of the same library, alternating
            oldf = _lib.s2s6_form(0)
            y0 = f6()
            t0 = min(timeit(f6, n=20), timeit(f6, n=20))
            _lib.s2s6_form(2)
            y2 = f6()
            t2 = min(timeit(f6, n=20), timeit(f6, n=20))
            _lib.s2s6_form(oldf)
            bad += 0 if torch.equal(y0, y2) else 1
            msg += f' | ping-pong {t0 * 1e3:8.1f} us {flops / t0 / 1e9:6.1f}, two-image {t2 * 1e3:8.1f} us {flops / t2 / 1e9:6.1f} TF/s, identical {torch.equal(y0, y2)}'
        print(msg, flush=True)
    # the swapped layout (the launch as data gradient of the transposed kind: M = Ci, K = Co) and the epilogue stages
    for B, Co, Ci, H, W in [(2, 64, 128, 8, 16), (1, 128, 64, 16, 32)]:
        g = torch.randn(B, Co, 2 * H + 1, 2 * W + 1, device=DEV)
        w = torch.randn(Co, Ci, 3, 3, device=DEV) / (3 * math.sqrt(Co))         # the model's weight of a Ci -> Co up-sampling layer
        want = F.conv2d(g.double(), w.double().transpose(0, 1), stride=2)        # adjoint of conv_transpose2d(x, w^T-layout, stride 2)
        got = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_S6SWAP), _lib.CONV_S2S6, Ci, H, W)
        ref = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_SWAP), _lib.CONV_S2, Ci, H, W)
        print(f'swap layout B{B} {Co}->{Ci} @{H}x{W}: vs fp64 {rel(got, want):.2e}, fp32 kernel vs fp64 {rel(ref, want):.2e}', flush=True)
        bad += 0 if rel(got, want) < 5e-6 else 1
    B, K, M, H, W = 2, 64, 128, 8, 32
    x = torch.randn(B, K, 2 * H + 1, 2 * W + 1, device=DEV)
    w = torch.randn(M, K, 3, 3, device=DEV) / (3 * math.sqrt(K))
    res, mref, bias = torch.randn(B, M, H, W, device=DEV), torch.randn(B, M, H, W, device=DEV), torch.randn(M, device=DEV)
    u6, ud = _lib.conv_pack(w, _lib.PACK_S6FWD), _lib.conv_pack(w, _lib.PACK_FWD)
    for act in (0, 3, 4):
        for r, m in ((None, None), (res, None), (None, mref), (res, mref)):
            a = _lib.conv(x, u6, _lib.CONV_S2S6, M, H, W, None, None, bias, act, res=r, mask_ref=m, mask_gain=1.3)
            # (the fp32 kernel of this kind has no residual / mask stages: apply them to its plain output)
            b_ = _lib.conv(x, ud, _lib.CONV_S2, M, H, W, None, None, bias, act)
            if r is not None:
                b_ = b_ + r
            if m is not None:
                b_ = b_ * torch.where(m > 0, 1.3, 0.2 * 1.3)
            pre = F.conv2d(x.double(), w.double(), stride=2) + bias.double()[None, :, None, None]
            keep = (pre.abs() > 1e-5) if act else torch.ones_like(pre, dtype=torch.bool)
            e = rel(a * keep, b_ * keep)
            bad += 0 if e < 5e-6 else 1
            print(f'epilogue act {act} res {r is not None} mask {m is not None}: vs fp32 kernel {e:.2e}', flush=True)
    print('FAILURES', bad, flush=True)


if __name__ == '__main__':
    main()
