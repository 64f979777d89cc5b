"""TE_CONV_3X3W6: the ping-pong form (round 5, te_conv_wino6_form(1)) against the block-phase form (round 4, form 0) of ONE library:
bit identity of the outputs (every epilogue stage, edge tiles on all sides, single- and multi-tile images) and HIP-event time at the
FFHQ-256 / batch-16 layer shapes.

    python tools/wino6_ab.py                     # the product library
    VARIANT=name python tools/wino6_ab.py        # tools/exp/libte_<name>.so built by tools/exp_build.py
"""
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


def rel2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    print('variant', os.environ.get('VARIANT', 'product'), flush=True)
    small = [(2, 32, 128, 8, 32), (3, 64, 256, 16, 64), (1, 48 * 2, 128, 24, 32), (5, 32, 384, 8, 32), (40, 32, 64, 8, 32), (2, 32, 64, 8, 32), (3, 96, 192, 24, 32), (2, 64, 64, 16, 64), (1, 32, 128, 40, 96), (2, 160, 64, 8, 64), (1, 48 * 2, 64, 32, 32)]
    big = [(16, 128, 128, 256, 256), (16, 256, 256, 128, 128), (16, 512, 512, 64, 64), (16, 512, 512, 32, 32)]
    bad = 0
    only_big = bool(os.environ.get('ONLY_BIG'))      # tuning variants: timing at the four large shapes only
    for B, K, M, H, W in ([] if only_big else small) + ([] if os.environ.get('SMALL') else big):
        assert _lib.wino6_ok(B, K, M, H, W), (B, K, M, H, W)
        torch.manual_seed(0)
        x = torch.randn(B, K, H, W, device=DEV)
        w = torch.randn(M, K, 3, 3, device=DEV) / (3 * math.sqrt(K))
        isc = 1 + 0.3 * torch.randn(B, K, device=DEV)
        osc = 1 + 0.3 * torch.randn(B, M, device=DEV)
        bias = torch.randn(M, device=DEV)
        u6 = _lib.conv_pack(w, _lib.PACK_W6FWD, 0.83)
        f6 = lambda: _lib.conv(x, u6, _lib.CONV_3X3W6, M, H, W, isc, osc, bias, 3)
        out = {}
        forms = (0, 1, 3) if os.environ.get('FORM2') else (0, 1)
        for form in forms:
            _lib.wino6_form(form)
            out[form] = f6()
        same = all(torch.equal(out[0], out[f]) for f in forms)
        bad += 0 if same else 1
        msg = f'B{B} {K}->{M} @{H}x{W}: forms bit-identical {same}'
        if not same:
            d = (out[0] - out[1]).abs()
            msg += f' (max diff {float(d.max()):.3e}, {int((d > 0).sum())} of {d.numel()} differ)'
        if B * K * M * H * W <= 2 ** 31:
            want = F.leaky_relu(F.conv2d(x.double() * isc.double()[:, :, None, None], w.double() * 0.83, padding=1)
                                * osc.double()[:, :, None, None] + bias.double()[None, :, None, None], 0.2) * math.sqrt(2)
            msg += f' | vs fp64 (L2): ping-pong {rel2(out[1], want):.2e}, block-phase {rel2(out[0], want):.2e}'
        flops = 2.0 * 9 * K * M * H * W * B
        t = {}
        for form in forms + forms:
            _lib.wino6_form(form)
            t[form] = min(t.get(form, 1e9), timeit(f6, n=20))
        msg += f' | block-phase {t[0] * 1e3:8.1f} us {flops / t[0] / 1e9:6.1f} TF/s, ping-pong {t[1] * 1e3:8.1f} us {flops / t[1] / 1e9:6.1f} TF/s'
        if 3 in t:
            msg += f', two-image {t[3] * 1e3:8.1f} us {flops / t[3] / 1e9:6.1f} TF/s'
        print(msg, flush=True)
    # epilogue stages (residual + mask, no activation / activation) and the data-gradient packing, small shapes, both forms
    for B, K, M, H, W in ([] if only_big else [(2, 64, 128, 16, 32), (1, 32, 64, 8, 96)]):
        x = torch.randn(B, K, H, W, device=DEV)
        w = torch.randn(M, K, 3, 3, device=DEV) / (3 * math.sqrt(K))
        res = torch.randn(B, M, H, W, device=DEV)
        mref = torch.randn(B, M, H, W, device=DEV)
        u6 = _lib.conv_pack(w, _lib.PACK_W6FWD, 1.0)
        ud = _lib.conv_pack(w, _lib.PACK_FWD, 1.0)
        for act in (0, 3):
            for r, m in ((res, None), (None, mref), (res, mref)):
                o = {}
                for form in ((0, 1, 3) if os.environ.get('FORM2') else (0, 1)):
                    _lib.wino6_form(form)
                    o[form] = _lib.conv(x, u6, _lib.CONV_3X3W6, M, H, W, None, None, None, act, res=r, mask_ref=m, mask_gain=2 ** 0.5)
                od = _lib.conv(x, ud, _lib.CONV_3X3, M, H, W, None, None, None, act, res=r, mask_ref=m, mask_gain=2 ** 0.5)
                same = all(torch.equal(o[0], v) for v in o.values())
                bad += 0 if same else 1
                print(f'epilogue B{B} {K}->{M} @{H}x{W} act {act} res {r is not None} mask {m is not None}: bit-identical {same}, '
                      f'vs direct kernel {rel2(o[1], od):.2e}', flush=True)
        g = torch.randn(B, M, H, W, device=DEV)
        want = F.conv_transpose2d(g.double(), w.double(), padding=1)
        ug = _lib.conv_pack(w, _lib.PACK_W6DGRAD)
        for form in ((0, 1) if _lib.wino6_ok(B, M, K, H, W) else ()):
            _lib.wino6_form(form)
            got = _lib.conv(g, ug, _lib.CONV_3X3W6, K, H, W)
            print(f'dgrad form {form} B{B} {M}->{K} @{H}x{W}: {rel2(got, want):.2e}', flush=True)
    _lib.wino6_form(2)
    print('MISMATCHES', bad, flush=True)


if __name__ == '__main__':
    main()
