"""The split-bf16 weight-gradient kernel (csrc/wgrad6.hip) against the fp32 kernel (same slabs) and against fp64 torch: error and
time at the FFHQ-256 / batch-16 layer shapes and at small shapes (chunk boundaries inside a column, grouped samples).
    python tools/wgrad6_check.py            VARIANT=name: tools/exp/libte_<name>.so      PROF=1: cycle counts (-DWG6_PROF build)"""
import ctypes
import math
import os
import sys

import numpy as np
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


def prof(nblocks):
    buf = np.zeros(2048 * 4 * 4, dtype=np.uint64)
    L = _lib.lib()
    L.te_debug_wgrad6_prof.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    assert L.te_debug_wgrad6_prof(buf.ctypes.data, buf.nbytes) == 0
    raw = buf.reshape(-1, 4, 4)[:min(2048, nblocks)]
    nst = (raw[:, :, 3] >> np.uint64(40)).astype(np.float64)
    tot = (raw[:, :, 3] & np.uint64((1 << 40) - 1)).astype(np.float64)
    d = raw.astype(np.float64)
    k = nst[:, 0] > 0
    return (f'steps/block {nst[k].mean():.0f}; cycles per step: stream {np.mean(d[k][:, :, 0] / nst[k]):.0f}, barrier+copy '
            f'{np.mean(d[k][:, :, 2] / nst[k]):.0f}, sweep heads {np.mean(d[k][:, :, 1] / nst[k]):.0f}; loop total {np.mean(tot[k] / nst[k]):.0f} '
            f'(matrix pipe needs 2304)')


def main():
    print('variant', os.environ.get('VARIANT', 'product'), flush=True)
    small = [(2, 64, 64, 8, 32), (3, 128, 64, 13, 64), (1, 64, 192, 40, 96), (4, 64, 64, 5, 32), (2, 128, 128, 32, 32)]
    big = [(16, 128, 128, 256, 256), (16, 256, 256, 128, 128), (16, 512, 512, 64, 64), (16, 512, 512, 32, 32), (32, 128, 128, 128, 128)]
    bad = 0
    for B, Co, Ci, H, W in small + ([] if os.environ.get('SMALL') else big):
        assert _lib.wgrad_split_ok(_lib.CONV_3X3, Co, Ci, H, W), (Co, Ci, H, W)
        torch.manual_seed(0)
        g = torch.randn(B, Co, H, W, device=DEV)
        x = torch.randn(B, Ci, H, W, device=DEV)
        out = {}
        for on in (0, 1):
            _lib.wgrad_split(on)
            out[on] = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W)
        msg = f'B{B} {Ci}->{Co} @{H}x{W} (S={out[0].shape[1]}):'
        s0, s1 = out[0].sum(1), out[1].sum(1)              # per-sample correlation [B, Co, Ci, 9]
        if B * Co * Ci * H * W <= 2 ** 28:
            # fp64 reference: dW[b] = correlation of g[b] and x[b]
            want = torch.stack([torch.nn.grad.conv2d_weight(x[b:b + 1].double(), (Co, Ci, 3, 3), g[b:b + 1].double(), padding=1) for b in range(B)])
            want = want.reshape(B, Co, Ci, 9)
            e1, e0 = rel(s1, want), rel(s0, want)
            msg += f' vs fp64 (max / L2): split {e1:.2e} / {rel2(s1, want):.2e}, fp32 kernel {e0:.2e} / {rel2(s0, want):.2e}'
            bad += 0 if e1 < 5e-6 else 1
        else:
            e = rel(s1, s0)
            msg += f' vs fp32 kernel: {e:.2e} / {rel2(s1, s0):.2e}'
            bad += 0 if e < 5e-6 else 1
        flops = 2.0 * 9 * Co * Ci * H * W * B
        t = {}
        for on in (0, 1, 0, 1):
            _lib.wgrad_split(on)
            t[on] = min(t.get(on, 1e9), timeit(lambda: _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W), n=10))
        msg += f' | split {t[1] * 1e3:8.1f} us {flops / t[1] / 1e9:6.1f} TF/s, fp32 kernel {t[0] * 1e3:8.1f} us {flops / t[0] / 1e9:6.1f} TF/s'
        if os.environ.get('PROF'):
            _lib.wgrad_split(1)
            _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W)
            torch.cuda.synchronize()
            msg += ' | ' + prof(B * out[1].shape[1] * (Co // 64) * (Ci // 64))
        print(msg, flush=True)
    # ---- transposed kind: slab[co][ci][ky][kx] = sum g[co, 2i + ky, 2j + kx] x[ci, i, j]
    small_t = [(2, 64, 64, 8, 16), (3, 128, 64, 13, 32), (1, 64, 192, 20, 48), (4, 64, 64, 5, 16), (1, 64, 64, 1, 16)]
    big_t = [(16, 128, 128, 128, 128), (16, 256, 256, 64, 64), (16, 512, 512, 32, 32), (16, 512, 512, 16, 16), (32, 256, 128, 64, 64),
             (16, 128, 256, 128, 128), (16, 256, 512, 64, 64), (32, 128, 256, 128, 128)]       # (the last three: the model's own (Co, Ci) at these sizes)
    for B, Co, Ci, H, W in small_t + ([] if os.environ.get('SMALL') else big_t):
        assert _lib.wgrad_split_ok(_lib.CONV_T2, Co, Ci, H, W), (Co, Ci, H, W)
        torch.manual_seed(0)
        g = torch.randn(B, Co, 2 * H + 1, 2 * W + 1, device=DEV)
        x = torch.randn(B, Ci, H, W, device=DEV)
        out = {}
        for on in (0, 1):
            _lib.wgrad_split(on)
            out[on] = _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W)
        msg = f'T2 B{B} {Ci}->{Co} @{H}x{W} (S={out[0].shape[1]}):'
        s0, s1 = out[0].sum(1), out[1].sum(1)
        if B * Co * Ci * H * W <= 2 ** 26:
            # fp64 reference: the weight gradient of a stride-2 convolution from g (the big image) to x's grid
            want = torch.stack([torch.nn.grad.conv2d_weight(g[b:b + 1].double(), (Ci, Co, 3, 3), x[b:b + 1].double(), stride=2)
                                for b in range(B)]).transpose(1, 2).reshape(B, Co, Ci, 9)
            e1, e0 = rel(s1, want), rel(s0, want)
            msg += f' vs fp64 (max / L2): split {e1:.2e} / {rel2(s1, want):.2e}, fp32 kernel {e0:.2e} / {rel2(s0, want):.2e}'
            bad += 0 if e1 < 5e-6 else 1
        else:
            e = rel(s1, s0)
            msg += f' vs fp32 kernel: {e:.2e} / {rel2(s1, s0):.2e}'
            bad += 0 if e < 5e-6 else 1
        flops = 2.0 * 9 * Co * Ci * H * W * B
        t = {}
        for on in (0, 1, 0, 1):
            _lib.wgrad_split(on)
            t[on] = min(t.get(on, 1e9), timeit(lambda: _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W), n=10))
        msg += f' | split {t[1] * 1e3:8.1f} us {flops / t[1] / 1e9:6.1f} TF/s, fp32 kernel {t[0] * 1e3:8.1f} us {flops / t[0] / 1e9:6.1f} TF/s'
        if hasattr(_lib.lib(), 'te_wgrad_t2_wide') and Ci % 128 == 0 and Co % 64 == 0:       # 64 x 64 form of the same library, alternating
            _lib.wgrad_split(1)
            tw = {}
            ow = {}
            for wide in (0, 1, 0, 1):
                oldw = _lib.wgrad_t2_wide(wide)
                ow[wide] = _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W)
                tw[wide] = min(tw.get(wide, 1e9), timeit(lambda: _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W), n=10))
                _lib.wgrad_t2_wide(oldw)
            same = torch.equal(ow[0], ow[1])
            bad += 0 if same else 1
            msg += f' | 64x64 {tw[0] * 1e3:8.1f} us {flops / tw[0] / 1e9:6.1f}, 64x128 {tw[1] * 1e3:8.1f} us {flops / tw[1] / 1e9:6.1f} TF/s, identical {same}'
        if os.environ.get('PROF'):
            _lib.wgrad_split(1)
            _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W)
            torch.cuda.synchronize()
            msg += ' | ' + prof(B * out[1].shape[1] * (Co // 64) * (Ci // 64)).replace('2304', '1728')
        print(msg, flush=True)
    # ---- 1x1 kind (round 6): slab[co][ci] = sum g[co, cell] x[ci, cell]
    for B, Co, Ci, H, W in [(2, 128, 128, 8, 16), (3, 256, 128, 13, 32)] + ([] if os.environ.get('SMALL') else
                                                                         [(32, 256, 128, 128, 128), (32, 512, 256, 64, 64), (32, 512, 512, 32, 32), (32, 512, 512, 16, 16),
                                                                          (16, 256, 128, 128, 128)]):
        assert _lib.wgrad_split_ok(_lib.CONV_1X1, Co, Ci, H, W), (Co, Ci, H, W)
        torch.manual_seed(0)
        g = torch.randn(B, Co, H, W, device=DEV)
        x = torch.randn(B, Ci, H, W, device=DEV)
        out = {}
        for on in (0, 1):
            _lib.wgrad_split(on)
            out[on] = _lib.wgrad_slabs(g, x, _lib.CONV_1X1, H, W)
        s0, s1 = out[0].sum(1), out[1].sum(1)
        msg = f'1x1 B{B} {Ci}->{Co} @{H}x{W} (S={out[0].shape[1]}):'
        if B * Co * Ci * H * W <= 2 ** 28:
            want = torch.einsum('bohw,bihw->boi', g.double(), x.double()).unsqueeze(-1)
            e1, e0 = rel(s1, want), rel(s0, want)
            msg += f' vs fp64 (max / L2): split {e1:.2e} / {rel2(s1, want):.2e}, fp32 kernel {e0:.2e} / {rel2(s0, want):.2e}'
            bad += 0 if e1 < 5e-6 else 1
        else:
            e = rel(s1, s0)
            msg += f' vs fp32 kernel: {e:.2e} / {rel2(s1, s0):.2e}'
            bad += 0 if e < 5e-6 else 1
        flops = 2.0 * Co * Ci * H * W * B
        t = {}
        for on in (0, 1, 0, 1):
            _lib.wgrad_split(on)
            t[on] = min(t.get(on, 1e9), timeit(lambda: _lib.wgrad_slabs(g, x, _lib.CONV_1X1, H, W), n=10))
        gb = (g.numel() + x.numel()) * 4 / 1e9
        msg += (f' | split {t[1] * 1e3:8.1f} us {flops / t[1] / 1e9:6.1f} TF/s {gb / t[1]:5.2f} TB/s, fp32 kernel {t[0] * 1e3:8.1f} us '
                f'{flops / t[0] / 1e9:6.1f} TF/s')
        print(msg, flush=True)
    # grouped form (samples share a slab): plain gradient of small images
    for B, Co, Ci, H, W in [(8, 128, 128, 32, 32), (32, 512, 512, 32, 32)]:
        g = torch.randn(B, Co, H, W, device=DEV)
        x = torch.randn(B, Ci, H, W, device=DEV)
        o = {}
        for on in (0, 1):
            _lib.wgrad_split(on)
            o[on] = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W, group=True)
        e = rel(o[1].sum((0, 1)), o[0].sum((0, 1)))
        bad += 0 if e < 5e-6 else 1
        print(f'grouped B{B} {Ci}->{Co} @{H}x{W}: slabs {tuple(o[1].shape)}, split vs fp32 kernel {e:.2e}', flush=True)
    _lib.wgrad_split(0)
    print('BAD', bad, flush=True)


if __name__ == '__main__':
    main()
