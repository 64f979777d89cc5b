"""Differential fuzz of the convolution / weight-gradient kernels against the framework's own GPU convolutions (an independent
implementation): random (kind, B, K, M, H, W), style scales / demodulation / bias / activation on or off.
    python tools/conv_fuzz.py [cases] [seed]
Prints every case whose relative error exceeds 2e-5 (fp32 accumulation-order noise is ~1e-6) and a summary line."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402

DEV = 'cuda'
TOL = 2e-5


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def one(g, kind):
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    B = ri(1, 5)
    K = ri(1, 9) * 16 if ri(0, 3) else ri(1, 70)            # mostly whole stages (FAST kernels), sometimes ragged
    if kind == '1X1' and ri(0, 1):
        K = ri(1, 4) * 64
    M = [ri(1, 40), ri(41, 100), ri(101, 300)][ri(0, 2)]
    H, W = ri(1, 70), ri(1, 70)
    if ri(0, 2) == 0:
        H, W = 2 ** ri(2, 6), 2 ** ri(2, 6)
    use_isc, use_osc, use_bias, act = ri(0, 1), ri(0, 1), ri(0, 1), [0, 3, 4][ri(0, 2)]
    ks = 1 if kind == '1X1' else 3
    hin, win = (2 * H + 1, 2 * W + 1) if kind == 'S2' else (H, W)
    x = torch.randn(B, K, hin, win, device=DEV, generator=None)
    w = torch.randn(M, K, ks, ks, device=DEV) / (ks * K ** 0.5)
    isc = (0.5 + torch.rand(B, K, device=DEV)) if use_isc else None
    osc = (0.5 + torch.rand(B, M, device=DEV)) if use_osc else None
    bias = torch.randn(M, device=DEV) if use_bias else None
    code = {'T2': _lib.CONV_T2, 'S2': _lib.CONV_S2, '3X3': _lib.CONV_3X3, '1X1': _lib.CONV_1X1}[kind]
    y = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD, 1.0), code, M, H, W, isc, osc, bias, act)
    xs = x * isc[:, :, None, None] if use_isc else x
    if kind == '3X3':
        r = F.conv2d(xs, w, padding=1)
    elif kind == '1X1':
        r = F.conv2d(xs, w)
    elif kind == 'S2':
        r = F.conv2d(xs, w, stride=2)
    else:
        r = F.conv_transpose2d(xs, w.transpose(0, 1), stride=2)
    if use_osc:
        r = r * osc[:, :, None, None]
    if use_bias:
        r = r + bias[None, :, None, None]
    if act:
        r = F.leaky_relu(r, 0.2) * (2 ** 0.5 if act == 3 else 1.0)
    e = rel(y, r)
    desc = f'{kind} B={B} K={K} M={M} H={H} W={W} isc={use_isc} osc={use_osc} bias={use_bias} act={act}'
    # weight gradient of the same layer (T2 takes the (2H+1)x(2W+1) gradient; S2 is the same correlation with roles swapped)
    ew = 0.0
    if kind in ('3X3', '1X1', 'T2'):
        gy = torch.randn_like(r)
        xx = x.clone().requires_grad_(False)
        ww = w.clone().requires_grad_(True)
        if kind == '3X3':
            rr = F.conv2d(xx, ww, padding=1)
        elif kind == '1X1':
            rr = F.conv2d(xx, ww)
        else:
            rr = F.conv_transpose2d(xx, ww.transpose(0, 1), stride=2)
        gw_ref, = torch.autograd.grad((rr * gy).sum(), ww)
        slabs = _lib.wgrad_slabs(gy, x, code, H, W)
        gw = slabs.sum(dim=(0, 1)).reshape(M, K, ks, ks)
        ew = rel(gw, gw_ref)
    return max(e, ew), e, ew, desc


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    bad, worst = 0, (0.0, '')
    for i in range(n):
        kind = ('3X3', '1X1', 'T2', 'S2')[i % 4]
        err, e, ew, desc = one(g, kind)
        if err > worst[0]:
            worst = (err, desc)
        if err > TOL:
            bad += 1
            print(f'[{i}] MISMATCH fwd {e:.2e} wgrad {ew:.2e}: {desc}', flush=True)
    print(f'conv_fuzz: {n} cases, {bad} above {TOL:g}; worst {worst[0]:.2e} ({worst[1]})')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
