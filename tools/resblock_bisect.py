"""Diagnostic: every intermediate of the fused ResBlock backward against fp64 torch, for the shapes where the D-step gradients
deviate from the fp64 oracle (tools/d_joint_probe.py).   python tools/resblock_bisect.py"""
import math
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from oracle import te_oracle as O
from transeditor_amd import _lib
from transeditor_amd.op.modconv import _dgrad_raw, _wgrad_plain
from transeditor_amd.op.upfirdn2d import _geometry, flipped_taps

DEV = 'cuda'
rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))


def run(B, C, H):
    torch.manual_seed(H * 100 + B)
    k = O.fir_kernel((1, 3, 3, 1)).to(DEV)
    pm = (2, 2, 2, 2)
    y1 = torch.randn(B, C, H, H, device=DEV)                 # conv1 output (after bias + lrelu): only its sign matters
    g2 = torch.randn(B, C, H // 2, H // 2, device=DEV)       # gradient at conv2's pre-activation
    w2 = torch.randn(C, C, 3, 3, device=DEV)
    s2 = 1 / math.sqrt(C * 9)
    x = torch.randn(B, C, H, H, device=DEV)
    # product
    g_yb = _dgrad_raw(g2, w2, 'down', wscale=s2)
    _, g_pad = _geometry((H, H), (4, 4), (1, 1), (1, 1), pm)
    g1, gb1 = _lib.blur_gradact(g_yb, y1, flipped_taps(k), g_pad, 0.2, math.sqrt(2))
    g1u = _lib.upfirdn2d_raw(g_yb, flipped_taps(k), (1, 1), (1, 1), g_pad)
    g1b, gb1b = _lib.bias_act_bwd(g1u, y1, 0.2, math.sqrt(2), want_bias=True)
    gw1 = _wgrad_plain(g1, x, '3x3', 3, 1.0)
    # fp64
    g2d, w2d, y1d, kd = g2.double().cpu(), w2.double().cpu(), y1.double().cpu(), k.double().cpu()
    r_yb = F.conv_transpose2d(g2d, w2d * s2, stride=2)
    r_blur = O.upfirdn2d(r_yb, torch.flip(kd, [0, 1]), pad=(g_pad[0], g_pad[1]) if g_pad[0] == g_pad[2] and g_pad[1] == g_pad[3] else None)
    slope = torch.where(y1d > 0, math.sqrt(2), 0.2 * math.sqrt(2))
    r_g1 = r_blur * slope
    r_gb1 = r_g1.sum(dim=(0, 2, 3))
    # the adjoint blur of the PRODUCT's own g_yb in fp64: separates the blur from the transposed conv
    r_blur_own = O.upfirdn2d(g_yb.double().cpu(), torch.flip(kd, [0, 1]), pad=(g_pad[0], g_pad[1]))
    r_gw1 = torch.einsum('bmhw,bkhwt->mkt', r_g1, F.unfold(F.pad(x.double().cpu(), (1, 1, 1, 1)), 3).view(B, C, 9, H, H).permute(0, 1, 3, 4, 2)).reshape(C, C, 3, 3)
    print(f'B={B} C={C} y1 {H}x{H}: g_pad {g_pad}')
    print(f'   T2 data gradient (g_yb {tuple(g_yb.shape[2:])})     {rel(g_yb, r_yb):.2e}')
    print(f'   adjoint blur alone (generic)           {rel(g1u, r_blur):.2e}   of the product g_yb: {rel(g1u, r_blur_own):.2e}')
    print(f'   blur_gradact                           {rel(g1, r_g1):.2e}   of the product g_yb: {rel(g1, r_blur_own * slope):.2e}   bias partials {rel(gb1, r_gb1):.2e}')
    print(f'   blur + bias_act_bwd (two passes)       {rel(g1b, r_g1):.2e}   bias {rel(gb1b, r_gb1):.2e}')
    print(f'   conv1 weight gradient from g1          {rel(gw1, r_gw1):.2e}')
    d = (g1.double().cpu() - r_g1).abs()
    i = int(d.argmax())
    print(f'   worst element of g1: |diff| {float(d.max()):.3e} at {tuple(int(v) for v in torch.unravel_index(torch.tensor(i), d.shape))}, |g1| there {float(r_g1.flatten()[i].abs()):.3e}, '
          f'rms {float(r_g1.pow(2).mean().sqrt()):.3e};  elements off by > 1e-4 rms: {int((d > 1e-4 * r_g1.pow(2).mean().sqrt()).sum())} of {d.numel()}')


for B, C, H in [(8, 512, 8), (16, 512, 8), (16, 512, 16), (8, 512, 16), (4, 64, 8)]:
    run(B, C, H)
