"""Randomised shapes for the FIR kernels (blur44 / fir_tile / fir_direct behind te_upfirdn2d_f32, and the fused
blur + bias + leaky-ReLU with its one-pass backward) against the CPU oracle: odd and 64n+1 sizes, every (up, down) the model
uses, pads incl. negative, planes that do not fill a block."""
import pytest
import torch

from oracle import te_oracle as O
from transeditor_amd import synth
from transeditor_amd.op import upfirdn2d
from transeditor_amd.op.fir_act import blur_bias_act

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max() / (b.detach().double().abs().max() + 1e-30))


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_upfirdn2d_random_shapes(seed):
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    worst = (0.0, '')
    for i in range(40):
        B, C = ri(1, 3), ri(1, 9)
        H, W = ri(3, 80), ri(3, 140)
        if ri(0, 2) == 0:
            H, W = 64 * ri(1, 2) + 1, 32 * ri(1, 4) + 1            # the transposed convolution's output sizes
        up, down = [(1, 1), (1, 1), (2, 1), (1, 2)][ri(0, 3)]
        taps = [(1, 3, 3, 1), (1, 3, 3, 1), (1, 2, 1), (1, 1)][ri(0, 3)]
        p0, p1 = ri(-1, 3), ri(-1, 3)
        k = O.fir_kernel(taps, float(up * up))
        if (H * up + p0 + p1 - k.shape[0]) // down + 1 <= 0 or (W * up + p0 + p1 - k.shape[1]) // down + 1 <= 0:
            continue
        x = synth.normal((B, C, H, W), f'ff.{seed}.{i}').requires_grad_(True)
        y_ref = O.upfirdn2d(x, k, up, down, (p0, p1))
        gy = synth.normal(tuple(y_ref.shape), f'ffg.{seed}.{i}')
        gx_ref, = torch.autograd.grad((y_ref * gy).sum(), x)
        xd = x.detach().to(DEV).requires_grad_(True)
        y = upfirdn2d(xd, k.to(DEV), up, down, (p0, p1))
        assert tuple(y.shape) == tuple(y_ref.shape)
        gx, = torch.autograd.grad((y * gy.to(DEV)).sum(), xd)
        e = max(rel(y, y_ref), rel(gx, gx_ref))
        if e > worst[0]:
            worst = (e, f'B={B} C={C} {H}x{W} up={up} down={down} taps={taps} pad=({p0},{p1})')
    assert worst[0] < 1e-5, worst


@pytest.mark.parametrize('seed', [0, 1])
def test_blur_bias_act_random_shapes(seed):
    g = torch.Generator().manual_seed(100 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    worst = (0.0, '')
    for i in range(24):
        B, C = ri(1, 3), ri(1, 9)
        H, W = ri(4, 70), ri(4, 135)
        if ri(0, 1):
            H, W = 32 * ri(1, 4) + 1, 64 * ri(1, 2) + 1
        pad = [(1, 1), (2, 2), (2, 1)][ri(0, 2)]
        k = O.fir_kernel((1, 3, 3, 1), 4.0 if pad == (1, 1) else 1.0)
        x = synth.normal((B, C, H, W), f'fb.{seed}.{i}').requires_grad_(True)
        b = (synth.normal((C,), f'fbb.{seed}.{i}') * 0.3).requires_grad_(True)
        y_ref = O.fused_leaky_relu(O.upfirdn2d(x, k, pad=pad), b)
        gy = synth.normal(tuple(y_ref.shape), f'fbg.{seed}.{i}')
        gy = gy * (y_ref.detach().abs() > 1e-3)                     # keep the upstream gradient off the leaky-ReLU kink
        gx_ref, gb_ref = torch.autograd.grad((y_ref * gy).sum(), (x, b))
        xd, bd = x.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
        y = blur_bias_act(xd, k.to(DEV), bd, pad)
        gx, gb = torch.autograd.grad((y * gy.to(DEV)).sum(), (xd, bd))
        e = max(rel(y, y_ref), rel(gx, gx_ref), rel(gb, gb_ref))
        if e > worst[0]:
            worst = (e, f'B={B} C={C} {H}x{W} pad={pad}')
    assert worst[0] < 2e-5, worst
