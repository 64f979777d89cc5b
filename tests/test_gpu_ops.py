"""Parity of every HIP op (called through the C ABI) against the CPU oracle and the committed golden
vectors.  Floating point: the north star allows 1e-3 relative (fp32); the per-op bars here are tighter."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import te_oracle as O
from oracle.gen_golden import UPFIRDN_CASES
from transeditor_amd import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'
OP_TOL = 2e-5      # single op, fp32 round-off class
SUM_TOL = 2e-4     # long reductions (weight gradients over all pixels)


def ops():
    from transeditor_amd import op
    return op


# ------------------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize('name', ['2d', '4d', '3d'])
def test_fused_leaky_relu_golden(golden, name):
    g = golden('fused_leaky_relu')
    x = g[f'{name}.x'].to(DEV).requires_grad_(True)
    b = g[f'{name}.b'].to(DEV).requires_grad_(True)
    wy = g[f'{name}.wy'].to(DEV).requires_grad_(True)
    y = ops().fused_leaky_relu(x, b)
    assert rel_err(y, g[f'{name}.y']) < OP_TOL
    gx, gb = torch.autograd.grad((y * wy).sum(), (x, b), create_graph=True)
    assert rel_err(gx, g[f'{name}.gx']) < OP_TOL and rel_err(gb, g[f'{name}.gb']) < OP_TOL
    ggy, = torch.autograd.grad((gx * g[f'{name}.u'].to(DEV)).sum() + (gb * g[f'{name}.ub'].to(DEV)).sum(), wy)
    assert rel_err(ggy, g[f'{name}.ggy']) < OP_TOL


@pytest.mark.parametrize('shape', [(2, 16, 64, 64), (3, 7, 33, 5), (16, 512), (1, 3, 1024, 1024), (2, 5, 3)])
def test_fused_leaky_relu_vs_oracle(shape):
    x = synth.normal(shape, 'k1.x').requires_grad_(True)
    b = synth.normal((shape[1],), 'k1.b').requires_grad_(True)
    w = synth.normal(shape, 'k1.w')
    y_ref = O.fused_leaky_relu(x, b)
    gx_ref, gb_ref = torch.autograd.grad((y_ref * w).sum(), (x, b))
    xd, bd = x.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    y = ops().fused_leaky_relu(xd, bd)
    assert torch.equal(y.cpu(), y_ref.detach())              # elementwise, no reassociation: bit-exact
    gx, gb = torch.autograd.grad((y * w.to(DEV)).sum(), (xd, bd))
    assert rel_err(gx, gx_ref) < 1e-6
    assert rel_err(gb, gb_ref) < SUM_TOL


def test_fused_leaky_relu_module_and_no_bias():
    m = ops().FusedLeakyReLU(6).to(DEV)
    assert tuple(m.bias.shape) == (6,) and float(m.bias.abs().sum()) == 0
    x = synth.normal((2, 6, 4, 4), 'k1.m')
    assert rel_err(m(x.to(DEV)), O.fused_leaky_relu(x, torch.zeros(6))) < OP_TOL
    m2 = ops().FusedLeakyReLU(6, bias=False).to(DEV)
    assert rel_err(m2(x.to(DEV)), O.fused_leaky_relu(x, None)) < OP_TOL


# ------------------------------------------------------------------------------------------------ K2
@pytest.mark.parametrize('case', UPFIRDN_CASES, ids=[c[0] for c in UPFIRDN_CASES])
def test_upfirdn2d_golden(golden, case):
    g = golden('upfirdn2d')
    name, _, _, _, up, down, pad = case
    x = g[f'{name}.x'].to(DEV).requires_grad_(True)
    y = ops().upfirdn2d(x, g[f'{name}.k'].to(DEV), up, down, pad)
    assert tuple(y.shape) == tuple(g[f'{name}.y'].shape)     # integer index path: exact
    assert rel_err(y, g[f'{name}.y']) < OP_TOL
    gx, = torch.autograd.grad((y * g[f'{name}.wy'].to(DEV)).sum(), x)
    assert rel_err(gx, g[f'{name}.gx']) < OP_TOL


@pytest.mark.parametrize('shape,gain,up,down,pad', [
    ((2, 3, 65, 65), 4.0, 1, 1, (1, 1)), ((2, 5, 129, 129), 4.0, 1, 1, (1, 1)), ((2, 3, 64, 64), 1.0, 1, 1, (2, 2)),
    ((2, 3, 32, 32), 4.0, 2, 1, (2, 1)), ((1, 3, 128, 128), 4.0, 2, 1, (2, 1)), ((2, 3, 66, 66), 4.0, 1, 2, (1, 1)),
    ((1, 2, 257, 257), 4.0, 1, 1, (1, 1)), ((2, 3, 17, 40), 1.0, 1, 2, (2, 2))])
def test_upfirdn2d_tiled_vs_oracle(shape, gain, up, down, pad):
    k = O.fir_kernel((1, 3, 3, 1), gain)
    x = synth.normal(shape, 'k2.x').requires_grad_(True)
    y_ref = O.upfirdn2d(x, k, up, down, pad)
    w = synth.normal(tuple(y_ref.shape), 'k2.w')
    gx_ref, = torch.autograd.grad((y_ref * w).sum(), x)
    xd = x.detach().to(DEV).requires_grad_(True)
    y = ops().upfirdn2d(xd, k.to(DEV), up, down, pad)
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert rel_err(y, y_ref) < OP_TOL
    gx, = torch.autograd.grad((y * w.to(DEV)).sum(), xd, create_graph=True)
    assert rel_err(gx, gx_ref) < OP_TOL
    # second order: the adjoint is linear in its input, so d/dw of <gx, v> is the forward op applied to v
    v = synth.normal(shape, 'k2.v')
    wd = w.to(DEV).requires_grad_(True)
    gx2, = torch.autograd.grad((ops().upfirdn2d(xd, k.to(DEV), up, down, pad) * wd).sum(), xd, create_graph=True)
    gg, = torch.autograd.grad((gx2 * v.to(DEV)).sum(), wd)
    assert rel_err(gg, O.upfirdn2d(v, k, up, down, pad)) < OP_TOL


@pytest.mark.parametrize('shape,pad,taps', [((2, 6, 33, 33), (1, 1), (1, 3, 3, 1)), ((1, 3, 129, 129), (1, 1), (1, 2, 3, 5)),
                                            ((2, 5, 17, 70), (2, 1), (1, 2, 3, 5)), ((3, 2, 40, 200), (2, 2), (4, 3, 2, 1))])
def test_blur_bias_act_fused_vs_oracle(shape, pad, taps):
    """forward (FIR + bias + lrelu epilogue) and the one-pass backward (lrelu gradient in the adjoint FIR's staging,
    bias gradient from per-tile partial sums), multi-tile planes and asymmetric taps included."""
    from transeditor_amd.op.fir_act import blur_bias_act
    k = O.fir_kernel(taps, 4.0)
    x = synth.normal(shape, 'fa.x').requires_grad_(True)
    b = synth.normal((shape[1],), 'fa.b').requires_grad_(True)
    y_ref = O.fused_leaky_relu(O.upfirdn2d(x, k, pad=pad), b)
    w = synth.normal(tuple(y_ref.shape), 'fa.w')
    gx_ref, gb_ref = torch.autograd.grad((y_ref * w).sum(), (x, b))
    xd, bd = x.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    y = blur_bias_act(xd, k.to(DEV), bd, pad)
    assert rel_err(y, y_ref) < OP_TOL
    gx, gb = torch.autograd.grad((y * w.to(DEV)).sum(), (xd, bd))
    assert rel_err(gx, gx_ref) < OP_TOL and rel_err(gb, gb_ref) < SUM_TOL


# ------------------------------------------------------------------------------------------------ F1 kernels
def _ref_conv(kind, x, w):
    if kind == '3x3':
        return F.conv2d(x, w, padding=1)
    if kind == '1x1':
        return F.conv2d(x, w)
    if kind == 'up':
        return F.conv_transpose2d(x, w.transpose(0, 1), stride=2)
    if kind == 'down':
        return F.conv2d(x, w, stride=2)
    raise ValueError(kind)


def _in_hw(kind, H, W):
    return (2 * H + 1, 2 * W + 1) if kind == 'down' else (H, W)


CONV_SHAPES = [  # B, K(ci), M(co), H, W
    (3, 6, 5, 7, 7), (2, 40, 36, 12, 12), (2, 136, 130, 33, 20), (16, 128, 128, 4, 4), (4, 64, 256, 8, 8),
    (2, 32, 64, 16, 16), (1, 24, 16, 64, 64), (2, 8, 8, 5, 3), (1, 16, 32, 40, 72),
    (2, 40, 128, 17, 31), (1, 16, 96, 64, 32), (3, 24, 200, 18, 65),
    # FAST kernels (K % 16 == 0, one sample per tile) on widths whose last tile straddles the image border inside a 16-byte
    # segment (block-uniform mask path), narrow / wide M, and enough tiles for the deep-stage 1x1 class (K % 64 == 0)
    (1, 16, 128, 24, 42), (2, 32, 48, 20, 38), (16, 64, 256, 32, 42)]


@pytest.mark.parametrize('kind', ['3x3', '1x1', 'up', 'down'])
@pytest.mark.parametrize('shape', CONV_SHAPES, ids=['x'.join(map(str, s)) for s in CONV_SHAPES])
def test_conv_trio_vs_torch(kind, shape):
    """forward, data gradient and weight gradient of the plain convolution (closed autograd trio)."""
    from transeditor_amd.op.modconv import conv_core
    B, K, M, H, W = shape
    ks = 1 if kind == '1x1' else 3
    x = synth.normal((B, K, *_in_hw(kind, H, W)), f'cv.x.{kind}').requires_grad_(True)
    w = (synth.normal((M, K, ks, ks), f'cv.w.{kind}') / math.sqrt(K * ks * ks)).requires_grad_(True)
    y_ref = _ref_conv(kind, x, w)
    gy = synth.normal(tuple(y_ref.shape), f'cv.g.{kind}')
    gx_ref, gw_ref = torch.autograd.grad((y_ref * gy).sum(), (x, w))
    xd, wd = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
    y = conv_core(xd, wd, kind)
    assert tuple(y.shape) == tuple(y_ref.shape)
    assert rel_err(y, y_ref) < OP_TOL, 'forward'
    gx, gw = torch.autograd.grad((y * gy.to(DEV)).sum(), (xd, wd))
    assert rel_err(gx, gx_ref) < OP_TOL, 'dgrad'
    assert rel_err(gw, gw_ref) < SUM_TOL, 'wgrad'


@pytest.mark.parametrize('kind', ['3x3', 'up', 'down'])
def test_conv_trio_second_order(kind):
    """grad-of-grad through the trio (what the path-length / R1 regularisers need) vs torch autograd on CPU."""
    from transeditor_amd.op.modconv import conv_core
    B, K, M, H, W = 2, 12, 10, 9, 9
    x = synth.normal((B, K, *_in_hw(kind, H, W)), 'cv2.x').requires_grad_(True)
    w = (synth.normal((M, K, 3, 3), 'cv2.w') / 10).requires_grad_(True)
    s = synth.normal((B, K), 'cv2.s').requires_grad_(True)

    def f(conv, x, w, s, dev):
        y = conv(x * s[:, :, None, None], w)
        gy = synth.normal(tuple(y.shape), 'cv2.g').to(dev)
        gs, = torch.autograd.grad((y * gy).sum(), s, create_graph=True)
        return torch.autograd.grad(gs.pow(2).sum(), (x, w))

    ref = f(lambda a, b: _ref_conv(kind, a, b), x, w, s, 'cpu')
    xd, wd, sd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, s))
    got = f(lambda a, b: conv_core(a, b, kind), xd, wd, sd, DEV)
    assert rel_err(got[0], ref[0]) < SUM_TOL and rel_err(got[1], ref[1]) < SUM_TOL


@pytest.mark.parametrize('kind', ['3x3', '1x1', 'up'])
@pytest.mark.parametrize('shape', [(3, 6, 5, 7, 7), (2, 72, 136, 20, 33), (16, 128, 64, 4, 4), (2, 32, 32, 64, 64),
                                   (2, 32, 64, 16, 16), (2, 64, 48, 16, 16)])
@pytest.mark.parametrize('act', [False, True])
def test_modconv_fused_kernel_vs_composite(kind, shape, act):
    """fused kernel (scales + bias + lrelu in the prologue/epilogue, slab-based backward) vs the plain math."""
    from transeditor_amd.op.modconv import modconv
    B, K, M, H, W = shape
    ks = 1 if kind == '1x1' else 3
    x = synth.normal((B, K, H, W), 'mf.x').requires_grad_(True)
    w = (synth.normal((M, K, ks, ks), 'mf.w') / math.sqrt(K * ks * ks)).requires_grad_(True)
    isc = (1 + 0.5 * synth.normal((B, K), 'mf.i')).requires_grad_(True)
    osc = (1 + 0.3 * synth.normal((B, M), 'mf.o')).abs().add(0.1).requires_grad_(True)
    bias = synth.normal((M,), 'mf.b').requires_grad_(True)
    if kind == 'up' and act:
        pytest.skip('upsampling layers fuse the activation into the blur kernel instead')
    ws = 1.0 if B == 3 else 1.7           # equalised-lr constant applied inside the kernels (pack / reducer)
    y_ref = _ref_conv(kind, x * isc[:, :, None, None], w * ws) * osc[:, :, None, None] + bias[None, :, None, None]
    if act:
        y_ref = F.leaky_relu(y_ref, 0.2) * math.sqrt(2)
    gy = synth.normal(tuple(y_ref.shape), 'mf.g')
    ref = torch.autograd.grad((y_ref * gy).sum(), (x, w, isc, osc, bias))
    d = [t.detach().to(DEV).requires_grad_(True) for t in (x, w, isc, osc, bias)]
    y = modconv(d[0], d[1], d[2], d[3], d[4], act, kind, ws)
    assert rel_err(y, y_ref) < OP_TOL
    got = torch.autograd.grad((y * gy.to(DEV)).sum(), d)
    for name, a, b in zip(('gx', 'gw', 'gisc', 'gosc', 'gbias'), got, ref):
        assert rel_err(a, b) < SUM_TOL, name


@pytest.mark.parametrize('shape', [(2, 128, 16, 16), (3, 40, 6, 6), (2, 512, 4, 4), (2, 24, 5, 5), (2, 128, 64, 64), (2, 32, 96, 96)])
def test_torgb_streaming_kernels(shape):
    """ToRGB path (1x1 -> 3 channels, style scale + bias): dedicated streaming kernels when H*W % 4 == 0 and K <= 512,
    MFMA 1x1 kernel otherwise; forward, dx, dW, ds, dbias against plain torch."""
    from transeditor_amd.op.modconv import modconv
    B, K, H, W = shape
    x = synth.normal((B, K, H, W), 'rgb.x').requires_grad_(True)
    w = (synth.normal((3, K, 1, 1), 'rgb.w') / math.sqrt(K)).requires_grad_(True)
    isc = (1 + 0.5 * synth.normal((B, K), 'rgb.i')).requires_grad_(True)
    bias = synth.normal((3,), 'rgb.b').requires_grad_(True)
    y_ref = F.conv2d(x * isc[:, :, None, None], w * 0.6) + bias[None, :, None, None]
    gy = synth.normal(tuple(y_ref.shape), 'rgb.g')
    ref = torch.autograd.grad((y_ref * gy).sum(), (x, w, isc, bias))
    d = [t.detach().to(DEV).requires_grad_(True) for t in (x, w, isc, bias)]
    y = modconv(d[0], d[1], d[2], None, d[3], False, '1x1', 0.6)
    assert rel_err(y, y_ref) < OP_TOL
    got = torch.autograd.grad((y * gy.to(DEV)).sum(), d)
    for name, a, b in zip(('gx', 'gw', 'gisc', 'gbias'), got, ref):
        assert rel_err(a, b) < SUM_TOL, name


@pytest.mark.parametrize('kind,shape', [('3x3', (2, 24, 40, 9, 9)), ('1x1', (2, 3, 130, 16, 16)), ('down', (2, 136, 72, 8, 8)),
                                        ('down', (4, 32, 32, 2, 2)), ('3x3', (4, 513, 512, 4, 4))])
def test_discriminator_conv_kinds_fused_bias_act(kind, shape):
    """EqualConv2d + FusedLeakyReLU as one fused launch (no modulation), incl. R1-style double backward wrt the input."""
    from transeditor_amd.op.modconv import modconv
    B, K, M, H, W = shape
    ks = 1 if kind == '1x1' else 3
    x = synth.normal((B, K, *_in_hw(kind, H, W)), 'dc.x').requires_grad_(True)
    w = (synth.normal((M, K, ks, ks), 'dc.w') / math.sqrt(K * ks * ks)).requires_grad_(True)
    bias = synth.normal((M,), 'dc.b').requires_grad_(True)

    def run(conv, x, w, bias, dev):
        y = conv(x, w, bias)
        gy = synth.normal(tuple(y.shape), 'dc.g').to(dev)
        gx, = torch.autograd.grad((y * gy).sum(), x, create_graph=True)
        r1 = gx.pow(2).sum()
        return y, gx, torch.autograd.grad(r1, (w, bias), retain_graph=True), torch.autograd.grad((y * gy).sum(), (w, bias))

    ref = run(lambda a, b, c: F.leaky_relu(_ref_conv(kind, a, b * 1.3) + c[None, :, None, None], 0.2) * math.sqrt(2), x, w, bias, 'cpu')
    d = [t.detach().to(DEV).requires_grad_(True) for t in (x, w, bias)]
    got = run(lambda a, b, c: modconv(a, b, None, None, c, True, kind, 1.3), d[0], d[1], d[2], DEV)
    assert rel_err(got[0], ref[0]) < OP_TOL and rel_err(got[1], ref[1]) < OP_TOL
    for a, b in zip(got[2] + got[3], ref[2] + ref[3]):
        assert rel_err(a, b) < 5e-4


@pytest.mark.parametrize('shape', [(16, 512, 512, 9), (3, 40, 24, 9), (2, 3, 130, 1), (64, 128, 256, 9)])
def test_demod_kernels(shape):
    """d = rsqrt(sum (wscale w s)^2 + eps) (model_spatial_query.py:300-304): forward, gw / gs, and the recorded backward."""
    from transeditor_amd.op.style import demod
    from plain_torch import demod as _torch_expr
    B, Co, Ci, T = shape
    k = 3 if T == 9 else 1
    w = synth.normal((Co, Ci, k, k), 'dm.w').requires_grad_(True)
    s = (1 + 0.5 * synth.normal((B, Ci), 'dm.s')).requires_grad_(True)
    ws = 1 / math.sqrt(Ci * T)
    gd = synth.normal((B, Co), 'dm.g')

    def run(f, w, s, gd):
        d = f(w, s, ws, 1e-8)
        gw, gs = torch.autograd.grad(d, (w, s), gd, create_graph=True)
        return d, gw, gs, torch.autograd.grad(gs.pow(2).sum() + gw.pow(2).sum(), (w, s))

    ref = run(_torch_expr, w.double(), s.double(), gd.double())
    wd, sd = (t.detach().to(DEV).requires_grad_(True) for t in (w, s))
    d = demod(wd, sd, ws, 1e-8)
    assert rel_err(d, ref[0].float()) < 1e-5
    gw, gs = torch.autograd.grad(d, (wd, sd), gd.to(DEV))
    assert rel_err(gw, ref[1].float()) < 2e-5 and rel_err(gs, ref[2].float()) < 2e-5
    got = run(demod, wd, sd, gd.to(DEV))
    assert rel_err(got[3][0], ref[3][0].float()) < 1e-4 and rel_err(got[3][1], ref[3][1].float()) < 1e-4


@pytest.mark.parametrize('kind', ['3x3', 'down'])
def test_conv_fused_activation_gain_one(kind):
    """act code 4: leaky-ReLU with gain 1 in the conv epilogue (ResBlock folds its 1/sqrt(2) into the branches)."""
    from transeditor_amd.op.modconv import modconv
    B, K, M, H, W = 2, 24, 40, 8, 8
    x = synth.normal((B, K, *_in_hw(kind, H, W)), 'g1.x').requires_grad_(True)
    w = (synth.normal((M, K, 3, 3), 'g1.w') / math.sqrt(K * 9)).requires_grad_(True)
    bias = synth.normal((M,), 'g1.b').requires_grad_(True)
    y_ref = F.leaky_relu(_ref_conv(kind, x, w * 0.8) + bias[None, :, None, None], 0.2)
    gy = synth.normal(tuple(y_ref.shape), 'g1.g')
    ref = torch.autograd.grad((y_ref * gy).sum(), (x, w, bias))
    d = [t.detach().to(DEV).requires_grad_(True) for t in (x, w, bias)]
    y = modconv(d[0], d[1], None, None, d[2], 1.0, kind, 0.8)
    assert rel_err(y, y_ref) < OP_TOL
    for a, b in zip(torch.autograd.grad((y * gy.to(DEV)).sum(), d), ref):
        assert rel_err(a, b) < SUM_TOL


# ------------------------------------------------------------------------------------------------ F1 module
@pytest.mark.parametrize('name', ['plain3', 'up3', 'rgb1', 'plain3_wide', 'up3_wide'])
def test_modulated_conv2d_module_golden(golden, name):
    """ModulatedConv2d (drop-in class) vs the reference's outputs, first grads and the second-order term."""
    from transeditor_amd.model_spatial_query import ModulatedConv2d
    g = golden('modulated_conv2d')
    demod, upsmp = bool(g[f'{name}.cfg'][0]), bool(g[f'{name}.cfg'][1])
    w = g[f'{name}.weight']
    m = ModulatedConv2d(w.shape[2], w.shape[1], w.shape[3], 16, demodulate=demod, upsample=upsmp).to(DEV)
    with torch.no_grad():
        m.weight.copy_(w)
        m.modulation.weight.copy_(g[f'{name}.mod_w'])
        m.modulation.bias.copy_(g[f'{name}.mod_b'])
    x = g[f'{name}.x'].to(DEV).requires_grad_(True)
    s = g[f'{name}.s'].to(DEV).requires_grad_(True)
    y = m(x, s)
    assert rel_err(y, g[f'{name}.y']) < 1e-4
    params = [m.weight, m.modulation.weight, m.modulation.bias]
    gr = torch.autograd.grad((y * g[f'{name}.wy'].to(DEV)).sum(), [x, s] + params, create_graph=True)
    for got, key in zip(gr, ('gx', 'gs', 'gw', 'gmw', 'gmb')):
        assert rel_err(got, g[f'{name}.{key}']) < 2e-4, key
    pl = gr[1].pow(2).sum()
    assert abs(float(pl) - float(g[f'{name}.pl'])) / float(g[f'{name}.pl']) < 2e-4
    g2 = torch.autograd.grad(pl, [x] + params, allow_unused=True)
    for got, key in zip(g2, ('pl_gx', 'pl_gw', 'pl_gmw', 'pl_gmb')):
        want = g[f'{name}.{key}']
        if float(want.abs().max()) == 0:
            # analytically zero (without demodulation y is linear in the style scale, so |dy/ds|^2 does not depend on it): the closed
            # family forms the scale gradient as a channel dot divided by the scale, whose s-derivative is two terms of size
            # ~2 pl / s that cancel to round-off
            assert got is None or float(got.abs().max()) < 1e-6 * max(1.0, float(pl))
        else:
            assert rel_err(got, want) < 5e-4, key


@pytest.mark.parametrize('name', ['down3', 'down3_wide', 'down1', 'down3_nodemod'])
def test_modulated_conv2d_downsample_golden(golden, name):
    """ModulatedConv2d(downsample=True) (model_spatial_query.py:270-276, 323-329) vs the reference's outputs, first grads
    and the second-order term: blur + the strided kernel with style scale / demodulation."""
    from transeditor_amd.model_spatial_query import ModulatedConv2d
    g = golden('modulated_conv2d_down')
    demod, k = bool(g[f'{name}.cfg'][0]), int(g[f'{name}.cfg'][1])
    w = g[f'{name}.weight']
    m = ModulatedConv2d(w.shape[2], w.shape[1], k, 16, demodulate=demod, downsample=True).to(DEV)
    with torch.no_grad():
        m.weight.copy_(w)
        m.modulation.weight.copy_(g[f'{name}.mod_w'])
        m.modulation.bias.copy_(g[f'{name}.mod_b'])
    x = g[f'{name}.x'].to(DEV).requires_grad_(True)
    s = g[f'{name}.s'].to(DEV).requires_grad_(True)
    y = m(x, s)
    assert y.shape == g[f'{name}.y'].shape
    assert rel_err(y, g[f'{name}.y']) < 1e-4
    params = [m.weight, m.modulation.weight, m.modulation.bias]
    gr = torch.autograd.grad((y * g[f'{name}.wy'].to(DEV)).sum(), [x, s] + params, create_graph=True)
    for got, key in zip(gr, ('gx', 'gs', 'gw', 'gmw', 'gmb')):
        assert rel_err(got, g[f'{name}.{key}']) < 2e-4, key
    pl = gr[1].pow(2).sum()
    assert abs(float(pl) - float(g[f'{name}.pl'])) / float(g[f'{name}.pl']) < 2e-4
    g2 = torch.autograd.grad(pl, [x] + params, allow_unused=True)
    for got, key in zip(g2, ('pl_gx', 'pl_gw', 'pl_gmw', 'pl_gmb')):
        want = g[f'{name}.{key}']
        if float(want.abs().max()) == 0:
            assert got is None or float(got.abs().max()) < 1e-6
        else:
            assert rel_err(got, want) < 5e-4, key
    # first-order backward through the fused node (no create_graph): same gradients
    y2 = m(x, s)
    gr1 = torch.autograd.grad((y2 * g[f'{name}.wy'].to(DEV)).sum(), [x, s] + params)
    for got, key in zip(gr1, ('gx', 'gs', 'gw', 'gmw', 'gmb')):
        assert rel_err(got, g[f'{name}.{key}']) < 2e-4, key


# ------------------------------------------------------------------------------------------------ F2
@pytest.mark.parametrize('N,M,L,C', [(5, 16, 16, 128), (16, 16, 16, 128), (1, 16, 16, 128)])
def test_attention_core_vs_torch(N, M, L, C):
    """attention core (model_spatial_query.py:888-894) against the einsum / softmax restatement (tests/plain_torch.py):
    values, first-order gradients (kernel), and the recorded backward (closed batched-GEMM family of op/linear.py::_Bmm +
    framework softmax) through a second differentiation w.r.t. q, k and v."""
    from transeditor_amd.op.attention import attention_core
    from plain_torch import attention_core as plain
    q = synth.normal((N, M, C), 'at.q').requires_grad_(True)
    k, v = (synth.normal((N, L, C), f'at.{n}').requires_grad_(True) for n in 'kv')
    scale = C ** -0.5 * 3.0
    o_ref, sim_ref = plain(q.double(), k.double(), v.double(), scale, 4)
    go, gs = synth.normal((N, M, C), 'at.go'), synth.normal((N, 4, M, L), 'at.gs')
    ref = torch.autograd.grad([o_ref, sim_ref], (q, k, v), [go.double(), gs.double()])
    d = [t.detach().to(DEV).requires_grad_(True) for t in (q, k, v)]
    o, sim = attention_core(d[0], d[1], d[2], scale, 4)
    assert rel_err(o, o_ref.float()) < OP_TOL and rel_err(sim, sim_ref.float()) < OP_TOL
    got = torch.autograd.grad([o, sim], d, [go.to(DEV), gs.to(DEV)])
    for a, b in zip(got, ref):
        assert rel_err(a, b.float()) < 1e-4

    def second(f, qkv, go, gs):
        o, sim = f(*qkv, scale, 4)
        g1 = torch.autograd.grad([o, sim], qkv, [go, gs], create_graph=True)
        return torch.autograd.grad(sum(t.square().sum() for t in g1), qkv)
    want = second(plain, [t.detach().double().requires_grad_(True) for t in (q, k, v)], go.double(), gs.double())
    got = second(attention_core, d, go.to(DEV), gs.to(DEV))
    for a, b in zip(got, want):
        assert rel_err(a, b.float()) < 2e-4


@pytest.mark.parametrize('name,cin', [('b0_528', 528), ('b_512', 512)])
def test_attention_block_golden(golden, name, cin):
    from test_oracle_golden import attention_block_params
    from transeditor_amd.model_spatial_query import AttentionBlock
    g = golden('attention_block')
    m = AttentionBlock(cin, cin, 512, lr_mul=0.01).to(DEV)
    m.load_state_dict(attention_block_params(name, cin, cin, DEV))
    x = g[f'{name}.x'].to(DEV).requires_grad_(True)
    p = g[f'{name}.p'].to(DEV).requires_grad_(True)
    y, sim = m(x, p, return_similarity=True)
    assert rel_err(y, g[f'{name}.y']) < 1e-4 and rel_err(sim, g[f'{name}.sim']) < 1e-4
    gx, gp = torch.autograd.grad((y * g[f'{name}.wy'].to(DEV)).sum(), (x, p))
    assert rel_err(gx, g[f'{name}.gx']) < 5e-4 and rel_err(gp, g[f'{name}.gp']) < 5e-4


# ------------------------------------------------------------------------------------------------ G2/A2 small GEMM
LINEAR_CASES = [   # rows-shape, K, N, act, residual, bias
    ((16, 16), 512, 128, None, False, True),       # attention q/k/v
    ((16, 16), 128, 512, None, True, True),        # attention proj + skip
    ((16, 16), 512, 512, 'gelu', False, True),     # MLP up
    ((16,), 512, 256, None, False, True),          # modulation (row slice of the latent)
    ((4, 512), 16, 14, None, False, True),         # adjust_style: K = 16, N = 14 (ragged tile)
    ((5,), 512, 1, None, False, True),             # discriminator's last layer (N = 1)
    ((7,), 96, 40, 'lrelu', False, True),          # fused_lrelu mapping-style layer, ragged everything
    ((33,), 36, 65, None, False, False),           # no bias
    ((32,), 8192, 512, 'lrelu', False, True),      # discriminator final_linear[0] (:831-834): split-K form, 8 chunks
    ((5,), 2304, 40, None, False, True),           # split-K with a ragged tile (3 chunks of 768)
    ((16, 512), 16, 14, None, False, True),        # adjust_style at batch 16: dW reduces over 8192 rows (split-K form)
    ((2055,), 24, 20, None, False, True),          # dW over a row count that does not split evenly: 2 chunks of 1024 + a 7-row tail
]


@pytest.mark.gpu
@pytest.mark.parametrize('rows,K,N,act,use_res,use_bias', LINEAR_CASES)
def test_linear_fused_matches_torch(rows, K, N, act, use_res, use_bias):
    from transeditor_amd.op.linear import linear_fused
    from plain_torch import equal_linear as _torch_expr
    g = torch.Generator().manual_seed(K * 7 + N)
    dev = 'cuda'
    x = torch.randn(*rows, K, generator=g).to(dev).requires_grad_(True)
    w = torch.randn(N, K, generator=g).to(dev).requires_grad_(True)
    b = torch.randn(N, generator=g).to(dev).requires_grad_(True) if use_bias else None
    r = torch.randn(*rows, N, generator=g).to(dev).requires_grad_(True) if use_res else None
    alpha, beta = 1 / math.sqrt(K), 0.7
    ins = [t for t in (x, w, b, r) if t is not None]
    y = linear_fused(x, w, b, alpha, beta, act, r)
    y_ref = _torch_expr(x.double(), w.double(), None if b is None else b.double(), alpha, beta, act,
                        None if r is None else r.double())
    assert y.shape == y_ref.shape
    R = x.numel() // K
    grow = max(1.0, math.sqrt(max(K, R) / 512))        # fp32 round-off of the longest reduction (K forward, rows for dW)
    assert rel_err(y, y_ref.float()) < 1e-5 * grow
    gy = torch.randn(y.shape, generator=g).to(dev)
    got = torch.autograd.grad(y, ins, gy)
    want = torch.autograd.grad(y_ref, ins, gy.double())
    for a_, b_ in zip(got, want):
        assert rel_err(a_, b_.float()) < 2e-5 * grow
    # recorded backward (path-length regulariser route): second derivative through the torch expression
    gx, = torch.autograd.grad(linear_fused(x, w, b, alpha, beta, act, r), x, gy, create_graph=True)
    gx_ref, = torch.autograd.grad(_torch_expr(x, w, b, alpha, beta, act, r), x, gy, create_graph=True)
    got2 = torch.autograd.grad(gx.square().sum(), w)
    want2 = torch.autograd.grad(gx_ref.square().sum(), w)
    assert rel_err(got2[0], want2[0]) < 1e-4


@pytest.mark.gpu
def test_linear_fused_strided_rows():
    from transeditor_amd.op.linear import linear_fused
    from plain_torch import equal_linear as _torch_expr
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(16, 14, 512, generator=g).cuda().requires_grad_(True)
    w = torch.randn(64, 512, generator=g).cuda().requires_grad_(True)
    b = torch.randn(64, generator=g).cuda()
    y = linear_fused(lat[:, 5], w, b, 0.04, 1.0)
    y_ref = _torch_expr(lat[:, 5], w, b, 0.04, 1.0)
    assert rel_err(y, y_ref) < 1e-5
    gy = torch.randn(16, 64, generator=g).cuda()
    for a_, b_ in zip(torch.autograd.grad(y, (lat, w), gy), torch.autograd.grad(y_ref, (lat, w), gy)):
        assert rel_err(a_, b_) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('B,C,H', [(16, 512, 4), (32, 512, 4), (4, 24, 3), (3, 8, 2), (8, 16, 4)])
def test_minibatch_stddev_kernel(B, C, H):
    """D1: minibatch stddev + concat as one launch vs the reference formula (model_spatial_query.py:844-852): forward,
    backward, and the recorded (R1) backward through the torch expression."""
    from transeditor_amd.op.stddev import minibatch_stddev
    from plain_torch import minibatch_stddev as _torch_expr
    x = synth.normal((B, C, H, H), f'sd.x.{B}').to(DEV).requires_grad_(True)
    group = min(B, 4)
    y = minibatch_stddev(x, 4)
    y_ref = _torch_expr(x.double(), group)
    assert tuple(y.shape) == (B, C + 1, H, H) and rel_err(y, y_ref) < 1e-6
    assert torch.equal(y[:, :C], x)
    gy = synth.normal(tuple(y.shape), f'sd.g.{B}').to(DEV)
    gx, = torch.autograd.grad(y, x, gy)
    gx_ref, = torch.autograd.grad(y_ref, x, gy.double())
    assert rel_err(gx, gx_ref) < 1e-5
    g1, = torch.autograd.grad(minibatch_stddev(x, 4), x, gy, create_graph=True)
    g1r, = torch.autograd.grad(_torch_expr(x, group), x, gy, create_graph=True)
    assert rel_err(torch.autograd.grad(g1.square().sum(), x)[0], torch.autograd.grad(g1r.square().sum(), x)[0]) < 1e-4


# ------------------------------------------------------------------------------------------------ G2 token-wise mapping
@pytest.mark.gpu
@pytest.mark.parametrize('B,D,Cn,T', [(16, 512, 16, 16), (3, 96, 8, 5), (33, 64, 16, 16)])
def test_token_mlp_batched_launch(B, D, Cn, T):
    """one launch for the T per-token EqualLinear + fused lrelu layers (model_spatial_query.py:626-646): values, every
    gradient and the recorded backward against the per-token F.linear restatement (tests/plain_torch.py) in fp64 / fp32."""
    from transeditor_amd.op.token_mlp import token_mlp
    from plain_torch import token_mlp as _torch_expr
    g = torch.Generator().manual_seed(B * 100 + D)
    x = torch.randn(B, D, Cn, generator=g).cuda().requires_grad_(True)
    ws = [(torch.randn(D, D, generator=g) / 0.01).cuda().requires_grad_(True) for _ in range(T)]
    bs = [torch.randn(D, generator=g).cuda().requires_grad_(True) for _ in range(T)]
    scale, lr_mul = 0.01 / math.sqrt(D), 0.01
    y = token_mlp(x, ws, bs, scale, lr_mul)
    ref = torch.stack([F.leaky_relu(F.linear(x[:, :, t].double(), ws[t].double() * scale, bs[t].double() * lr_mul), 0.2)
                       * math.sqrt(2) for t in range(T)], 1)
    assert tuple(y.shape) == (B, T, D) and rel_err(y, ref.float()) < 1e-5
    gy = torch.randn(B, T, D, generator=g).cuda()
    got = torch.autograd.grad(y, [x] + ws + bs, gy)
    want = torch.autograd.grad(ref, [x] + ws + bs, gy.double())
    for a_, b_ in zip(got, want):
        assert rel_err(a_, b_.float()) < 2e-5
    gx, = torch.autograd.grad(token_mlp(x, ws, bs, scale, lr_mul), x, gy, create_graph=True)
    gx_ref, = torch.autograd.grad(_torch_expr(x, ws, bs, scale, lr_mul), x, gy, create_graph=True)
    a_, = torch.autograd.grad(gx.square().sum(), ws[0])
    b_, = torch.autograd.grad(gx_ref.square().sum(), ws[0])
    assert rel_err(a_, b_) < 1e-4


# ------------------------------------------------------------------------------------------------ A2 sample-wise layer norm
@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(16, 16, 512), (16, 16, 528), (3, 5, 12), (2, 4, 4096)])
def test_sample_layer_norm_kernels(shape):
    """F.layer_norm(x, x.size()[1:]) of the attention blocks (model_spatial_query.py:924/931): forward, backward and the
    recorded backward against torch (fp64 for the first-order quantities)."""
    from transeditor_amd.op.layernorm import sample_layer_norm
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * 2 + 0.3).cuda().requires_grad_(True)
    gy = torch.randn(*shape, generator=g).cuda()
    y = sample_layer_norm(x)
    ref = F.layer_norm(x.double(), shape[1:], eps=1e-5)
    assert rel_err(y, ref.float()) < 1e-5
    gx, = torch.autograd.grad(y, x, gy)
    gx_ref, = torch.autograd.grad(ref, x, gy.double())
    assert rel_err(gx, gx_ref.float()) < 2e-5
    a_, = torch.autograd.grad(sample_layer_norm(x), x, gy, create_graph=True)
    b_, = torch.autograd.grad(F.layer_norm(x, shape[1:], eps=1e-5), x, gy, create_graph=True)
    ga, = torch.autograd.grad(a_.square().sum(), x)
    gb, = torch.autograd.grad(b_.square().sum(), x)
    assert rel_err(ga, gb) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('shape,dim', [((16, 512, 16), 1), ((3, 40, 8), 1), ((4, 512, 16), 2), ((2, 7, 5), 1)])
def test_pixel_norm_kernels(shape, dim):
    """PixelNorm (model_spatial_query.py:80-81): kernel path for [B, D, C] / dim 1, torch expression otherwise."""
    from transeditor_amd.op.layernorm import pixel_norm
    from plain_torch import pixel_norm as _pixel_norm_expr
    g = torch.Generator().manual_seed(sum(shape) + dim)
    x = torch.randn(*shape, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(*shape, generator=g).cuda()
    y = pixel_norm(x, dim)
    ref = _pixel_norm_expr(x.double(), dim)
    assert rel_err(y, ref.float()) < 1e-5
    gx, = torch.autograd.grad(y, x, gy)
    gx_ref, = torch.autograd.grad(ref, x, gy.double())
    assert rel_err(gx, gx_ref.float()) < 2e-5
    a_, = torch.autograd.grad(pixel_norm(x, dim), x, gy, create_graph=True)
    b_, = torch.autograd.grad(_pixel_norm_expr(x, dim), x, gy, create_graph=True)
    assert rel_err(torch.autograd.grad(a_.square().sum(), x)[0], torch.autograd.grad(b_.square().sum(), x)[0]) < 1e-4


# ------------------------------------------------------------------------------------------------ K1 / K2 in half and double
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-13), (torch.float16, 3e-3)])
def test_fused_leaky_relu_other_dtypes(dtype, tol):
    """The reference dispatches fused_bias_act over half / float / double (fused_bias_act_kernel.cu:79) with its float alpha /
    scale converted to scalar_t: forward, gradient and gradient of the gradient through te_bias_act_f16 / _f64."""
    import numpy as np
    from transeditor_amd.op import fused_leaky_relu
    alpha, scale = float(np.float32(0.2)), float(np.float32(2 ** 0.5))          # what a C float argument holds
    x = synth.normal((3, 5, 6, 7), 'k1o.x').double()
    b = (synth.normal((5,), 'k1o.b') * 0.3).double()
    if dtype == torch.float16:
        x, b = x.half().double(), b.half().double()                              # representable inputs
    x.requires_grad_(True); b.requires_grad_(True)
    y_ref = O.fused_leaky_relu(x, b, alpha, scale)
    gy = synth.normal(tuple(y_ref.shape), 'k1o.g').double()
    gx_ref, gb_ref = torch.autograd.grad((y_ref * gy).sum(), (x, b), create_graph=True)
    u = synth.normal(tuple(x.shape), 'k1o.u').double()
    xd = x.detach().to(DEV, dtype).requires_grad_(True)
    bd = b.detach().to(DEV, dtype).requires_grad_(True)
    gyd = gy.to(DEV, dtype).requires_grad_(True)
    y = fused_leaky_relu(xd, bd, 0.2, 2 ** 0.5)
    assert y.dtype == dtype
    assert rel_err(y.double(), y_ref) < tol
    gx, gb = torch.autograd.grad((y * gyd).sum(), (xd, bd), create_graph=True)
    assert rel_err(gx.double(), gx_ref) < tol and rel_err(gb.double(), gb_ref) < 10 * tol
    ggy, = torch.autograd.grad((gx * u.to(DEV, dtype)).sum(), gyd)               # grad-grad path (mode 31 on the saved output)
    gy_leaf = gy.clone().requires_grad_(True)
    gx2, = torch.autograd.grad((O.fused_leaky_relu(x, b, alpha, scale) * gy_leaf).sum(), x, create_graph=True)
    ggy_ref, = torch.autograd.grad((gx2 * u).sum(), gy_leaf)
    assert rel_err(ggy.double(), ggy_ref) < tol


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-13), (torch.float16, 3e-3)])
@pytest.mark.parametrize('up,down,pad,taps', [(1, 1, (2, 1), (1, 3, 3, 1)), (2, 1, (2, 1), (1, 3, 3, 1)), (1, 2, (1, 1), (1, 3, 3, 1)),
                                               (3, 2, (3, 2), (1, 2, 3, 2, 1))])
def test_upfirdn2d_other_dtypes(dtype, tol, up, down, pad, taps):
    """te_upfirdn2d_f16 / _f64 (upfirdn2d_kernel.cu:57-58 dispatches over both), forward and adjoint."""
    from transeditor_amd.op import upfirdn2d
    k = O.fir_kernel(taps, float(up * up)).double()
    x = synth.normal((2, 3, 9, 11), 'k2o.x').double()
    if dtype == torch.float16:
        x, k = x.half().double(), k.half().double()
    x.requires_grad_(True)
    y_ref = O.upfirdn2d(x, k, up, down, pad)
    gy = synth.normal(tuple(y_ref.shape), 'k2o.g').double()
    gx_ref, = torch.autograd.grad((y_ref * gy).sum(), x)
    xd = x.detach().to(DEV, dtype).requires_grad_(True)
    y = upfirdn2d(xd, k.to(DEV, dtype), up, down, pad)
    assert y.dtype == dtype and tuple(y.shape) == tuple(y_ref.shape)
    assert rel_err(y.double(), y_ref) < tol
    gx, = torch.autograd.grad((y * gy.to(DEV, dtype)).sum(), xd)
    assert rel_err(gx.double(), gx_ref) < tol


# ------------------------------------------------------------------------------------------------ channel scale / dot pair
@pytest.mark.parametrize('shape', [(2, 5, 7, 9), (3, 16, 8, 8), (2, 4, 64, 64), (1, 3, 1, 1), (2, 130, 33, 20), (1, 5, 33, 33),
                                   (2, 6, 65, 65), (1, 2, 129, 257)])
def test_chan_scale_pair_any_order(shape):
    """x * s[:, :, None, None] on te_chan_scale / te_chan_dot: values, first gradients and the gradient of a gradient
    (what the path-length regulariser differentiates) against the framework's broadcast expression."""
    from transeditor_amd.op.chanscale import chan_scale
    x = synth.normal(shape, 'cs.x').double().requires_grad_(True)
    s = (1 + 0.3 * synth.normal(shape[:2], 'cs.s')).double().requires_grad_(True)
    gy = synth.normal(shape, 'cs.g').double().requires_grad_(True)
    u, v = synth.normal(shape, 'cs.u').double(), synth.normal(shape[:2], 'cs.v').double()

    def run(x, s, gy, f, dev, dt):
        y = f(x, s)
        gx, gs = torch.autograd.grad((y * gy).sum(), (x, s), create_graph=True)
        pen = (gx * u.to(dev, dt)).sum() + (gs * v.to(dev, dt)).sum() + gs.pow(2).sum()
        return (y, gx, gs) + torch.autograd.grad(pen, (x, s, gy))

    ref = run(x, s, gy, lambda a, b: a * b[:, :, None, None], 'cpu', torch.float64)
    xd, sd, gd = (t.detach().to(DEV, torch.float32).requires_grad_(True) for t in (x, s, gy))
    got = run(xd, sd, gd, chan_scale, DEV, torch.float32)
    for name, a, b in zip(('y', 'gx', 'gs', 'ppx', 'pps', 'ppg'), got, ref):
        assert rel_err(a.double(), b) < 2e-5, name


@pytest.mark.gpu
@pytest.mark.parametrize('B,widths', [(16, [512, 512, 512, 256, 256, 128, 128, 512]), (2, [64] * 18 + [32, 32]), (1, [512, 3, 512])])
def test_batched_modulation_vs_single_layers(B, widths):
    """op/modulation.py: all style modulations of a pass as a few batched launches against the per-layer EqualLinear calls
    (model_spatial_query.py:286, 299): values, d latent (entries read by several layers accumulate), every dW / db; then the
    recorded backward against the per-layer route."""
    from transeditor_amd.model_spatial_query import EqualLinear
    from transeditor_amd.op.modulation import batched_modulation, supported
    L = 6
    mods = []
    for i, wd in enumerate(widths):
        m = EqualLinear(512, wd, bias_init=1)
        synth.fill_state_dict(m.state_dict(), 300 + i)
        mods.append(m.to(DEV))
    index = [i % L for i in range(len(widths))]
    lat = synth.normal((B, L, 512), 'bm.lat').to(DEV).requires_grad_(True)
    assert supported(lat, mods)
    got = batched_modulation(lat, mods, index)
    ref = [m(lat[:, i]) for m, i in zip(mods, index)]
    gys = [synth.normal(tuple(r.shape), f'bm.g{j}').to(DEV) for j, r in enumerate(ref)]
    for a, b in zip(got, ref):
        assert tuple(a.shape) == tuple(b.shape) and rel_err(a, b) < 1e-5
    params = [p for m in mods for p in (m.weight, m.bias)]
    ga = torch.autograd.grad(got, [lat] + params, gys)
    gb = torch.autograd.grad(ref, [lat] + params, gys)
    for j, (a, b) in enumerate(zip(ga, gb)):
        assert rel_err(a, b) < 2e-5, j
    # recorded backward: || d(sum_j <s_j, g_j>) / d lat ||^2 differentiated w.r.t. the weights
    lat2 = lat.detach().clone().requires_grad_(True)
    g1, = torch.autograd.grad(batched_modulation(lat2, mods, index), lat2, gys, create_graph=True)
    g2, = torch.autograd.grad([m(lat2[:, i]) for m, i in zip(mods, index)], lat2, gys, create_graph=True)
    wa = torch.autograd.grad(g1.square().sum(), [m.weight for m in mods])
    wb = torch.autograd.grad(g2.square().sum(), [m.weight for m in mods])
    for a, b in zip(wa, wb):
        assert rel_err(a, b) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('shape,K,J,n', [((16, 16, 512), 512, 128, 7), ((16, 16, 128), 128, 128, 2), ((3, 5, 64), 64, 32, 3),
                                         ((2, 1200, 64), 64, 16, 2)])
def test_shared_input_linears_vs_single_layers(shape, K, J, n):
    """op/linear.py::shared_input_linears - the key / value projections of an attention block and the query projections of
    blocks 1..7 (model_spatial_query.py:889-890, :675-678) as one launch - against the per-layer EqualLinear calls: values,
    dx (the sum over the layers), every dW / db, and the recorded backward."""
    from transeditor_amd.model_spatial_query import EqualLinear
    from transeditor_amd.op.linear import shared_input_linears, _SharedInputLinears
    mods = []
    for i in range(n):
        m = EqualLinear(K, J)
        synth.fill_state_dict(m.state_dict(), 400 + i)
        mods.append(m.to(DEV))
    x = synth.normal(shape, 'sil.x').to(DEV).requires_grad_(True)
    got = shared_input_linears(x, mods)
    assert got[0].grad_fn is not None and 'SharedInputLinears' in type(got[0].grad_fn).__name__
    ref = [m(x) for m in mods]
    gys = [synth.normal(tuple(r.shape), f'sil.g{j}').to(DEV) for j, r in enumerate(ref)]
    for a, b in zip(got, ref):
        assert tuple(a.shape) == tuple(b.shape) and rel_err(a, b) < 1e-5
    params = [p for m in mods for p in (m.weight, m.bias)]
    ga = torch.autograd.grad(got, [x] + params, gys)
    gb = torch.autograd.grad(ref, [x] + params, gys)
    for j, (a, b) in enumerate(zip(ga, gb)):
        assert rel_err(a, b) < 2e-5, j
    # only some outputs used downstream (the others get zero gradients from autograd)
    ga = torch.autograd.grad(shared_input_linears(x, mods)[0].square().sum(), [x, mods[0].weight])
    gb = torch.autograd.grad(mods[0](x).square().sum(), [x, mods[0].weight])
    for a, b in zip(ga, gb):
        assert rel_err(a, b) < 2e-5
    x2 = x.detach().clone().requires_grad_(True)
    g1, = torch.autograd.grad(shared_input_linears(x2, mods), x2, gys, create_graph=True)
    g2, = torch.autograd.grad([m(x2) for m in mods], x2, gys, create_graph=True)
    wa = torch.autograd.grad(g1.square().sum(), [m.weight for m in mods])
    wb = torch.autograd.grad(g2.square().sum(), [m.weight for m in mods])
    for a, b in zip(wa, wb):
        assert rel_err(a, b) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('Z,I,J,K', [(64, 16, 16, 32), (3, 37, 5, 70), (16, 16, 512, 512)])
def test_bmm_closed_family(Z, I, J, K, ta, tb):
    """op/linear.py::bmm - the batched product the recorded backward of the attention core / token-wise mapping is built from -
    against torch.matmul in fp64 for all four transposition forms: value, both gradients, and a second differentiation (every
    derivative is another member of the family, so this walks all of its branches)."""
    from transeditor_amd.op.linear import bmm
    a = synth.normal((Z, K, I) if ta else (Z, I, K), 'bmm.a').to(DEV).requires_grad_(True)
    b = synth.normal((Z, J, K) if tb else (Z, K, J), 'bmm.b').to(DEV).requires_grad_(True)
    gc = synth.normal((Z, I, J), 'bmm.g').to(DEV)

    def plain(a, b):
        return 0.37 * torch.matmul(a.transpose(1, 2) if ta else a, b.transpose(1, 2) if tb else b)

    def run(f, a, b, gc):
        c = f(a, b)
        ga, gb = torch.autograd.grad(c, (a, b), gc, create_graph=True)
        return (c, ga, gb) + torch.autograd.grad(ga.square().sum() + (gb * gb.detach().roll(1, 0)).sum(), (a, b))
    want = run(plain, a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True), gc.double())
    got = run(lambda a, b: bmm(a, b, ta, tb, 0.37), a, b, gc)
    grow = max(1.0, math.sqrt(K / 512))
    for name, x, y in zip(('c', 'ga', 'gb', 'gga', 'ggb'), got, want):
        assert rel_err(x, y.float()) < 3e-5 * grow, name


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['3x3', 'T2', '1x1'])
@pytest.mark.parametrize('B,C,H,W', [(32, 512, 4, 4), (16, 512, 8, 8), (32, 512, 16, 16), (32, 256, 32, 32), (16, 128, 64, 64),
                                     (32, 512, 5, 3), (24, 384, 32, 64)])
def test_wgrad_grouped_slabs_equal_per_sample_slabs(kind, B, C, H, W):
    """te_wgrad_group_f32 (NB samples share a slab: the plain weight gradient of the discriminator's small layers) against the
    per-sample form te_wgrad_f32 and against torch's fp64 convolution weight gradient: scalar and 16-byte staging paths, odd
    image sizes, the transposed kind's (2H+1)-sized operand; and the plan really groups where it should."""
    from transeditor_amd import _lib
    code = {'3x3': _lib.CONV_3X3, 'T2': _lib.CONV_T2, '1x1': _lib.CONV_1X1}[kind]
    x = synth.normal((B, C, H, W), 'wgg.x').to(DEV)
    g = synth.normal((B, C, 2 * H + 1, 2 * W + 1) if kind == 'T2' else (B, C, H, W), 'wgg.g').to(DEV)
    per = _lib.wgrad_slabs(g, x, code, H, W)
    grp = _lib.wgrad_slabs(g, x, code, H, W, group=True)
    assert per.shape[0] == B and grp.shape[2:] == per.shape[2:]
    if C % 128 == 0 and per.shape[1] == 1 and B * (C // 128) * (C // 64) // 2 >= 256:
        assert grp.shape[0] < B and B % grp.shape[0] == 0, tuple(grp.shape)          # grouping applies to these shapes
    a, b = grp.sum(dim=(0, 1)), per.sum(dim=(0, 1))
    assert rel_err(a, b) < 2e-5
    xd, gd = x.double().cpu(), g.double().cpu()
    if kind == 'T2':        # conv_transpose2d(x, w, stride 2) -> d w[ci, co, ky, kx];  slabs are [co, ci, tap]
        w = torch.zeros(C, C, 3, 3, dtype=torch.float64, requires_grad=True)
        ref, = torch.autograd.grad(F.conv_transpose2d(xd[:, :64], w[:64], stride=2), w, gd)
        ref = ref[:64].permute(1, 0, 2, 3).reshape(C, 64, 9)
        assert rel_err(a[:, :64], ref.float()) < 2e-5
    else:
        ks = 3 if kind == '3x3' else 1
        w = torch.zeros(64, C, ks, ks, dtype=torch.float64, requires_grad=True)
        ref, = torch.autograd.grad(F.conv2d(xd, w, padding=ks // 2), w, gd[:, :64])
        assert rel_err(a[:64], ref.reshape(64, C, ks * ks).float()) < 2e-5
