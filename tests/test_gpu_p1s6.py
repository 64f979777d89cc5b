"""The plain 1x1 convolution on the bf16 matrix pipe (csrc/p1s6.hip, kind TE_CONV_1X1S6, round 6: three bf16 pieces per fp32 operand, six
exact piece products per multiply-add, fp32 accumulation) against fp64 torch and against the fp32 kernel (TE_CONV_1X1) - reference: the
1x1 EqualConv2d of ResBlock.skip, model_spatial_query.py:173-181, :780-798, and its data gradient.  Shapes: one to eight 64-channel
stages, one to four blocks of 128 output channels, one and several 256-pixel tiles per sample, odd batches; with and without the
residual; the data-gradient packing; range sweep; non-finite propagation; the selection rule; the ResBlock node on top of it.
Pinned at the bar of the fp32 kernels: 5e-6 against fp64."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from transeditor_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'

#          B, K,   M,   H,  W
# (the kind is taken from half a block per CU up - te_conv_p1s6_supported -, hence the batches)
SHAPES = [(128, 64, 128, 16, 16), (40, 128, 256, 16, 32), (9, 512, 512, 32, 32), (16, 128, 256, 64, 64), (15, 192, 384, 16, 48),
          (32, 512, 512, 16, 16), (17, 256, 128, 32, 64)]


def _l2(a, want):
    return float((a.double() - want).norm() / want.norm())


@pytest.mark.parametrize('B,K,M,H,W', SHAPES)
@pytest.mark.parametrize('with_res', [False, True])
def test_split_bf16_1x1_forward_vs_fp64(B, K, M, H, W, with_res):
    assert _lib.p1s6_ok(B, K, M, H, W)
    x = synth.normal((B, K, H, W), f'p1.x.{K}.{H}').to(DEV)
    w = (synth.normal((M, K, 1, 1), f'p1.w.{M}.{K}') / math.sqrt(K)).to(DEV)
    res = synth.normal((B, M, H, W), f'p1.r.{M}.{H}').to(DEV) if with_res else None
    ws = 0.71
    want = F.conv2d(x.double(), w.double() * ws)
    if with_res:
        want = want + res.double()
    got = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_P6FWD, ws), _lib.CONV_1X1S6, M, H, W, res=res)
    ref = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD, ws), _lib.CONV_1X1, M, H, W, res=res)
    print(f'split-bf16 1x1 {K}->{M} @{H}x{W} B{B} res {with_res}: max {rel_err(got, want):.2e} (fp32 kernel {rel_err(ref, want):.2e}), '
          f'L2 {_l2(got, want):.2e} ({_l2(ref, want):.2e})')
    assert rel_err(got, want) < 5e-6
    assert _l2(got, want) < 2.5 * _l2(ref, want) + 1e-7          # fp32-equivalent: the yardstick of the other split kernels


@pytest.mark.parametrize('B,K,M,H,W', SHAPES[:4])
def test_split_bf16_1x1_data_gradient_packing_vs_fp64(B, K, M, H, W):
    """the data gradient of a Co -> Ci... 1x1 layer with weight w [Co = K, Ci = M] is the same product with the weights transposed"""
    g = synth.normal((B, K, H, W), f'p1.g.{K}.{H}').to(DEV)
    w = (synth.normal((K, M, 1, 1), f'p1.wd.{M}.{K}') / math.sqrt(K)).to(DEV)          # a layer from M to K channels
    want = F.conv_transpose2d(g.double(), w.double() * 0.9)
    got = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_P6DGRAD, 0.9), _lib.CONV_1X1S6, M, H, W)
    ref = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_DGRAD, 0.9), _lib.CONV_1X1, M, H, W)
    assert rel_err(got, want) < 5e-6
    assert _l2(got, want) < 2.5 * _l2(ref, want) + 1e-7


def test_split_bf16_1x1_every_pixel_and_channel_lands_where_it_belongs():
    """one-hot probes: output (b, m, p) must be w[m, k] for an input that is 1 at (b, k, p) only - catches any permutation of pixels
    inside the 256-pixel tile (the kernel keeps pixel 4 q + e at LDS position 32 e + q), of channels inside a stage, of samples"""
    B, K, M, H, W = 33, 128, 256, 16, 32
    w = synth.normal((M, K, 1, 1), 'p1.hot.w').to(DEV)
    wp = _lib.conv_pack(w, _lib.PACK_P6FWD)
    gen = torch.Generator().manual_seed(5)
    x = torch.zeros(B, K, H, W)
    probes = []
    for b in range(B):
        for _ in range(12):
            k, y, xx = int(torch.randint(K, (1,), generator=gen)), int(torch.randint(H, (1,), generator=gen)), int(torch.randint(W, (1,), generator=gen))
            if x[b, :, y, xx].abs().sum() == 0:
                x[b, k, y, xx] = 1.0
                probes.append((b, k, y, xx))
    out = _lib.conv(x.to(DEV), wp, _lib.CONV_1X1S6, M, H, W)
    touched = torch.zeros(B, H, W, dtype=torch.bool)
    for b, k, y, xx in probes:
        assert rel_err(out[b, :, y, xx], w[:, k, 0, 0]) < 1e-6, (b, k, y, xx)
        touched[b, y, xx] = True
    assert float(out.permute(0, 2, 3, 1)[~touched.to(DEV)].abs().max()) == 0.0


@pytest.mark.parametrize('scale', [1e-30, 1e-15, 1e+15, 1e+30])
def test_split_bf16_1x1_scale_sweep(scale):
    B, K, M, H, W = 64, 128, 128, 16, 32
    x = (synth.normal((B, K, H, W), 'p1.sx') * math.sqrt(scale)).to(DEV)
    w = (synth.normal((M, K, 1, 1), 'p1.sw') * math.sqrt(scale) / math.sqrt(K)).to(DEV)
    want = F.conv2d(x.double(), w.double())
    got = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_P6FWD), _lib.CONV_1X1S6, M, H, W)
    assert torch.isfinite(got).all()
    assert rel_err(got, want) < (5e-6 if scale > 1e-20 else 2e-2)      # (1e-30: the low pieces are bf16 subnormals, as in the other split kernels)


def test_split_bf16_1x1_non_finite_inputs_stay_at_their_pixel():
    B, K, M, H, W = 128, 64, 128, 16, 16
    x = synth.normal((B, K, H, W), 'p1.nx').to(DEV)
    w = synth.normal((M, K, 1, 1), 'p1.nw').to(DEV)
    w = torch.where(w.abs() < 1e-3, torch.full_like(w, 1e-3), w)
    clean = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD), _lib.CONV_1X1, M, H, W)
    xp = x.clone()
    plants = [(0, 3, 5, 7, float('inf')), (127, 60, 15, 15, float('nan')), (64, 17, 0, 0, float('-inf'))]
    for b, k, y, xx, v in plants:
        xp[b, k, y, xx] = v
    got = _lib.conv(xp, _lib.conv_pack(w, _lib.PACK_P6FWD), _lib.CONV_1X1S6, M, H, W)
    expect = torch.zeros(B, 1, H, W, dtype=torch.bool, device=DEV)
    for b, k, y, xx, v in plants:
        expect[b, 0, y, xx] = True
    expect = expect.expand(B, M, H, W)
    assert torch.equal(~torch.isfinite(got), expect)
    assert rel_err(got[~expect], clean[~expect]) < 5e-6


def test_split_bf16_1x1_selection_and_argument_checks():
    ok = _lib.p1s6_ok
    assert ok(32, 128, 256, 128, 128) and ok(32, 512, 512, 16, 16) and ok(16, 256, 128, 128, 128)
    assert not ok(32, 512, 512, 8, 8) and not ok(32, 96, 256, 32, 32) and not ok(32, 128, 192, 32, 32) and not ok(1, 128, 128, 16, 16)
    from transeditor_amd.op import modconv
    w = torch.empty(256, 128, 1, 1)
    assert modconv.plain_1x1_kinds(32, w, 128, 128) == (_lib.PACK_P6FWD, _lib.CONV_1X1S6)
    assert modconv.plain_1x1_kinds(32, w, 128, 128, dgrad=True) == (_lib.PACK_P6DGRAD, _lib.CONV_1X1S6)
    assert modconv.plain_1x1_kinds(32, w, 8, 8) == (_lib.PACK_FWD, _lib.CONV_1X1)
    old = modconv.USE_SPLIT_1X1
    try:
        modconv.USE_SPLIT_1X1 = False
        assert modconv.plain_1x1_kinds(32, w, 128, 128) == (_lib.PACK_FWD, _lib.CONV_1X1)
    finally:
        modconv.USE_SPLIT_1X1 = old
    x = torch.randn(64, 128, 16, 16, device=DEV)
    wp = _lib.conv_pack(torch.randn(256, 128, 1, 1, device=DEV), _lib.PACK_P6FWD)
    assert _lib.p1s6_ok(64, 128, 256, 16, 16)
    with pytest.raises(RuntimeError):                       # scales / bias / activation are not this kind's business
        _lib.conv(x, wp, _lib.CONV_1X1S6, 256, 16, 16, None, None, torch.randn(256, device=DEV), 3)


def test_resblock_node_on_the_split_1x1_kernel_matches_the_fp32_route():
    """the discriminator's ResBlock (op/resblock.py) forward + backward with the skip branch on TE_CONV_1X1S6 against the same node
    with TE_SPLIT_1X1 off: outputs and every gradient at 5e-6"""
    from transeditor_amd.model_spatial_query import ResBlock
    from transeditor_amd.op import modconv
    torch.manual_seed(3)
    blk = ResBlock(128, 256).to(DEV)
    x = torch.randn(32, 128, 64, 64, device=DEV, requires_grad=True)
    gout = torch.randn(32, 256, 32, 32, device=DEV)
    assert _lib.p1s6_ok(32, 128, 256, 32, 32) and _lib.p1s6_ok(32, 256, 128, 32, 32)
    outs = {}
    old = modconv.USE_SPLIT_1X1
    try:
        for on in (True, False):
            modconv.USE_SPLIT_1X1 = on
            for p_ in list(blk.parameters()) + [x]:
                p_.grad = None
            y = blk(x)
            y.backward(gout)
            outs[on] = [y.detach().clone(), x.grad.clone()] + [p_.grad.clone() for p_ in blk.parameters()]
    finally:
        modconv.USE_SPLIT_1X1 = old
    for a, b in zip(outs[True], outs[False]):
        assert rel_err(a, b) < 5e-6
