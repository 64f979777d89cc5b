"""The weight-gradient correlation of the 3x3 convolutions on the bf16 matrix pipe (csrc/wgrad6.hip: pair form F(3,2), three bf16 pieces per
fp32 operand, six exact piece products per multiply-add, fp32 accumulation) against fp64 torch and against the fp32 kernel (same slabs,
wgrad.hip) - reference: autograd of F.conv2d(groups = B) in ModulatedConv2d.forward, model_spatial_query.py:318-333.  Shapes: one and
several 32-column tiles, chunk boundaries inside a column (S > 1), image borders on all sides, several channel blocks, the transposed
kind (TE_CONV_T2: the up- / down-sampling layers), the grouped form
(samples share a slab), the three gradients of the reducer on top of the slabs, the range sweep, the selection rule and the switch.
Pinned at the bar of the fp32 kernels: 5e-6 against fp64."""
import math

import pytest
import torch

from conftest import rel_err
from transeditor_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'

SHAPES = [(2, 64, 64, 8, 32), (3, 128, 64, 13, 64), (1, 64, 192, 40, 96), (4, 64, 64, 5, 32), (2, 128, 128, 32, 32), (1, 64, 64, 1, 32),
          (2, 64, 128, 2, 64), (1, 192, 64, 3, 32)]
# round 6, the sample-pair form (Co == Ci == 32, even batch: two samples on the diagonal of one 64 x 64 block tile) - the 32-channel
# layer of the FFHQ-1024 generator (reference channel table model_spatial_query.py:473-483)
PAIR_SHAPES = [(2, 32, 32, 8, 32), (4, 32, 32, 13, 64), (6, 32, 32, 40, 96), (2, 32, 32, 1, 32), (8, 32, 32, 3, 32)]


@pytest.fixture(autouse=True)
def _restore_switch():
    old = _lib.wgrad_split()
    yield
    _lib.wgrad_split(old)


def _fp64_corr(g, x):
    B, Co, Ci = g.shape[0], g.shape[1], x.shape[1]
    return torch.stack([torch.nn.grad.conv2d_weight(x[b:b + 1].double(), (Co, Ci, 3, 3), g[b:b + 1].double(), padding=1)
                        for b in range(B)]).reshape(B, Co, Ci, 9)


@pytest.mark.parametrize('B,Co,Ci,H,W', SHAPES + PAIR_SHAPES)
def test_split_bf16_weight_gradient_slabs_vs_fp64(B, Co, Ci, H, W):
    assert _lib.wgrad_split_ok(_lib.CONV_3X3, Co, Ci, H, W)
    g = synth.normal((B, Co, H, W), f'wg6.g.{Co}.{H}').to(DEV)
    x = synth.normal((B, Ci, H, W), f'wg6.x.{Ci}.{H}').to(DEV)
    want = _fp64_corr(g, x)
    _lib.wgrad_split(1)
    got = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W)
    _lib.wgrad_split(0)
    ref = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W)
    if Co != 32:
        assert got.shape == ref.shape                   # same slab count, same layout: the reducers do not know which kernel ran
    else:                                               # (the pair form runs one block per sample PAIR: its plan may take more chunks)
        assert got.shape[0] == ref.shape[0] and got.shape[2:] == ref.shape[2:]
    got, ref = got.sum(1), ref.sum(1)
    l2 = lambda a: float((a.double() - want).norm() / want.norm())
    print(f'split-bf16 weight gradient {Ci}->{Co} @{H}x{W} B{B}: max {rel_err(got, want):.2e} (fp32 kernel {rel_err(ref, want):.2e}), '
          f'L2 {l2(got):.2e} ({l2(ref):.2e})')
    assert rel_err(got, want) < 5e-6
    assert l2(got) < 2.5 * l2(ref) + 1e-7               # fp32-equivalent: the yardstick of the other split kernels


def test_split_bf16_weight_gradient_long_reduction_vs_fp32_kernel():
    """a chunk of several thousand steps per accumulator (the 128-channel layer of FFHQ-256 at batch 2): against the fp32 kernel"""
    B, Co, Ci, H, W = 2, 128, 128, 256, 256
    g = synth.normal((B, Co, H, W), 'wg6.lg').to(DEV)
    x = synth.normal((B, Ci, H, W), 'wg6.lx').to(DEV)
    _lib.wgrad_split(1)
    got = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W).sum(1)
    _lib.wgrad_split(0)
    ref = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W).sum(1)
    want = _fp64_corr(g[:1, :64], x[:1, :64])           # fp64 for one 64 x 64 channel block of sample 0
    e_split, e_ref = rel_err(got[:1, :64, :64], want), rel_err(ref[:1, :64, :64], want)
    print(f'long reduction: split vs fp32 kernel {rel_err(got, ref):.2e}; vs fp64 (one block): split {e_split:.2e}, fp32 kernel {e_ref:.2e}')
    assert rel_err(got, ref) < 5e-6
    assert e_split < 5e-6 and e_split < 2.5 * e_ref + 1e-7


def test_split_bf16_weight_gradient_pair_form_long_reduction_and_odd_batch():
    """the sample-pair form at the FFHQ-1024 layer's own size class (32 channels, 256 x 256 here, batch 4: thousands of steps per
    accumulator, S > 1 chunks that end inside a column) against the fp32 kernel and fp64; an ODD batch cannot be paired and runs the
    fp32 kernel: identical slabs whatever the switch says"""
    B, C, H, W = 4, 32, 256, 256
    g = synth.normal((B, C, H, W), 'wg6.pg').to(DEV)
    x = synth.normal((B, C, H, W), 'wg6.px').to(DEV)
    _lib.wgrad_split(1)
    got = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W).sum(1)
    _lib.wgrad_split(0)
    ref = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W).sum(1)
    want = _fp64_corr(g, x)
    e_split, e_ref = rel_err(got, want), rel_err(ref, want)
    print(f'pair form, long reduction: vs fp64 split {e_split:.2e}, fp32 kernel {e_ref:.2e}; split vs fp32 kernel {rel_err(got, ref):.2e}')
    assert e_split < 5e-6 and e_split < 2.5 * e_ref + 1e-7
    for b in range(B):                                   # every sample's slab is ITS correlation (no mix-up inside a pair)
        assert rel_err(got[b], want[b]) < 5e-6
    g3, x3 = g[:3, :, :16, :32].contiguous(), x[:3, :, :16, :32].contiguous()
    _lib.wgrad_split(1)
    a = _lib.wgrad_slabs(g3, x3, _lib.CONV_3X3, 16, 32)
    _lib.wgrad_split(0)
    b3 = _lib.wgrad_slabs(g3, x3, _lib.CONV_3X3, 16, 32)
    assert torch.equal(a, b3)


def test_split_bf16_weight_gradient_pair_form_reduced():
    """the reducer's three gradients (dW, d style scale, d demodulation) on the pair form's slabs against the fp32 kernel's"""
    B, C, H, W = 4, 32, 64, 64
    g = synth.normal((B, C, H, W), 'wg6.qg').to(DEV)
    x = synth.normal((B, C, H, W), 'wg6.qx').to(DEV)
    w = (synth.normal((C, C, 3, 3), 'wg6.qw') / (3 * math.sqrt(C))).to(DEV)
    isc, osc = (1 + 0.3 * synth.normal((B, C), 'wg6.qi')).to(DEV), (1 + 0.3 * synth.normal((B, C), 'wg6.qo')).to(DEV)
    out = {}
    for on in (0, 1):
        _lib.wgrad_split(on)
        out[on] = _lib.wgrad_reduce(_lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W), w, 0.7, isc, osc, True, True, True)
    for a, b, name in zip(out[1], out[0], ('dW', 'd isc', 'd osc')):
        print(f'reducer on pair-form slabs, {name}: vs fp32 slabs {rel_err(a, b):.2e}')
        assert rel_err(a, b) < 5e-6


def test_split_bf16_weight_gradient_grouped_and_reduced():
    """the grouped form (NB samples per slab, the discriminator's plain gradient) and the reducer's three gradients on the split slabs"""
    B, Co, Ci, H, W = 8, 128, 128, 32, 32
    g = synth.normal((B, Co, H, W), 'wg6.gg').to(DEV)
    x = synth.normal((B, Ci, H, W), 'wg6.gx').to(DEV)
    w = (synth.normal((Co, Ci, 3, 3), 'wg6.w') / (3 * math.sqrt(Ci))).to(DEV)
    isc, osc = (1 + 0.3 * synth.normal((B, Ci), 'wg6.i')).to(DEV), (1 + 0.3 * synth.normal((B, Co), 'wg6.o')).to(DEV)
    out = {}
    for on in (0, 1):
        _lib.wgrad_split(on)
        grouped = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W, group=True)
        slabs = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W)
        out[on] = (grouped.sum((0, 1)),) + tuple(_lib.wgrad_reduce(slabs, w, 0.7, isc, osc, True, True, True))
    want_plain = _fp64_corr(g, x).sum(0)
    assert rel_err(out[1][0].reshape(Co, Ci, 9), want_plain) < 5e-6
    for a, b, name in zip(out[1][1:], out[0][1:], ('dW', 'd isc', 'd osc')):
        print(f'reducer on split slabs, {name}: vs fp32 slabs {rel_err(a, b):.2e}')
        assert rel_err(a, b) < 5e-6


T_SHAPES = [(2, 64, 64, 8, 16), (3, 128, 64, 13, 32), (1, 64, 192, 20, 48), (4, 64, 64, 5, 16), (1, 64, 64, 1, 16), (2, 128, 128, 32, 32),
            # round 6, narrow sides: 32 valid channels in the last 64-channel block of a side (the 64 -> 32 up-sampling layer of FFHQ-1024)
            (2, 32, 64, 8, 16), (3, 64, 32, 13, 32), (1, 96, 64, 20, 48), (2, 32, 96, 5, 16), (1, 96, 160, 6, 16)]


@pytest.mark.parametrize('B,Co,Ci,H,W', T_SHAPES)
def test_split_bf16_weight_gradient_transposed_kind_vs_fp64(B, Co, Ci, H, W):
    """kind TE_CONV_T2: slab[co][ci][ky][kx] = sum g[co, 2i + ky, 2j + kx] x[ci, i, j] - the weight gradient of the generator's up-sampling
    convolutions (model_spatial_query.py:310-321) and, with the two tensors swapped, of the discriminator's down-sampling ones (:765-779)"""
    assert _lib.wgrad_split_ok(_lib.CONV_T2, Co, Ci, H, W)
    g = synth.normal((B, Co, 2 * H + 1, 2 * W + 1), f'wg6t.g.{Co}.{H}').to(DEV)
    x = synth.normal((B, Ci, H, W), f'wg6t.x.{Ci}.{H}').to(DEV)
    want = torch.stack([torch.nn.grad.conv2d_weight(g[b:b + 1].double(), (Ci, Co, 3, 3), x[b:b + 1].double(), stride=2)
                        for b in range(B)]).transpose(1, 2).reshape(B, Co, Ci, 9)
    _lib.wgrad_split(1)
    got = _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W)
    _lib.wgrad_split(0)
    ref = _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W)
    assert got.shape == ref.shape
    got, ref = got.sum(1), ref.sum(1)
    l2 = lambda a: float((a.double() - want).norm() / want.norm())
    print(f'split-bf16 weight gradient, transposed kind {Ci}->{Co} @{H}x{W} B{B}: max {rel_err(got, want):.2e} (fp32 kernel '
          f'{rel_err(ref, want):.2e}), L2 {l2(got):.2e} ({l2(ref):.2e})')
    assert rel_err(got, want) < 5e-6
    assert l2(got) < 2.5 * l2(ref) + 1e-7


@pytest.mark.parametrize('B,Co,Ci,H,W', [(2, 128, 128, 128, 128), (2, 32, 64, 128, 128)])
def test_split_bf16_weight_gradient_transposed_kind_long_reduction(B, Co, Ci, H, W):
    g = synth.normal((B, Co, 2 * H + 1, 2 * W + 1), 'wg6t.lg').to(DEV)
    x = synth.normal((B, Ci, H, W), 'wg6t.lx').to(DEV)
    _lib.wgrad_split(1)
    got = _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W).sum(1)
    _lib.wgrad_split(0)
    ref = _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W).sum(1)
    assert rel_err(got, ref) < 5e-6


@pytest.mark.parametrize('scale', [1e-30, 1e-12, 1.0, 1e12, 1e18])
def test_split_bf16_weight_gradient_range(scale):
    """operands from 1e-30 to 1e18: the split keeps 24 mantissa bits wherever the pieces stay normal bf16 numbers (they share fp32's
    exponent range), and products that overflow fp32 overflow in both kernels"""
    B, Co, Ci, H, W = 1, 64, 64, 8, 32
    g = (synth.normal((B, Co, H, W), 'wg6.rg') * scale).to(DEV)
    x = synth.normal((B, Ci, H, W), 'wg6.rx').to(DEV)
    want = _fp64_corr(g, x)
    _lib.wgrad_split(1)
    got = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W).sum(1)
    assert torch.isfinite(got).all()
    tol = 5e-6 if scale >= 1e-12 else 2e-2               # (1e-30: the low pieces are bf16 subnormals / flushed, as in the forward kernels)
    assert rel_err(got, want) < tol


def test_split_bf16_weight_gradient_selection_and_switch():
    ok = _lib.wgrad_split_ok
    assert ok(_lib.CONV_3X3, 128, 128, 256, 256) and ok(_lib.CONV_3X3, 512, 64, 4, 32)
    assert not ok(_lib.CONV_3X3, 96, 128, 32, 32) and not ok(_lib.CONV_3X3, 128, 32, 32, 32)      # whole 64-channel blocks only ...
    assert ok(_lib.CONV_3X3, 32, 32, 1024, 1024) and not ok(_lib.CONV_3X3, 32, 64, 32, 32)        # ... or the 32 x 32 sample-pair form
    assert not ok(_lib.CONV_3X3, 128, 128, 16, 16) and not ok(_lib.CONV_3X3, 128, 128, 32, 48)    # whole 32-column tiles only
    assert ok(_lib.CONV_T2, 128, 128, 32, 32) and ok(_lib.CONV_T2, 512, 512, 16, 16)
    assert not ok(_lib.CONV_T2, 128, 128, 8, 8) and not ok(_lib.CONV_T2, 80, 128, 32, 32)
    assert ok(_lib.CONV_1X1, 128, 128, 32, 32) and ok(_lib.CONV_1X1, 512, 256, 64, 64)          # round 6: whole 128-channel blocks, W % 16 == 0
    assert not ok(_lib.CONV_1X1, 128, 64, 32, 32) and not ok(_lib.CONV_1X1, 128, 128, 8, 8) and not ok(_lib.CONV_1X1, 128, 128, 32, 24)
    assert ok(_lib.CONV_T2, 32, 64, 512, 512) and ok(_lib.CONV_T2, 96, 128, 32, 32) and not ok(_lib.CONV_T2, 32, 32, 32, 32)   # narrow sides
    old = _lib.wgrad_split(0)
    assert _lib.wgrad_split() == 0 and _lib.wgrad_split(1) == 0 and _lib.wgrad_split() == 1
    _lib.wgrad_split(old)
    # a shape the kernel does not cover runs the fp32 kernel whatever the switch says: identical slabs
    g = synth.normal((2, 96, 16, 16), 'wg6.sg').to(DEV)
    x = synth.normal((2, 64, 16, 16), 'wg6.sx').to(DEV)
    _lib.wgrad_split(1)
    a = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, 16, 16)
    _lib.wgrad_split(0)
    b = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, 16, 16)
    assert torch.equal(a, b)


@pytest.mark.parametrize('B,Co,Ci,H,W', [(2, 64, 128, 8, 16), (3, 128, 256, 13, 32), (1, 64, 384, 20, 48), (2, 192, 128, 5, 16), (2, 128, 128, 64, 64),
                                         (4, 64, 128, 1, 16), (1, 256, 512, 16, 16)])
def test_split_bf16_weight_gradient_transposed_kind_wide_form_is_bit_identical(B, Co, Ci, H, W):
    """wgrad6tw_kernel (round 6: 64 channels of the big tensor x 128 of the small one per block, taken where Ci % 128 == 0) against
    wgrad6t_kernel (64 x 64): the same products in the same order per slab element - identical bits; 1 - 4 blocks of 128 channels, 1 - 3
    of 64, single rows, odd row counts, several samples; and against fp64"""
    g = synth.normal((B, Co, 2 * H + 1, 2 * W + 1), f'wg6w.g.{Co}.{H}').to(DEV)
    x = synth.normal((B, Ci, H, W), f'wg6w.x.{Ci}.{H}').to(DEV)
    _lib.wgrad_split(1)
    old = _lib.wgrad_t2_wide(-1)
    try:
        _lib.wgrad_t2_wide(0)
        narrow = _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W)
        _lib.wgrad_t2_wide(1)
        wide = _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W)
    finally:
        _lib.wgrad_t2_wide(old)
    assert _lib.wgrad_t2_wide(-1) == old
    assert torch.equal(narrow, wide)
    want = torch.stack([torch.nn.grad.conv2d_weight(g[b:b + 1].double(), (Ci, Co, 3, 3), x[b:b + 1].double(), stride=2)
                        for b in range(B)]).transpose(1, 2).reshape(B, Co, Ci, 9)
    assert rel_err(wide.sum(1), want) < 5e-6


@pytest.mark.parametrize('B,Co,Ci,H,W', [(2, 128, 128, 8, 16), (3, 256, 128, 13, 32), (1, 128, 384, 20, 48), (4, 128, 128, 1, 16), (2, 256, 128, 64, 64),
                                         (2, 128, 128, 5, 80), (8, 512, 512, 16, 16)])
def test_split_bf16_weight_gradient_1x1_kind_vs_fp64(B, Co, Ci, H, W):
    """kind TE_CONV_1X1 (round 6, wgrad6p_kernel): slab[co][ci] = sum over cells g[co, cell] x[ci, cell] - the weight gradient of the
    discriminator's ResBlock skip convolutions (model_spatial_query.py:173-181, :780-798) - against fp64 and the fp32 kernel: single rows,
    odd row counts, 1 - 3 blocks of 128 channels per side, several column tiles, and the grouped form (samples share a slab)"""
    assert _lib.wgrad_split_ok(_lib.CONV_1X1, Co, Ci, H, W)
    g = synth.normal((B, Co, H, W), f'wg6p.g.{Co}.{H}').to(DEV)
    x = synth.normal((B, Ci, H, W), f'wg6p.x.{Ci}.{H}').to(DEV)
    want = torch.einsum('bohw,bihw->boi', g.double(), x.double())
    _lib.wgrad_split(1)
    got = _lib.wgrad_slabs(g, x, _lib.CONV_1X1, H, W)
    _lib.wgrad_split(0)
    ref = _lib.wgrad_slabs(g, x, _lib.CONV_1X1, H, W)
    # (the chunk count S follows the kernel that runs: 128 x 128 channel blocks for the split kernel)
    assert got.shape[0] == ref.shape[0] == B and tuple(got.shape[2:]) == tuple(ref.shape[2:]) == (Co, Ci, 1)
    got, ref = got.sum(1).squeeze(-1), ref.sum(1).squeeze(-1)
    l2 = lambda a: float((a.double() - want).norm() / want.norm())
    print(f'split-bf16 weight gradient, 1x1 kind {Ci}->{Co} @{H}x{W} B{B}: max {rel_err(got, want):.2e} (fp32 kernel {rel_err(ref, want):.2e}), '
          f'L2 {l2(got):.2e} ({l2(ref):.2e})')
    assert rel_err(got, want) < 5e-6
    assert l2(got) < 2.5 * l2(ref) + 1e-7
    # grouped form: the plain gradient, samples of a group share a slab
    _lib.wgrad_split(1)
    gg = _lib.wgrad_slabs(g, x, _lib.CONV_1X1, H, W, group=True)
    _lib.wgrad_split(0)
    gr = _lib.wgrad_slabs(g, x, _lib.CONV_1X1, H, W, group=True)
    assert tuple(gg.shape[2:]) == tuple(gr.shape[2:]) == (Co, Ci, 1)
    assert rel_err(gg.sum((0, 1)).squeeze(-1), want.sum(0)) < 5e-6
    assert rel_err(gr.sum((0, 1)).squeeze(-1), want.sum(0)) < 5e-6


@pytest.mark.parametrize('scale', [1e-30, 1e-12, 1e12, 1e18])
def test_split_bf16_weight_gradient_1x1_kind_range(scale):
    B, Co, Ci, H, W = 2, 128, 128, 8, 32
    g = (synth.normal((B, Co, H, W), 'wg6p.rg') * math.sqrt(scale)).to(DEV)
    x = (synth.normal((B, Ci, H, W), 'wg6p.rx') * math.sqrt(scale)).to(DEV)
    want = torch.einsum('bohw,bihw->boi', g.double(), x.double())
    _lib.wgrad_split(1)
    got = _lib.wgrad_slabs(g, x, _lib.CONV_1X1, H, W).sum(1).squeeze(-1)
    assert torch.isfinite(got).all()
    assert rel_err(got, want) < 5e-6
