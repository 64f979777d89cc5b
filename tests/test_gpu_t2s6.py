"""TE_CONV_T2S6: the transposed 3x3 / stride 2 convolution on the bf16 matrix pipe (csrc/t2s6.hip: body cells, three bf16 pieces per
fp32 operand, six exact piece products per multiply-add, fp32 accumulation; last output row / column through the fp32 kernel from the
plain copy of the weights in the same packed buffer) against fp64 torch and against the fp32 kernel (TE_CONV_T2) - both weight layouts
(forward of the generator's up-sampling layers, conv_transpose2d(stride 2) of ModulatedConv2d.forward, model_spatial_query.py:310-321;
data gradient of the discriminator's down-sampling convolutions, :765-779), style scale at staging, demodulation scale / bias /
leaky-ReLU epilogue, image borders (the halo row above / column left of the image, the last output row and column), single- and
multi-tile images, the range sweep, non-finite inputs, the selection rule.  Pinned at the bar of the fp32 kernels: 5e-6 against fp64."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from transeditor_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'

SHAPES = [(2, 32, 64, 8, 16), (3, 96, 192, 24, 32), (1, 48, 64, 16, 48), (2, 160, 128, 8, 16), (1, 512, 512, 16, 16), (2, 128, 256, 32, 64),
          (4, 64, 64, 8, 16)]


@pytest.mark.parametrize('B,K,M,H,W', SHAPES)
def test_split_bf16_transposed_conv_vs_fp64(B, K, M, H, W):
    assert _lib.t2s6_ok(B, K, M, H, W)
    x = synth.normal((B, K, H, W), f't6.x.{K}.{H}').to(DEV)
    w = (synth.normal((M, K, 3, 3), f't6.w.{M}.{K}') / (3 * math.sqrt(K))).to(DEV)          # model layout [Co, Ci, 3, 3]
    isc = (1 + 0.3 * synth.normal((B, K), 't6.isc')).to(DEV)
    osc = (1 + 0.3 * synth.normal((B, M), 't6.osc')).to(DEV)
    ws = 0.83
    want = F.conv_transpose2d(x.double() * isc.double()[:, :, None, None], (w.double() * ws).transpose(0, 1), stride=2) * osc.double()[:, :, None, None]
    got = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_T6FWD, ws), _lib.CONV_T2S6, M, H, W, isc, osc)
    direct = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD, ws), _lib.CONV_T2, M, H, W, isc, osc)
    assert got.shape == (B, M, 2 * H + 1, 2 * W + 1)
    l2 = lambda a: float((a.double() - want).norm() / want.norm())
    print(f'split-bf16 transposed {K}->{M} @{H}x{W}: max {rel_err(got, want):.2e} (fp32 kernel {rel_err(direct, want):.2e}), L2 {l2(got):.2e} ({l2(direct):.2e})')
    assert rel_err(got, want) < 5e-6
    assert l2(got) < 2.5 * l2(direct) + 1e-7
    # the border rows / columns on their own (the halo of the body tiles and the two fp32 regions)
    for sl in ((slice(None), slice(None), slice(0, 2)), (slice(None), slice(None), slice(2 * H - 1, None)),
               (slice(None), slice(None), slice(None), slice(0, 2)), (slice(None), slice(None), slice(None), slice(2 * W - 1, None))):
        assert rel_err(got[sl], want[sl]) < 5e-6
    # the launch as data gradient of the strided kind: from M to K channels on the swapped layout (weight [Co = M, Ci = K, 3, 3])
    if _lib.t2s6_ok(B, M, K, H, W):
        g = synth.normal((B, M, H, W), f't6.g.{M}.{H}').to(DEV)
        want_g = F.conv_transpose2d(g.double(), w.double() * ws, stride=2)
        got_g = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_T6SWAP, ws), _lib.CONV_T2S6, K, H, W)
        assert rel_err(got_g, want_g) < 5e-6


@pytest.mark.parametrize('act', [0, 3, 4])
def test_split_bf16_transposed_conv_epilogue(act):
    B, K, M, H, W = 2, 64, 128, 8, 32
    x = synth.normal((B, K, H, W), 't6.ex').to(DEV)
    w = (synth.normal((M, K, 3, 3), 't6.ew') / (3 * math.sqrt(K))).to(DEV)
    isc, osc = (1 + 0.3 * synth.normal((B, K), 't6.ei')).to(DEV), (1 + 0.3 * synth.normal((B, M), 't6.eo')).to(DEV)
    bias = synth.normal((M,), 't6.eb').to(DEV)
    a = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_T6FWD), _lib.CONV_T2S6, M, H, W, isc, osc, bias, act)
    b = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD), _lib.CONV_T2, M, H, W, isc, osc, bias, act)
    pre = F.conv_transpose2d(x.double() * isc.double()[:, :, None, None], w.double().transpose(0, 1), stride=2) * osc.double()[:, :, None, None] \
        + bias.double()[None, :, None, None]
    keep = (pre.abs() > 1e-5) if act else torch.ones_like(pre, dtype=torch.bool)
    assert rel_err(a * keep, b * keep) < 5e-6


@pytest.mark.parametrize('scale', [1e-30, 1e-15, 1e+15, 1e+30])
def test_split_bf16_transposed_conv_scale_sweep(scale):
    B, K, M, H, W = 2, 64, 128, 8, 32
    x = (synth.normal((B, K, H, W), 't6.sx') * math.sqrt(scale)).to(DEV)
    w = (synth.normal((M, K, 3, 3), 't6.sw') / (3 * math.sqrt(K)) * math.sqrt(scale)).to(DEV)
    want = F.conv_transpose2d(x.double(), w.double().transpose(0, 1), stride=2)
    got = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_T6FWD), _lib.CONV_T2S6, M, H, W)
    assert torch.isfinite(got).all()
    assert rel_err(got, want) < 5e-6


def test_split_bf16_transposed_conv_non_finite_inputs_propagate_like_the_fp32_kernel():
    B, K, M, H, W = 2, 32, 64, 8, 32
    x = synth.normal((B, K, H, W), 't6.nx').to(DEV)
    w = (synth.normal((M, K, 3, 3), 't6.nw') / (3 * math.sqrt(K))).to(DEV)
    w = torch.where(w.abs() < 1e-3, torch.full_like(w, 1e-3), w)
    clean = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD), _lib.CONV_T2, M, H, W)
    xp = x.clone()
    for b, k, y, xx, v in [(0, 3, 5, 17, float('inf')), (0, 7, 0, 0, float('-inf')), (1, 30, 7, 31, float('nan')), (1, 0, 3, 15, float('inf')),
                           (0, 16, 4, 16, float('nan'))]:
        xp[b, k, y, xx] = v
    got = _lib.conv(xp, _lib.conv_pack(w, _lib.PACK_T6FWD), _lib.CONV_T2S6, M, H, W)
    direct = _lib.conv(xp, _lib.conv_pack(w, _lib.PACK_FWD), _lib.CONV_T2, M, H, W)
    expect = ~torch.isfinite(direct)
    assert int(expect.sum()) > 0
    assert torch.equal(~torch.isfinite(got), expect)
    assert rel_err(got[~expect], clean[~expect]) < 5e-6


def test_split_bf16_transposed_selection():
    from transeditor_amd.op import modconv
    e = torch.empty
    assert modconv.fwd_kinds('up', 16, e(128, 256, 3, 3), 128, 128) == (_lib.PACK_T6FWD, _lib.CONV_T2S6)
    assert modconv.bwd_kinds('down', 32, e(256, 128, 3, 3), 128, 128) == (_lib.PACK_T6SWAP, _lib.CONV_T2S6)
    assert modconv.fwd_kinds('up', 16, e(512, 512, 3, 3), 8, 8) == (_lib.PACK_FWD, _lib.CONV_T2)
    old = modconv.USE_SPLIT_T2
    try:
        modconv.USE_SPLIT_T2 = False
        assert modconv.fwd_kinds('up', 16, e(128, 256, 3, 3), 128, 128) == (_lib.PACK_FWD, _lib.CONV_T2)
    finally:
        modconv.USE_SPLIT_T2 = old


@pytest.mark.parametrize('B,K,M,H,W', [(2, 32, 64, 8, 16), (3, 96, 192, 24, 32), (1, 48, 64, 16, 48), (2, 64, 128, 64, 64), (2, 160, 128, 72, 80),
                                       (1, 512, 512, 16, 16)])
@pytest.mark.parametrize('styled', [True, False])
def test_last_row_and_column_kernel_with_and_without_the_column_scratch(B, K, M, H, W, styled):
    """the last output row / column of TE_CONV_T2S6 (t2_edge_kernel, round 6): with the optional scratch (te_conv_t2s6_ws_floats) the
    body kernel hands it the scaled last input column, without it (ws = NULL, legal) it gathers the column from the input - the same
    products in the same order: identical bits; both tile widths (lines of >= 64 / < 64 cells), 16-channel tails of the 32-channel
    chunks (K = 48, 96, 160), and the lines against fp64 on their own"""
    x = synth.normal((B, K, H, W), f't6.ex.{K}.{H}').to(DEV)
    w = (synth.normal((M, K, 3, 3), f't6.ew.{M}.{K}') / (3 * math.sqrt(K))).to(DEV)
    isc = (1 + 0.3 * synth.normal((B, K), 't6.eisc')).to(DEV) if styled else None
    osc = (1 + 0.3 * synth.normal((B, M), 't6.eosc')).to(DEV)
    bias = synth.normal((M,), 't6.eb').to(DEV)
    wp = _lib.conv_pack(w, _lib.PACK_T6FWD, 0.9)
    assert _lib.lib().te_conv_t2s6_ws_floats(B, K, H) == B * K * H
    got = _lib.conv(x, wp, _lib.CONV_T2S6, M, H, W, isc, osc, bias, 3)
    bare = torch.empty_like(got)
    rc = _lib.lib().te_conv_res_f32(bare.data_ptr(), None, x.data_ptr(), wp.data_ptr(), isc.data_ptr() if styled else None, osc.data_ptr(),
                                   bias.data_ptr(), None, None, 1.0, 3, _lib.CONV_T2S6, B, K, M, H, W, _lib._stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(got, bare)
    xs = x.double() * (isc.double()[:, :, None, None] if styled else 1.0)
    want = F.conv_transpose2d(xs, (w.double() * 0.9).transpose(0, 1), stride=2) * osc.double()[:, :, None, None] + bias.double()[None, :, None, None]
    want = F.leaky_relu(want, 0.2) * math.sqrt(2)
    assert rel_err(got[:, :, 2 * H], want[:, :, 2 * H]) < 5e-6
    assert rel_err(got[:, :, :, 2 * W], want[:, :, :, 2 * W]) < 5e-6
    assert rel_err(got, want) < 5e-6


@pytest.mark.parametrize('B,K,M,H,W', [(2, 32, 128, 8, 16), (3, 96, 256, 24, 32), (1, 48, 128, 16, 48), (2, 160, 128, 8, 16), (1, 512, 512, 16, 16),
                                       (2, 128, 256, 32, 64), (5, 32, 384, 8, 32), (2, 64, 64, 8, 16), (3, 64, 192, 8, 32)])
def test_split_bf16_transposed_kernel_forms_are_bit_identical(B, K, M, H, W):
    """two-image form (t2s6q_kernel, round 6, default where M % 128 == 0 and every CU gets a block) and ping-pong form (t2s6_kernel) of
    TE_CONV_T2S6: the same products in the same order per output element - identical bits, with style scales and the epilogue stages, on
    single- and multi-tile images, 2 - 32 channel stages, 1 - 4 blocks of 128 output channels (form 2 = the two-image kernel whatever
    the grid size; it runs the ping-pong kernel where M % 128 != 0)"""
    x = synth.normal((B, K, H, W), f't6.fx.{K}.{H}').to(DEV)
    w = (synth.normal((M, K, 3, 3), f't6.fw.{M}.{K}') / (3 * math.sqrt(K))).to(DEV)
    isc, osc = (1 + 0.3 * synth.normal((B, K), 't6.fi')).to(DEV), (1 + 0.3 * synth.normal((B, M), 't6.fo')).to(DEV)
    bias = synth.normal((M,), 't6.fb').to(DEV)
    u6 = _lib.conv_pack(w, _lib.PACK_T6FWD, 0.9)
    out = {}
    old = _lib.t2s6_form(-1)
    try:
        for form in (0, 2):
            _lib.t2s6_form(form)
            out[form] = (_lib.conv(x, u6, _lib.CONV_T2S6, M, H, W, isc, osc, bias, 3), _lib.conv(x, u6, _lib.CONV_T2S6, M, H, W, None, None, bias, 4),
                         _lib.conv(x, u6, _lib.CONV_T2S6, M, H, W))
    finally:
        _lib.t2s6_form(old)
    assert _lib.t2s6_form(-1) == old
    for a, b in zip(out[0], out[2]):
        assert torch.equal(a, b)
    want = F.conv_transpose2d(x.double(), (w.double() * 0.9).transpose(0, 1), stride=2)
    assert rel_err(out[2][2], want) < 5e-6
