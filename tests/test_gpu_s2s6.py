"""TE_CONV_S2S6: the 3x3 / stride 2 / pad 0 convolution on the bf16 matrix pipe (csrc/s2s6.hip: three bf16 pieces per fp32 operand, six
exact piece products per multiply-add, fp32 accumulation) against fp64 torch and against the fp32 kernel of the same kind (TE_CONV_S2) -
both weight layouts (forward of the discriminator's down-sampling convolutions, model_spatial_query.py:765-779; data gradient of the
generator's up-sampling layers = adjoint of conv_transpose2d(stride 2), :318), style scale at staging, every epilogue stage, single- and
multi-tile images, 2 - 32 channel stages, the range sweep and the non-finite behaviour documented in te_hip.h, and the selection rule.
Pinned at the bar of the fp32 kernels: 5e-6 against fp64."""
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
def test_split_bf16_strided_conv_vs_fp64(B, K, M, H, W):
    assert _lib.s2s6_ok(B, K, M, H, W)
    x = synth.normal((B, K, 2 * H + 1, 2 * W + 1), f's6.x.{K}.{H}').to(DEV)
    w = (synth.normal((M, K, 3, 3), f's6.w.{M}.{K}') / (3 * math.sqrt(K))).to(DEV)
    isc = (1 + 0.3 * synth.normal((B, K), 's6.isc')).to(DEV)
    ws = 0.83
    want = F.conv2d(x.double() * isc.double()[:, :, None, None], w.double() * ws, stride=2)
    got = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_S6FWD, ws), _lib.CONV_S2S6, M, H, W, isc)
    direct = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD, ws), _lib.CONV_S2, M, H, W, isc)
    l2 = lambda a: float((a.double() - want).norm() / want.norm())
    print(f'split-bf16 strided {K}->{M} @{H}x{W}: max {rel_err(got, want):.2e} (fp32 kernel {rel_err(direct, want):.2e}), L2 {l2(got):.2e} ({l2(direct):.2e})')
    assert rel_err(got, want) < 5e-6
    assert l2(got) < 2.5 * l2(direct) + 1e-7          # fp32-equivalent: the same yardstick as the split Winograd kernel
    # the launch as data gradient of the transposed kind: a convolution from M to K channels on the swapped layout
    if _lib.s2s6_ok(B, M, K, H, W):
        g = synth.normal((B, M, 2 * H + 1, 2 * W + 1), f's6.g.{M}.{H}').to(DEV)
        osc = (1 + 0.3 * synth.normal((B, K), 's6.osc')).to(DEV)
        want_g = F.conv2d(g.double(), (w.double() * ws).transpose(0, 1), stride=2) * osc.double()[:, :, None, None]
        got_g = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_S6SWAP, ws), _lib.CONV_S2S6, K, H, W, None, osc)
        assert rel_err(got_g, want_g) < 5e-6
        ref_g = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_SWAP, ws), _lib.CONV_S2, K, H, W, None, osc)
        assert rel_err(got_g, ref_g) < 5e-6


@pytest.mark.parametrize('act', [0, 3, 4])
@pytest.mark.parametrize('epi', ['plain', 'res', 'mask', 'res+mask'])
def test_split_bf16_strided_conv_epilogue_stages(act, epi):
    """demodulation scale, bias, leaky ReLU (gain sqrt(2) / 1), residual, activation-gradient mask: out = (act(osc * conv + bias) + res) *
    slope(mask) - against the fp32 kernel (which has the first three) with the last two applied to its output"""
    B, K, M, H, W = 2, 64, 128, 8, 32
    x = synth.normal((B, K, 2 * H + 1, 2 * W + 1), 's6.ex').to(DEV)
    w = (synth.normal((M, K, 3, 3), 's6.ew') / (3 * math.sqrt(K))).to(DEV)
    isc, osc = (1 + 0.3 * synth.normal((B, K), 's6.ei')).to(DEV), (1 + 0.3 * synth.normal((B, M), 's6.eo')).to(DEV)
    bias = synth.normal((M,), 's6.eb').to(DEV)
    res = synth.normal((B, M, H, W), 's6.er').to(DEV) if 'res' in epi else None
    mref = synth.normal((B, M, H, W), 's6.em').to(DEV) if 'mask' in epi else None
    a = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_S6FWD), _lib.CONV_S2S6, M, H, W, isc, osc, bias, act, res=res, mask_ref=mref, mask_gain=1.3)
    b = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD), _lib.CONV_S2, M, H, W, isc, osc, bias, act)
    if res is not None:
        b = b + res
    if mref is not None:
        b = b * torch.where(mref > 0, 1.3, 0.2 * 1.3)
    pre = F.conv2d(x.double() * isc.double()[:, :, None, None], w.double(), stride=2) * osc.double()[:, :, None, None] + bias.double()[None, :, None, None]
    keep = (pre.abs() > 1e-5) if act else torch.ones_like(pre, dtype=torch.bool)      # (a pre-activation at the kink may take either slope)
    assert rel_err(a * keep, b * keep) < 5e-6


@pytest.mark.parametrize('scale', [1e-30, 1e-15, 1e+15, 1e+30])
def test_split_bf16_strided_conv_scale_sweep(scale):
    B, K, M, H, W = 2, 64, 128, 8, 32
    x = (synth.normal((B, K, 2 * H + 1, 2 * W + 1), 's6.sx') * math.sqrt(scale)).to(DEV)
    w = (synth.normal((M, K, 3, 3), 's6.sw') / (3 * math.sqrt(K)) * math.sqrt(scale)).to(DEV)
    want = F.conv2d(x.double(), w.double(), stride=2)
    got = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_S6FWD), _lib.CONV_S2S6, M, H, W)
    assert torch.isfinite(got).all()
    assert rel_err(got, want) < 5e-6


def test_split_bf16_strided_conv_non_finite_inputs_propagate_like_the_fp32_kernel():
    """an Inf / NaN input element makes exactly the outputs whose window contains it non-finite (as NaN where the fp32 kernel gives
    +-Inf: te_hip.h); every other output is untouched"""
    B, K, M, H, W = 2, 32, 64, 8, 32
    x = synth.normal((B, K, 2 * H + 1, 2 * W + 1), 's6.nx').to(DEV)
    w = (synth.normal((M, K, 3, 3), 's6.nw') / (3 * math.sqrt(K))).to(DEV)
    w = torch.where(w.abs() < 1e-3, torch.full_like(w, 1e-3), w)
    clean = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD), _lib.CONV_S2, M, H, W)
    plants = [(0, 3, 5, 17, float('inf')), (0, 7, 0, 0, float('-inf')), (1, 30, 16, 64, float('nan')), (1, 0, 8, 32, float('inf')),
              (0, 16, 9, 33, float('nan'))]
    xp = x.clone()
    for b, k, y, xx, v in plants:
        xp[b, k, y, xx] = v
    got = _lib.conv(xp, _lib.conv_pack(w, _lib.PACK_S6FWD), _lib.CONV_S2S6, M, H, W)
    direct = _lib.conv(xp, _lib.conv_pack(w, _lib.PACK_FWD), _lib.CONV_S2, M, H, W)
    expect = ~torch.isfinite(direct)
    assert int(expect.sum()) > 0
    assert torch.equal(~torch.isfinite(got), expect)
    assert rel_err(got[~expect], clean[~expect]) < 5e-6


def test_split_bf16_strided_layout_and_selection():
    """the three pieces of a packed element add up to the weight (to 2^-24) in the documented fragment order; the module path takes
    the split kernel exactly where te_conv_s2s6_supported says so; TE_SPLIT_S2 / TE_SPLIT_BF16 switch it off"""
    from transeditor_amd.op import modconv
    Co, Ci = 64, 48
    w = (synth.normal((Co, Ci, 3, 3), 's6.lw')).to(DEV)
    u = _lib.conv_pack(w, _lib.PACK_S6FWD, 1.0).view(torch.int16)
    MT = Co // 32
    u = u[:27 * Ci * Co].view(Ci // 16, 3, 9, MT, 64, 8).to(torch.int32)
    pieces = ((u & 0xFFFF) << 16).view(torch.float32)                               # bf16 bits -> fp32
    tot = pieces.sum(dim=1)                                                          # [K/16, tap, MT, lane, 8]
    # lane = m % 32 + 32 * (k % 16 / 8), element = k % 8
    back = tot.view(Ci // 16, 9, MT, 2, 32, 8).permute(2, 4, 0, 3, 5, 1).reshape(Co, Ci, 9)
    assert rel_err(back, w.reshape(Co, Ci, 9)) < 2e-7
    e = torch.empty
    assert modconv.fwd_kinds('down', 32, e(256, 128, 3, 3), 128, 128) == (_lib.PACK_S6FWD, _lib.CONV_S2S6)
    assert modconv.bwd_kinds('up', 16, e(128, 256, 3, 3), 128, 128) == (_lib.PACK_S6SWAP, _lib.CONV_S2S6)
    assert modconv.fwd_kinds('down', 32, e(512, 512, 3, 3), 8, 8) == (_lib.PACK_FWD, _lib.CONV_S2)
    old = modconv.USE_SPLIT_S2
    try:
        modconv.USE_SPLIT_S2 = False
        assert modconv.fwd_kinds('down', 32, e(256, 128, 3, 3), 128, 128) == (_lib.PACK_FWD, _lib.CONV_S2)
    finally:
        modconv.USE_SPLIT_S2 = old


@pytest.mark.parametrize('B,K,M,H,W', [(2, 32, 128, 8, 16), (3, 96, 256, 24, 32), (1, 48, 128, 16, 48), (2, 160, 128, 8, 16), (1, 512, 512, 16, 16),
                                       (2, 128, 256, 32, 64), (5, 32, 384, 8, 32), (2, 64, 64, 8, 16), (3, 64, 192, 8, 32)])
def test_split_bf16_strided_kernel_forms_are_bit_identical(B, K, M, H, W):
    """two-image form (s2s6q_kernel, round 6, default where M % 128 == 0 and every CU gets a block) and ping-pong form (s2s6_kernel) of
    TE_CONV_S2S6 issue the same products in the same order per output element: identical bits, with style scales and every epilogue
    stage, on single- and multi-tile images, 2 - 32 channel stages, 1 - 4 blocks of 128 output channels (form 2 = the two-image kernel
    whatever the grid size; it runs the ping-pong kernel where M % 128 != 0)"""
    x = synth.normal((B, K, 2 * H + 1, 2 * W + 1), f's6.fx.{K}.{H}').to(DEV)
    w = (synth.normal((M, K, 3, 3), f's6.fw.{M}.{K}') / (3 * math.sqrt(K))).to(DEV)
    isc, osc = (1 + 0.3 * synth.normal((B, K), 's6.fi')).to(DEV), (1 + 0.3 * synth.normal((B, M), 's6.fo')).to(DEV)
    bias = synth.normal((M,), 's6.fb').to(DEV)
    res, mref = synth.normal((B, M, H, W), 's6.fr').to(DEV), synth.normal((B, M, H, W), 's6.fm').to(DEV)
    u6 = _lib.conv_pack(w, _lib.PACK_S6FWD, 0.9)
    out = {}
    old = _lib.s2s6_form(-1)
    try:
        for form in (0, 2):
            _lib.s2s6_form(form)
            out[form] = (_lib.conv(x, u6, _lib.CONV_S2S6, M, H, W, isc, osc, bias, 3),
                         _lib.conv(x, u6, _lib.CONV_S2S6, M, H, W, None, None, bias, 4, res=res, mask_ref=mref, mask_gain=1.3),
                         _lib.conv(x, u6, _lib.CONV_S2S6, M, H, W))
    finally:
        _lib.s2s6_form(old)
    assert _lib.s2s6_form(-1) == old
    for a, b in zip(out[0], out[2]):
        assert torch.equal(a, b)
    want = F.conv2d(x.double(), w.double() * 0.9, stride=2)
    assert rel_err(out[2][2], want) < 5e-6
