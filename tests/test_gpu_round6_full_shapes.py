"""The round-6 kernel forms at the FFHQ-256 / batch-16 layer shapes bench.py times (BASELINE configs[1] / [2]): the two-image forms of the
strided and transposed kinds (s2s6q_kernel / t2s6q_kernel + t2_edge_kernel) and the wide transposed-kind weight gradient (wgrad6tw_kernel)
give the bits of the forms they replace, and all of them - with the split 1x1 weight gradient (wgrad6p_kernel) - stay within the 5e-6 bar of
the fp32 MFMA kernels of the same kinds (whose own parity with the fp64 oracle is pinned at the small shapes and in the model-level tests).
Reference layers: model_spatial_query.py:310-321 (up-sampling convolution), :765-779 (down-sampling), :173-181 / :780-798 (ResBlock skip)."""
import math

import pytest
import torch

from conftest import rel_err
from transeditor_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# (B, K, M, H, W) on the low-resolution grid
CONV_SHAPES = [(16, 128, 256, 128, 128), (16, 256, 512, 64, 64), (16, 512, 512, 32, 32)]
T2_SHAPES = [(16, 256, 128, 128, 128), (16, 512, 256, 64, 64), (16, 512, 512, 32, 32), (16, 512, 512, 16, 16)]


@pytest.mark.parametrize('B,K,M,H,W', CONV_SHAPES)
def test_strided_kind_two_image_form_at_the_timed_shapes(B, K, M, H, W):
    x = synth.normal((B, K, 2 * H + 1, 2 * W + 1), f'r6.sx.{K}').to(DEV)
    w = (synth.normal((M, K, 3, 3), f'r6.sw.{M}') / (3 * math.sqrt(K))).to(DEV)
    isc, osc = (1 + 0.3 * synth.normal((B, K), 'r6.si')).to(DEV), (1 + 0.3 * synth.normal((B, M), 'r6.so')).to(DEV)
    bias = synth.normal((M,), 'r6.sb').to(DEV)
    u6 = _lib.conv_pack(w, _lib.PACK_S6FWD, 0.9)
    old = _lib.s2s6_form(0)
    try:
        a = _lib.conv(x, u6, _lib.CONV_S2S6, M, H, W, isc, osc, bias, 3)
        _lib.s2s6_form(1)
        b = _lib.conv(x, u6, _lib.CONV_S2S6, M, H, W, isc, osc, bias, 3)
    finally:
        _lib.s2s6_form(old)
    assert torch.equal(a, b)
    ref = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD, 0.9), _lib.CONV_S2, M, H, W, isc, osc, bias, 3)
    assert rel_err(b, ref) < 5e-6


@pytest.mark.parametrize('B,K,M,H,W', T2_SHAPES)
def test_transposed_kind_two_image_form_and_edge_kernel_at_the_timed_shapes(B, K, M, H, W):
    x = synth.normal((B, K, H, W), f'r6.tx.{K}.{H}').to(DEV)
    w = (synth.normal((M, K, 3, 3), f'r6.tw.{M}.{H}') / (3 * math.sqrt(K))).to(DEV)
    isc, osc = (1 + 0.3 * synth.normal((B, K), 'r6.ti')).to(DEV), (1 + 0.3 * synth.normal((B, M), 'r6.to')).to(DEV)
    bias = synth.normal((M,), 'r6.tb').to(DEV)
    u6 = _lib.conv_pack(w, _lib.PACK_T6FWD, 0.9)
    old = _lib.t2s6_form(0)
    try:
        a = _lib.conv(x, u6, _lib.CONV_T2S6, M, H, W, isc, osc, bias, 3)
        _lib.t2s6_form(1)
        b = _lib.conv(x, u6, _lib.CONV_T2S6, M, H, W, isc, osc, bias, 3)
    finally:
        _lib.t2s6_form(old)
    assert torch.equal(a, b)
    ref = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD, 0.9), _lib.CONV_T2, M, H, W, isc, osc, bias, 3)
    assert rel_err(b, ref) < 5e-6
    # the last output row and column on their own (t2_edge_kernel against the thin regions of the fp32 kernel)
    assert rel_err(b[:, :, 2 * H], ref[:, :, 2 * H]) < 5e-6
    assert rel_err(b[:, :, :, 2 * W], ref[:, :, :, 2 * W]) < 5e-6


@pytest.mark.parametrize('B,Co,Ci,H,W', [(16, 128, 256, 128, 128), (16, 256, 512, 64, 64), (16, 512, 512, 32, 32), (16, 512, 512, 16, 16)])
def test_transposed_kind_weight_gradient_wide_form_at_the_timed_shapes(B, Co, Ci, H, W):
    g = synth.normal((B, Co, 2 * H + 1, 2 * W + 1), f'r6.wg.{Co}.{H}').to(DEV)
    x = synth.normal((B, Ci, H, W), f'r6.wx.{Ci}.{H}').to(DEV)
    old_s, old_w = _lib.wgrad_split(1), _lib.wgrad_t2_wide(0)
    try:
        narrow = _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W)
        _lib.wgrad_t2_wide(1)
        wide = _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W)
        _lib.wgrad_split(0)
        ref = _lib.wgrad_slabs(g, x, _lib.CONV_T2, H, W)
    finally:
        _lib.wgrad_t2_wide(old_w)
        _lib.wgrad_split(old_s)
    assert torch.equal(narrow, wide)
    assert rel_err(wide.sum(1), ref.sum(1)) < 5e-6


@pytest.mark.parametrize('B,Co,Ci,H,W', [(32, 256, 128, 128, 128), (32, 512, 256, 64, 64), (32, 512, 512, 32, 32), (32, 512, 512, 16, 16)])
def test_1x1_weight_gradient_on_the_split_pipe_at_the_timed_shapes(B, Co, Ci, H, W):
    g = synth.normal((B, Co, H, W), f'r6.pg.{Co}.{H}').to(DEV)
    x = synth.normal((B, Ci, H, W), f'r6.px.{Ci}.{H}').to(DEV)
    old = _lib.wgrad_split(1)
    try:
        got = _lib.wgrad_slabs(g, x, _lib.CONV_1X1, H, W, group=True)
        _lib.wgrad_split(0)
        ref = _lib.wgrad_slabs(g, x, _lib.CONV_1X1, H, W, group=True)
    finally:
        _lib.wgrad_split(old)
    assert rel_err(got.sum((0, 1)), ref.sum((0, 1))) < 5e-6
