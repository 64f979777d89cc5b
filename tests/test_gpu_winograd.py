"""TE_CONV_3X3W: the 1-D Winograd F(2,3) form of the 3x3 / stride 1 / pad 1 convolution (csrc/wino.hip) against fp64 torch and
against the direct kernel - forward layout and data-gradient layout (flipped, transposed taps), style scale at staging, every
epilogue stage (demodulation scale, bias, leaky-ReLU with gain sqrt(2) / 1, residual, activation-gradient mask), single-tile
and multi-tile images, several M blocks, edge tiles on all four sides - and the weight transform of the packing kernel.
Reference: the grouped F.conv2d of ModulatedConv2d.forward (model_spatial_query.py:331-333), EqualConv2d.forward (:173-181)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from transeditor_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'

SHAPES = [(2, 8, 128, 4, 32), (1, 16, 128, 8, 64), (3, 24, 256, 12, 32), (2, 128, 128, 32, 32), (2, 64, 384, 16, 96), (1, 512, 512, 32, 32),
          # 64- and 32-row M blocks (8 / 16 rows per tile): the narrow layers of the FFHQ-1024 tail
          (2, 16, 64, 8, 32), (1, 64, 64, 32, 64), (2, 32, 32, 16, 32), (1, 64, 32, 64, 64), (1, 32, 96, 32, 32), (1, 8, 192, 24, 32)]


@pytest.mark.parametrize('B,K,M,H,W', SHAPES)
def test_winograd_forward_and_data_gradient_vs_fp64(B, K, M, H, W):
    assert _lib.wino_ok(B, K, M, H, W)
    x = synth.normal((B, K, H, W), f'wino.x.{K}.{H}').to(DEV)
    w = (synth.normal((M, K, 3, 3), f'wino.w.{M}.{K}') / (3 * math.sqrt(K))).to(DEV)
    isc = (1 + 0.3 * synth.normal((B, K), 'wino.isc')).to(DEV)
    ws = 0.83
    want = F.conv2d(x.double() * isc.double()[:, :, None, None], w.double() * ws, padding=1)
    up = _lib.conv_pack(w, _lib.PACK_WFWD, ws)
    got = _lib.conv(x, up, _lib.CONV_3X3W, M, H, W, isc)
    assert rel_err(got, want) < 5e-6
    direct = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD, ws), _lib.CONV_3X3, M, H, W, isc)
    assert rel_err(got, direct) < 5e-6
    # data gradient: a convolution from M to K channels with the flipped, transposed taps - only where THAT problem is covered
    if _lib.wino_ok(B, M, K, H, W):
        g = synth.normal((B, M, H, W), f'wino.g.{M}.{H}').to(DEV)
        osc = (1 + 0.3 * synth.normal((B, K), 'wino.osc')).to(DEV)
        want_g = F.conv_transpose2d(g.double(), w.double() * ws, padding=1) * osc.double()[:, :, None, None]
        got_g = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_WDGRAD, ws), _lib.CONV_3X3W, K, H, W, None, osc)
        assert rel_err(got_g, want_g) < 5e-6


@pytest.mark.parametrize('act', [0, 3, 4])
@pytest.mark.parametrize('epi', ['plain', 'res', 'res+mask'])
def test_winograd_epilogue_stages_equal_the_direct_kernel(act, epi):
    B, K, M, H, W = 2, 32, 256, 8, 64
    x = synth.normal((B, K, H, W), 'wino.ex').to(DEV)
    w = (synth.normal((M, K, 3, 3), 'wino.ew') / (3 * math.sqrt(K))).to(DEV)
    isc, osc = (1 + 0.3 * synth.normal((B, K), 'wino.ei')).to(DEV), (1 + 0.3 * synth.normal((B, M), 'wino.eo')).to(DEV)
    if epi != 'plain':          # (the direct kernel has the residual / mask stages for unmodulated launches only: the discriminator's)
        isc, osc = None, None
    bias = synth.normal((M,), 'wino.eb').to(DEV)
    res = synth.normal((B, M, H, W), 'wino.er').to(DEV) if epi != 'plain' else None
    mref = synth.normal((B, M, H, W), 'wino.em').to(DEV) if epi == 'res+mask' else None
    a = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_WFWD), _lib.CONV_3X3W, M, H, W, isc, osc, bias, act, res=res, mask_ref=mref, mask_gain=1.3)
    b = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD), _lib.CONV_3X3, M, H, W, isc, osc, bias, act, res=res, mask_ref=mref, mask_gain=1.3)
    one = lambda t, n: t.double() if t is not None else torch.ones(B, n, device=DEV, dtype=torch.float64)
    pre = F.conv2d(x.double() * one(isc, K)[:, :, None, None], w.double(), padding=1) * one(osc, M)[:, :, None, None] + bias.double()[None, :, None, None]
    if act:
        # a pre-activation within round-off of the kink may take the other slope in the two kernels: compare away from it
        keep = pre.abs() > 1e-4
        assert rel_err(a * keep, b * keep) < 5e-6
    else:
        assert rel_err(a, b) < 5e-6


def test_winograd_layouts_through_the_multi_pack_launch_and_the_module_path():
    """both Winograd layouts through te_conv_pack_weights_multi_f32 equal the single-layout launch and the transform written out
    with torch; and the module path really takes the Winograd kernel where it applies (same result as with the switch off)"""
    from transeditor_amd.op import modconv
    M, K = 256, 136
    w = synth.normal((M, K, 3, 3), 'wino.pw').to(DEV)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], device=DEV)
    u = torch.einsum('cx,mkyx->yckm', G, w * 0.7)                                   # [3][4][K][M]
    want_f = u.reshape(3, 4, K // 8, 8, M).permute(2, 0, 1, 3, 4).reshape(-1)
    wd = torch.flip(w, [2, 3]).transpose(0, 1)                                      # [K (as M), M (as K), 3, 3]
    ud = torch.einsum('cx,mkyx->yckm', G, wd * 0.7)
    want_d = ud.reshape(3, 4, M // 8, 8, K).permute(2, 0, 1, 3, 4).reshape(-1)
    assert rel_err(_lib.conv_pack(w, _lib.PACK_WFWD, 0.7), want_f) < 1e-6
    assert rel_err(_lib.conv_pack(w, _lib.PACK_WDGRAD, 0.7), want_d) < 1e-6
    a, b = _lib.conv_pack2(w, _lib.PACK_WFWD, _lib.PACK_WDGRAD, 0.7)
    assert torch.equal(a, _lib.conv_pack(w, _lib.PACK_WFWD, 0.7)) and torch.equal(b, _lib.conv_pack(w, _lib.PACK_WDGRAD, 0.7))
    # module path: forward + backward of a modulated layer with and without the Winograd form
    x = synth.normal((2, 128, 32, 32), 'wino.mx').to(DEV).requires_grad_(True)
    wm = (synth.normal((128, 128, 3, 3), 'wino.mw')).to(DEV).requires_grad_(True)
    s = (1 + 0.3 * synth.normal((2, 128), 'wino.ms')).to(DEV).requires_grad_(True)
    bias = synth.normal((128,), 'wino.mb').to(DEV).requires_grad_(True)
    gy = synth.normal((2, 128, 32, 32), 'wino.mg').to(DEV)
    outs = []
    for flag in (True, False):
        modconv.USE_WINOGRAD = flag
        try:
            y = modconv.modconv(x, wm, s, None, bias, True, '3x3', 1 / math.sqrt(128 * 9), demod_eps=1e-8)
            # (no upstream gradient near the leaky-ReLU kink: the two kernels may legitimately pick different slopes there)
            gs = torch.autograd.grad(y, (x, wm, s, bias), gy * (y.detach().abs() > 1e-4))
        finally:
            modconv.USE_WINOGRAD = True
        outs.append((y.detach(),) + gs)
    for name, p, q in zip(('y', 'dx', 'dw', 'ds', 'db'), outs[0], outs[1]):
        assert rel_err(p, q) < 2e-5, name


# ---- the pair form (1-D Winograd F(3,2)) of the 3x3 weight gradient (csrc/wgrad.hip, wgrad_mfma_kernel<3X3, 4, WINO>)
PAIR_SHAPES = [  # B, Ci, Co, H, W: full 32-cell rows (16-byte staging), ragged widths / heights, 1- and 2-pixel-wide images, channel tails
    (2, 128, 128, 64, 64), (1, 96, 160, 40, 40), (3, 72, 200, 37, 45), (2, 130, 66, 5, 3), (2, 256, 128, 9, 1), (1, 513, 512, 4, 4),
    (2, 128, 192, 2, 2), (4, 512, 512, 8, 8)]


@pytest.mark.parametrize('B,Ci,Co,H,W', PAIR_SHAPES)
def test_weight_gradient_pair_form_vs_fp64(B, Ci, Co, H, W):
    """slabs of the pair form, summed, against the fp64 correlation (the weight gradient of F.conv2d, model_spatial_query.py:331-335)"""
    assert _lib.wgrad_pair_form(_lib.CONV_3X3, Co, Ci, H, W)
    g = synth.normal((B, Co, H, W), f'pair.g.{Co}.{H}.{W}').to(DEV)
    x = synth.normal((B, Ci, H, W), f'pair.x.{Ci}.{H}.{W}').to(DEV)
    slabs = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W)
    got = slabs.double().sum(dim=(0, 1))
    want = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, 3, 3), g.double(), padding=1).reshape(Co, Ci, 9)
    assert rel_err(got, want) < 5e-6
    # per sample: every slab group is the correlation of ITS sample (the reducer derives d style / d demod from them)
    per = slabs.double().sum(dim=1)
    for b in range(B):
        wb = torch.nn.grad.conv2d_weight(x[b:b + 1].double(), (Co, Ci, 3, 3), g[b:b + 1].double(), padding=1).reshape(Co, Ci, 9)
        assert rel_err(per[b], wb) < 5e-6
    # grouped form (plain gradient of small images: NB samples share a slab)
    grouped = _lib.wgrad_slabs(g, x, _lib.CONV_3X3, H, W, group=True)
    assert rel_err(grouped.double().sum(dim=(0, 1)), want) < 5e-6


def test_weight_gradient_pair_form_selection():
    """8-wave tile only (one of the channel counts above 64), 3x3 only"""
    assert _lib.wgrad_pair_form(_lib.CONV_3X3, 128, 128, 256, 256) and _lib.wgrad_pair_form(_lib.CONV_3X3, 32, 96, 7, 5)
    assert not _lib.wgrad_pair_form(_lib.CONV_3X3, 64, 64, 512, 512)
    assert not _lib.wgrad_pair_form(_lib.CONV_T2, 128, 128, 64, 64) and not _lib.wgrad_pair_form(_lib.CONV_1X1, 128, 128, 64, 64)
    assert not _lib.wgrad_pair_form(_lib.CONV_3X3, 128, 128, 1, 1)        # a stage of 2 cells: no pair couple


# ---- TE_CONV_3X3W6: the same Winograd form with its products on the bf16 matrix pipe (csrc/wino6.hip): every fp32 operand split into
# three bf16 pieces, six exact piece products accumulated in fp32.  Pinned at the SAME bar as the fp32 kernels (5e-6 against fp64), and
# the deviation from fp64 must not exceed the fp32 direct kernel's by more than a rounding unit: it is fp32 arithmetic, not bf16.
W6_SHAPES = [(2, 32, 64, 8, 32), (1, 64, 128, 16, 64), (3, 96, 192, 24, 32), (2, 128, 64, 32, 96), (1, 32, 128, 40, 96), (2, 512, 512, 32, 32),
             (1, 64, 64, 64, 64), (4, 128, 128, 64, 64),
             # round 6: 16-column images, two samples side by side in a tile row (wino6p_kernel, lgpw = 3) - the 16 x 16 layers of both networks
             # (taken from half a block per CU up: hence the batches)
             (128, 32, 64, 16, 16), (128, 64, 128, 8, 16), (86, 96, 64, 24, 16), (16, 512, 512, 16, 16)]


@pytest.mark.parametrize('B,K,M,H,W', W6_SHAPES)
def test_split_bf16_winograd_forward_and_data_gradient_vs_fp64(B, K, M, H, W):
    assert _lib.wino6_ok(B, K, M, H, W)
    x = synth.normal((B, K, H, W), f'w6.x.{K}.{H}').to(DEV)
    w = (synth.normal((M, K, 3, 3), f'w6.w.{M}.{K}') / (3 * math.sqrt(K))).to(DEV)
    isc = (1 + 0.3 * synth.normal((B, K), 'w6.isc')).to(DEV)
    ws = 0.83
    want = F.conv2d(x.double() * isc.double()[:, :, None, None], w.double() * ws, padding=1)
    got = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_W6FWD, ws), _lib.CONV_3X3W6, M, H, W, isc)
    direct = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD, ws), _lib.CONV_3X3, M, H, W, isc)
    e_split, e_direct = rel_err(got, want), rel_err(direct, want)
    l2 = lambda a: float((a.double() - want).norm() / want.norm())
    msg = f'split-bf16 winograd {K}->{M} @{H}x{W}: max {e_split:.2e} (direct fp32 kernel {e_direct:.2e}), L2 {l2(got):.2e} ({l2(direct):.2e})'
    assert e_split < 5e-6
    # fp32-equivalent: the deviation from fp64 is that of the fp32 Winograd kernel (same transforms, fp32 MFMA chain) up to noise, and
    # within 2.5x of the direct fp32 kernel's (the Winograd transforms themselves cost up to ~2x at 512 channels, in both forms)
    if _lib.wino_ok(B, K, M, H, W):
        w32 = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_WFWD, ws), _lib.CONV_3X3W, M, H, W, isc)
        msg += f'; fp32 winograd kernel L2 {l2(w32):.2e}'
        assert l2(got) < 1.25 * l2(w32) + 1e-7, msg
    print(msg)
    assert l2(got) < 2.5 * l2(direct) + 1e-7, msg
    if _lib.wino6_ok(B, M, K, H, W):
        g = synth.normal((B, M, H, W), f'w6.g.{M}.{H}').to(DEV)
        osc = (1 + 0.3 * synth.normal((B, K), 'w6.osc')).to(DEV)
        want_g = F.conv_transpose2d(g.double(), w.double() * ws, padding=1) * osc.double()[:, :, None, None]
        got_g = _lib.conv(g, _lib.conv_pack(w, _lib.PACK_W6DGRAD, ws), _lib.CONV_3X3W6, K, H, W, None, osc)
        assert rel_err(got_g, want_g) < 5e-6


@pytest.mark.parametrize('W', [64, 16])
@pytest.mark.parametrize('act', [0, 3, 4])
@pytest.mark.parametrize('epi', ['plain', 'res', 'res+mask'])
def test_split_bf16_winograd_epilogue_stages_equal_the_direct_kernel(act, epi, W):
    B, K, M, H = (2 if W >= 32 else 64), 32, 128, 16
    x = synth.normal((B, K, H, W), 'w6.ex').to(DEV)
    w = (synth.normal((M, K, 3, 3), 'w6.ew') / (3 * math.sqrt(K))).to(DEV)
    isc, osc = (1 + 0.3 * synth.normal((B, K), 'w6.ei')).to(DEV), (1 + 0.3 * synth.normal((B, M), 'w6.eo')).to(DEV)
    if epi != 'plain':
        isc, osc = None, None
    bias = synth.normal((M,), 'w6.eb').to(DEV)
    res = synth.normal((B, M, H, W), 'w6.er').to(DEV) if epi != 'plain' else None
    mref = synth.normal((B, M, H, W), 'w6.em').to(DEV) if epi == 'res+mask' else None
    a = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_W6FWD), _lib.CONV_3X3W6, M, H, W, isc, osc, bias, act, res=res, mask_ref=mref, mask_gain=1.3)
    b = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD), _lib.CONV_3X3, M, H, W, isc, osc, bias, act, res=res, mask_ref=mref, mask_gain=1.3)
    # (elements whose pre-activation sits within round-off of the leaky-ReLU kink may take different slopes in the two kernels)
    one = lambda t, n: t.double() if t is not None else torch.ones(B, n, device=DEV, dtype=torch.float64)
    pre = F.conv2d(x.double() * one(isc, K)[:, :, None, None], w.double(), padding=1) * one(osc, M)[:, :, None, None] + bias.double()[None, :, None, None]
    keep = (pre.abs() > 1e-5) if act else torch.ones_like(pre, dtype=torch.bool)
    assert rel_err(a * keep, b * keep) < 5e-6


def test_split_bf16_layouts_through_the_multi_pack_launch_and_selection():
    """the split layouts through te_conv_pack_weights2_f32 equal the single-layout launch; the three pieces of an element add up to
    the fp32 Winograd weight (to 2^-24); the module path takes the split kernel exactly where te_conv_wino6_supported says so"""
    from transeditor_amd.op import modconv
    M, K = 128, 96
    w = synth.normal((M, K, 3, 3), 'w6.pw').to(DEV)
    a, b = _lib.conv_pack2(w, _lib.PACK_W6FWD, _lib.PACK_W6DGRAD, 0.7)
    assert torch.equal(a, _lib.conv_pack(w, _lib.PACK_W6FWD, 0.7)) and torch.equal(b, _lib.conv_pack(w, _lib.PACK_W6DGRAD, 0.7))
    # U6[K/16][piece][ky][c][M/32][lane = m % 32 + 32 * (k % 16 / 8)][k % 8]  (bf16)  ->  sum of pieces, reordered to [ky][c][k][m]
    pieces = a.view(torch.bfloat16).float().view(K // 16, 3, 3, 4, M // 32, 2, 32, 8)       # [stage][piece][ky][c][mt][kh][m][j]
    u = pieces.sum(dim=1).permute(1, 2, 0, 4, 6, 3, 5).reshape(3, 4, K, M)                  # [ky][c][stage][kh][j][mt][m] -> [ky][c][k][m]
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], device=DEV)
    want = torch.einsum('cx,mkyx->yckm', G, w * 0.7)
    assert rel_err(u, want) < 2e-7
    assert modconv.fwd_kinds('3x3', 16, torch.empty(128, 128, 3, 3), 256, 256) == (_lib.PACK_W6FWD, _lib.CONV_3X3W6)
    assert modconv.bwd_kinds('3x3', 16, torch.empty(512, 256, 3, 3), 64, 64) == (_lib.PACK_W6DGRAD, _lib.CONV_3X3W6)
    assert modconv.fwd_kinds('3x3', 4, torch.empty(32, 32, 3, 3), 1024, 1024) == (_lib.PACK_WFWD, _lib.CONV_3X3W)       # M % 64 != 0: fp32 form
    assert modconv.fwd_kinds('3x3', 16, torch.empty(512, 512, 3, 3), 4, 4) == (_lib.PACK_FWD, _lib.CONV_3X3)


# ---- round 5: range and non-finite behaviour of the split kernel, and the two kernel forms

@pytest.mark.parametrize('which', ['input', 'weight', 'both'])
@pytest.mark.parametrize('scale', [1e-30, 1e-15, 1e+15, 1e+30])
def test_split_bf16_winograd_scale_sweep(scale, which):
    """The three-piece split keeps fp32's RANGE, not only its precision at unit scale: tensors scaled by 1e-30 ... 1e+30 (inputs, weights,
    or the square roots of the factor on both) give the scaled result at the same 5e-6 bar against fp64 and against the direct fp32
    kernel.  (x = h + m + l with l ~ 2^-17 x: the last piece is a normal bf16 number down to |x| ~ 1e-33; elements below that lose
    bits of an already negligible contribution.)"""
    B, K, M, H, W = 2, 64, 128, 16, 64
    x = synth.normal((B, K, H, W), 'w6.sx').to(DEV)
    w = (synth.normal((M, K, 3, 3), 'w6.sw') / (3 * math.sqrt(K))).to(DEV)
    isc = (1 + 0.3 * synth.normal((B, K), 'w6.si')).to(DEV)
    sx, sw = {'input': (scale, 1.0), 'weight': (1.0, scale), 'both': (math.sqrt(scale), math.sqrt(scale))}[which]
    xs, ws_ = (x * sx).contiguous(), (w * sw).contiguous()
    want = F.conv2d(xs.double() * isc.double()[:, :, None, None], ws_.double(), padding=1)
    got = _lib.conv(xs, _lib.conv_pack(ws_, _lib.PACK_W6FWD), _lib.CONV_3X3W6, M, H, W, isc)
    direct = _lib.conv(xs, _lib.conv_pack(ws_, _lib.PACK_FWD), _lib.CONV_3X3, M, H, W, isc)
    assert torch.isfinite(got).all()
    l2 = lambda a: float((a.double() - want).norm() / want.norm())
    print(f'split-bf16 scale sweep {which} x {scale:g}: L2 vs fp64 {l2(got):.2e} (direct fp32 kernel {l2(direct):.2e}), max {rel_err(got, want):.2e}')
    assert rel_err(got, want) < 5e-6
    assert l2(got) < 2.5 * l2(direct) + 1e-7


def test_split_bf16_winograd_tiny_magnitudes_degrade_gracefully():
    """below the range the last piece can hold (|x| ~ 1e-33 ... 1e-38: l, then m, become subnormal / zero) the kernel degrades to a
    two-piece, then one-piece product - it never produces non-finite values or garbage: relative error bounded by 2^-15 at 1e-36."""
    B, K, M, H, W = 1, 32, 64, 8, 32
    x = synth.normal((B, K, H, W), 'w6.tx').to(DEV)
    w = (synth.normal((M, K, 3, 3), 'w6.tw') / (3 * math.sqrt(K))).to(DEV)
    for s, bar in ((1e-33, 1e-5), (1e-36, 2 ** -15)):
        xs = (x * s).contiguous()
        want = F.conv2d(xs.double(), w.double(), padding=1)
        got = _lib.conv(xs, _lib.conv_pack(w, _lib.PACK_W6FWD), _lib.CONV_3X3W6, M, H, W)
        assert torch.isfinite(got).all()
        e = float((got.double() - want).norm() / want.norm())
        print(f'split-bf16 at input scale {s:g}: L2 vs fp64 {e:.2e}')
        assert e < bar


@pytest.mark.parametrize('form,M', [(0, 64), (1, 64), (3, 128)])
def test_split_bf16_winograd_non_finite_inputs_propagate_like_the_direct_kernel(form, M):
    """Inf / NaN in the input reach exactly the outputs whose 3x3 windows contain them - the set the direct fp32 kernel marks - and
    nothing else; every other output is unaffected (same value as without the planted element, 5e-6).  WHAT the marked outputs hold
    differs by design: the direct kernel gives w * inf = +-inf where the split kernel gives NaN (inf = h, inf - h = NaN is the
    second piece; and Winograd's t = d0 - d2 may cancel infinities) - a non-finite value either way, documented in te_hip.h."""
    B, K, H, W = 2, 32, 16, 64
    x = synth.normal((B, K, H, W), 'w6.nx').to(DEV)
    w = (synth.normal((M, K, 3, 3), 'w6.nw') / (3 * math.sqrt(K))).to(DEV)
    w = torch.where(w.abs() < 1e-3, torch.full_like(w, 1e-3), w)        # no exact zeros / tiny taps: w * inf is +-inf in the direct kernel
    clean = _lib.conv(x, _lib.conv_pack(w, _lib.PACK_FWD), _lib.CONV_3X3, M, H, W)
    plants = [(0, 3, 5, 17, float('inf')), (0, 7, 0, 0, float('-inf')), (1, 30, 15, 63, float('nan')), (1, 0, 8, 31, float('inf')),
              (0, 16, 7, 32, float('nan')), (1, 5, 15, 0, float('-inf'))]     # interior, corners, tile borders (x = 31 | 32, y = 7 | 8)
    xp = x.clone()
    for b, k, y, xx, v in plants:
        xp[b, k, y, xx] = v
    old = _lib.wino6_form(form)
    try:
        got = _lib.conv(xp, _lib.conv_pack(w, _lib.PACK_W6FWD), _lib.CONV_3X3W6, M, H, W)
    finally:
        _lib.wino6_form(old)
    direct = _lib.conv(xp, _lib.conv_pack(w, _lib.PACK_FWD), _lib.CONV_3X3, M, H, W)
    bad_direct, bad_got = ~torch.isfinite(direct), ~torch.isfinite(got)
    expect = torch.zeros(B, 1, H, W, dtype=torch.bool, device=DEV)
    for b, k, y, xx, v in plants:
        expect[b, 0, max(0, y - 1):y + 2, max(0, xx - 1):xx + 2] = True
    expect = expect.expand(B, M, H, W)
    assert torch.equal(bad_direct, expect)              # (the yardstick itself)
    assert torch.equal(bad_got, expect), f'{int((bad_got ^ expect).sum())} outputs differ in finiteness'
    fin = ~expect
    assert rel_err(got[fin], clean[fin]) < 5e-6


@pytest.mark.parametrize('B,K,M,H,W', W6_SHAPES + [(2, 160, 64, 8, 64), (1, 96, 64, 32, 32), (2, 32, 128, 8, 32), (3, 64, 256, 16, 64),
                                             (1, 96, 128, 24, 32), (5, 32, 384, 8, 32), (2, 160, 128, 40, 96)])
def test_split_bf16_kernel_forms_are_bit_identical(B, K, M, H, W):
    """two-image (round 6, default where M % 128 == 0), ping-pong (round 5) and block-phase (round 4) forms of TE_CONV_3X3W6 issue the
    same products in the same order per output element: identical bits, with every epilogue stage, on single- and multi-tile images,
    2 - 32 channel stages, 1 - 3 blocks of 128 output channels (form 3 = the two-image kernel whatever the grid size; it runs
    the ping-pong kernel where M % 128 != 0)"""
    x = synth.normal((B, K, H, W), f'w6.fx.{K}.{H}').to(DEV)
    w = (synth.normal((M, K, 3, 3), f'w6.fw.{M}.{K}') / (3 * math.sqrt(K))).to(DEV)
    isc, osc = (1 + 0.3 * synth.normal((B, K), 'w6.fi')).to(DEV), (1 + 0.3 * synth.normal((B, M), 'w6.fo')).to(DEV)
    bias = synth.normal((M,), 'w6.fb').to(DEV)
    res, mref = synth.normal((B, M, H, W), 'w6.fr').to(DEV), synth.normal((B, M, H, W), 'w6.fm').to(DEV)
    u6 = _lib.conv_pack(w, _lib.PACK_W6FWD, 0.9)
    out = {}
    old = _lib.wino6_form(-1)
    try:
        for form in (0, 1, 3):
            _lib.wino6_form(form)
            out[form] = (_lib.conv(x, u6, _lib.CONV_3X3W6, M, H, W, isc, osc, bias, 3),
                         _lib.conv(x, u6, _lib.CONV_3X3W6, M, H, W, None, None, bias, 4, res=res, mask_ref=mref, mask_gain=1.3),
                         _lib.conv(x, u6, _lib.CONV_3X3W6, M, H, W))
    finally:
        _lib.wino6_form(old)
    assert _lib.wino6_form(-1) == old
    for a, b, c in zip(out[0], out[1], out[3]):
        assert torch.equal(a, b) and torch.equal(a, c)
