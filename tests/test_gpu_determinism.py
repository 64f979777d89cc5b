"""Bit-reproducible gradients: no atomic add is left on any path of libte_hip.so (the
library is built without -munsafe-fp-atomics); reductions that span thread blocks go through per-block partials in a
workspace and a fixed-order second pass.

* the reducers themselves, every dispatch path (fused single pass with and without chunk splitting, generic odd-shaped,
  ToRGB), against fp64 torch AND bit for bit between two runs;
* the bias gradient of the activation-gradient passes (streaming form with partials, per-channel forms);
* the 256-px generator backward, the 256-px discriminator step backward (joint batch) and a second-order step, each run
  twice: every gradient bit-identical.
"""
import math

import pytest
import torch

from conftest import rel_err
from transeditor_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _bits_equal(a, b):
    return (a is None and b is None) or torch.equal(a.view(torch.int32), b.view(torch.int32))


REDUCE_CASES = [   # B, S, Co, Ci, taps, (isc, osc)
    (4, 3, 64, 64, 9, (True, True)),        # fused single pass, one chunk group
    (16, 16, 32, 32, 9, (True, True)),      # fused, slab chunks split over blockIdx.z (narrow layer): dW through parts as well
    (3, 5, 128, 96, 9, (True, False)),      # fused, odd batch (tail sample), no demodulation
    (16, 1, 256, 256, 9, (True, True)),     # fused, round 6: ONE chunk per sample, the BATCH split over blockIdx.z (4 ranges of 4 samples)
    (16, 4, 128, 128, 9, (True, True)),     # fused, chunks AND batch split (4 x 4)
    (9, 2, 256, 256, 9, (True, True)),      # fused, odd batch split in two ranges (4 + 5 samples: a tail sample in one of them), 2 chunk ranges
    (5, 3, 128, 128, 9, (False, True)),     # fused, three chunk ranges of one chunk, no batch split
    (2, 2, 24, 20, 9, (True, True)),        # generic path (not multiples of 16 / 32)
    (5, 7, 40, 12, 1, (True, True)),        # generic, 1x1
    (8, 4, 512, 512, 1, (False, False)),    # plain 1x1 (discriminator skip): dW only, chunks over (b, s)
    (4, 64, 3, 128, 1, (True, False)),      # ToRGB path, chunks split over blockIdx.z
    (16, 2, 3, 512, 1, (True, False)),      # ToRGB path, one chunk group
]


@pytest.mark.parametrize('B,S,Co,Ci,taps,mods', REDUCE_CASES)
def test_wgrad_reduce_paths_vs_fp64_and_bit_reproducible(B, S, Co, Ci, taps, mods):
    slabs = synth.normal((B, S, Co, Ci, taps), f'det.slab.{B}.{S}.{Co}.{Ci}').to(DEV)
    w = synth.normal((Co, Ci, taps), f'det.w.{Co}.{Ci}.{taps}').to(DEV)
    isc = (1 + 0.3 * synth.normal((B, Ci), 'det.isc')).to(DEV) if mods[0] else None
    osc = (1 + 0.3 * synth.normal((B, Co), 'det.osc')).to(DEV) if mods[1] else None
    ws = 0.37
    want = (True, mods[0], mods[1])
    got = _lib.wgrad_reduce(slabs, w, ws, isc, osc, want_w=True, want_isc=want[1], want_osc=want[2])
    again = _lib.wgrad_reduce(slabs, w, ws, isc, osc, want_w=True, want_isc=want[1], want_osc=want[2])
    for a, b in zip(got, again):
        assert _bits_equal(a, b)
    sl = slabs.double().sum(1)                                        # [B, Co, Ci, T]
    i64 = isc.double() if isc is not None else torch.ones(B, Ci, device=DEV, dtype=torch.float64)
    o64 = osc.double() if osc is not None else torch.ones(B, Co, device=DEV, dtype=torch.float64)
    gw = ws * torch.einsum('bo,bi,boit->oit', o64, i64, sl)
    gi = ws * torch.einsum('oit,bo,boit->bi', w.double(), o64, sl)
    go = ws * torch.einsum('oit,bi,boit->bo', w.double(), i64, sl)
    assert rel_err(got[0], gw) < 2e-5
    if want[1]:
        assert rel_err(got[1], gi) < 2e-5
    if want[2]:
        assert rel_err(got[2], go) < 2e-5


@pytest.mark.parametrize('shape', [(4, 64, 64, 64), (2, 6, 32, 32), (3, 5, 7, 9), (16, 512), (2, 33, 6, 6)])
def test_bias_act_bwd_bias_gradient_written_and_bit_reproducible(shape):
    g = synth.normal(shape, 'det.g').to(DEV)
    ref = synth.normal(shape, 'det.ref').to(DEV)
    gi, gb = _lib.bias_act_bwd(g, ref, 0.2, math.sqrt(2))
    gi2, gb2 = _lib.bias_act_bwd(g, ref, 0.2, math.sqrt(2))
    assert _bits_equal(gi, gi2) and _bits_equal(gb, gb2)
    want = g.double() * torch.where(ref > 0, 1.0, 0.2).double() * math.sqrt(2)
    assert rel_err(gi, want) < 1e-6
    dims = [0] + list(range(2, len(shape)))
    assert rel_err(gb, want.sum(dim=dims)) < 1e-5


def test_conv_without_workspace_matches_the_split_form():
    """te_conv_f32 has no workspace argument: it no longer splits the channel loop into atomically combined parts; the result
    equals the workspace (split + fixed-order sum) form up to summation order"""
    import ctypes as C
    x = synth.normal((2, 512, 4, 4), 'det.cx').to(DEV)
    w = (synth.normal((512, 512, 3, 3), 'det.cw') / 68).to(DEV)
    wp = _lib.conv_pack(w, _lib.PACK_FWD)
    assert _lib.lib().te_conv_splitk_count(_lib.CONV_3X3, 2, 512, 512, 4, 4) > 1
    ref = _lib.conv(x, wp, _lib.CONV_3X3, 512, 4, 4)
    outs = []
    for _ in range(2):
        out = torch.empty_like(ref)
        rc = _lib.lib().te_conv_f32(out.data_ptr(), x.data_ptr(), wp.data_ptr(), None, None, None, 0, _lib.CONV_3X3, 2, 512, 512, 4, 4,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        outs.append(out)
    assert _bits_equal(outs[0], outs[1])
    assert rel_err(outs[0], ref) < 1e-5


def _grads_twice(run):
    a = run()
    b = run()
    bad = [i for i, (x, y) in enumerate(zip(a, b)) if not _bits_equal(x, y)]
    return a, bad


def test_generator256_backward_is_bit_reproducible():
    from test_oracle_golden import generator_state
    G, sd = generator_state(256, 7)
    G.load_state_dict(sd)
    G = G.to(DEV)
    z, p = synth.latents(4, 8080)
    w = synth.normal((4, 3, 256, 256), 'det.w256').to(DEV)
    params = [q for q in G.parameters()]

    def run():
        zd, pd = z.to(DEV).requires_grad_(True), p.to(DEV).requires_grad_(True)
        img = G(zd, pd)[0]
        return [img.detach()] + list(torch.autograd.grad((img * w).sum() / img.numel(), [zd, pd] + params, allow_unused=True))
    _, bad = _grads_twice(run)
    assert not bad, bad[:8]


def test_discriminator256_step_and_r1_backward_are_bit_reproducible():
    from transeditor_amd.model_spatial_query import Discriminator
    from transeditor_amd.op.modconv import second_order
    from transeditor_amd.train_step import d_logistic_loss, d_r1_loss
    Dn = Discriminator(256)
    synth.fill_state_dict(Dn.state_dict(), 5)
    Dn = Dn.to(DEV)
    img = synth.normal((8, 3, 256, 256), 'det.dimg').clamp(-1, 1).to(DEV)
    params = list(Dn.parameters())

    def step():
        fake_pred, real_pred = Dn(img, chunks=2).chunk(2)
        return list(torch.autograd.grad(d_logistic_loss(real_pred, fake_pred), params))

    def r1():
        x = img[:4].detach().requires_grad_(True)
        with second_order():
            pred = Dn(x)
        return list(torch.autograd.grad(10 / 2 * d_r1_loss(pred, x) * 16 + 0 * pred[0], params, allow_unused=True))
    for run in (step, r1):
        _, bad = _grads_twice(run)
        assert not bad, (run.__name__, bad[:8])
