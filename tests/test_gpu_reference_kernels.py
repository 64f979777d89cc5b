"""The REFERENCE's own native kernels, compiled for gfx950 from utils/op/fused_bias_act_kernel.cu and upfirdn2d_kernel.cu where
they lie (oracle/build_ref_ops.py -> oracle/_ref/*.so, built in the build container, shipped prebuilt), run on the MI355X
beside ours:

* K1 `fused_bias_act` in the three modes the reference's autograd wrapper uses (utils/op/fused_act.py:27-31, 43-45, 54):
  forward with bias, gradient with the saved output as reference, gradient-of-gradient with a bias term - fp32 BIT FOR BIT,
  half and double through te_bias_act_f16 / _f64;
* K2 `upfirdn2d` on the configurations the model uses (blur after the transposed convolution, the discriminator's blurs,
  x2 up / down resampling of the skip paths) and their adjoints, plus odd shapes, crops (negative pads) and non-square taps;
* the CPU oracle's restatements of both kernels against the real kernels - which pins the one piece of arithmetic in the golden
  fixtures that was "the oracle's restatement" (fused_bias_act_kernel.cu:26-47 has no CPU twin in the reference).

Skipped where oracle/_ref has not been built (no /root/reference at build time)."""
import math

import pytest
import torch

from conftest import rel_err
from oracle import build_ref_ops
from oracle import te_oracle as O
from transeditor_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'
SQRT2 = math.sqrt(2)


def _load(name):
    try:
        m = build_ref_ops.load_module(name)
    except Exception as e:      # built against another torch / ROCm than the one on this box: the checker is unavailable, not the product
        pytest.skip(f'oracle/_ref/{name}.so does not load here: {type(e).__name__}: {e}')
    if m is None:
        pytest.skip(f'oracle/_ref/{name}.so not built (reference tree absent at build time)')
    return m


@pytest.fixture(scope='module')
def ref_fused():
    return _load('te_ref_fused')


@pytest.fixture(scope='module')
def ref_fir():
    return _load('te_ref_upfirdn2d')


def _bits(a, b):
    return a.dtype == b.dtype and torch.equal(a.view(torch.int32 if a.dtype == torch.float32 else (torch.int16 if a.dtype == torch.float16 else torch.int64)),
                                              b.view(torch.int32 if b.dtype == torch.float32 else (torch.int16 if b.dtype == torch.float16 else torch.int64)))


@pytest.mark.parametrize('shape', [(4, 64, 32, 32), (2, 7, 5, 3), (16, 512), (3, 33), (2, 128, 64, 64)])
def test_fused_bias_act_bit_for_bit_vs_reference_kernel(ref_fused, shape):
    x = synth.normal(shape, 'refk.x').to(DEV)
    b = (0.5 * synth.normal((shape[1],), 'refk.b')).to(DEV)
    empty = x.new_empty(0)
    # forward (fused_act.py:54)
    want = ref_fused.fused_bias_act(x, b, empty, 3, 0, 0.2, SQRT2)
    got = _lib.bias_act(x, b, None, 3, 0, 0.2, SQRT2)
    assert _bits(got, want)
    # the oracle's restatement, on the CPU, against the real kernel
    orc = O.fused_leaky_relu(x.cpu(), b.cpu(), 0.2, SQRT2)
    assert _bits(orc, want.cpu()), float((orc - want.cpu()).abs().max())
    # gradient pass (fused_act.py:27-31): grad_output, no bias, the saved OUTPUT as the slope reference
    g = synth.normal(shape, 'refk.g').to(DEV)
    want_g = ref_fused.fused_bias_act(g, empty, want, 3, 1, 0.2, SQRT2)
    assert _bits(_lib.bias_act(g, None, want, 3, 1, 0.2, SQRT2), want_g)
    gi, gb = _lib.bias_act_bwd(g, want, 0.2, SQRT2)                   # our single pass: the same values + the bias gradient
    assert _bits(gi, want_g)
    dims = [0] + list(range(2, len(shape)))
    assert rel_err(gb, want_g.double().sum(dim=dims)) < 1e-5           # (fused_act.py:33-38: grad_input.sum(dim))
    # gradient of the gradient (fused_act.py:43-45): bias term present, slope from the saved output
    ggi, ggb = synth.normal(shape, 'refk.ggi').to(DEV), synth.normal((shape[1],), 'refk.ggb').to(DEV)
    want_gg = ref_fused.fused_bias_act(ggi, ggb, want, 3, 1, 0.2, SQRT2)
    assert _bits(_lib.bias_act(ggi, ggb, want, 3, 1, 0.2, SQRT2), want_gg)


@pytest.mark.parametrize('dtype', [torch.float16, torch.float64])
def test_fused_bias_act_other_dtypes_vs_reference_kernel(ref_fused, dtype):
    shape = (3, 16, 9, 11)
    x = synth.normal(shape, 'refk.xd').to(DEV).to(dtype)
    b = synth.normal((16,), 'refk.bd').to(DEV).to(dtype)
    empty = x.new_empty(0)
    want = ref_fused.fused_bias_act(x, b, empty, 3, 0, 0.2, SQRT2)
    got = _lib.bias_act(x, b, None, 3, 0, 0.2, SQRT2)
    assert got.dtype == dtype
    if dtype == torch.float64:
        assert _bits(got, want)
    else:       # half: the reference computes in half, we accumulate in fp32 and round once (te_hip.h): within one half ulp
        assert rel_err(got, want) < 2e-3
    g = synth.normal(shape, 'refk.gd').to(DEV).to(dtype)
    want_g = ref_fused.fused_bias_act(g, empty, want, 3, 1, 0.2, SQRT2)
    got_g = _lib.bias_act(g, None, want, 3, 1, 0.2, SQRT2)
    assert _bits(got_g, want_g) if dtype == torch.float64 else rel_err(got_g, want_g) < 2e-3


FIR_CASES = [   # (B, C, H, W), taps, up, down, (px0, px1, py0, py1)
    ((2, 8, 33, 33), (1, 3, 3, 1), 1, 1, (1, 1, 1, 1)),       # blur after the transposed convolution (2H+1 -> 2H), :264-266
    ((2, 8, 32, 32), (1, 3, 3, 1), 1, 1, (2, 2, 2, 2)),       # its adjoint / the discriminator's blur before the strided conv
    ((1, 4, 257, 257), (1, 3, 3, 1), 1, 1, (1, 1, 1, 1)),     # the FFHQ-256 top shape
    ((1, 4, 256, 256), (1, 3, 3, 1), 1, 1, (2, 2, 2, 2)),
    ((2, 3, 16, 16), (1, 3, 3, 1), 2, 1, (2, 1, 2, 1)),       # ToRGB skip: Upsample, :103-108
    ((2, 3, 32, 32), (1, 3, 3, 1), 1, 2, (1, 1, 1, 1)),       # its adjoint: Downsample, :124-129
    ((2, 8, 64, 64), (1, 3, 3, 1), 1, 2, (2, 2, 2, 2)),       # blur + every second pixel (discriminator skip branch)
    ((1, 5, 13, 9), (1, 2, 1), 1, 1, (0, 3, 2, 0)),           # odd sizes, 3 taps, asymmetric pads
    ((2, 2, 11, 17), (1, 3, 3, 1), 1, 1, (-1, 2, 1, -2)),     # negative pads = crop
    ((2, 3, 9, 7), (1, 1), 2, 1, (1, 0, 1, 0)),               # x2 up with 2 taps (the reference's mode 4)
    ((2, 3, 18, 14), (1, 1), 1, 2, (0, 0, 0, 0)),             # x2 down with 2 taps (mode 6)
    ((1, 2, 40, 70), (1, 2, 1), 1, 1, (1, 1, 1, 1)),          # 3 taps (mode 2), more than one 16 x 64 tile
]
# (upfirdn2d_kernel.cu:176-215 launches a kernel for exactly six (up, down, taps) classes - the ones above; for any other
# configuration the reference op returns its uninitialised output buffer.  Our generic form, fir_direct_kernel, is checked
# against the oracle in tests/test_gpu_fir_fuzz.py.)


@pytest.mark.parametrize('shape,taps,up,down,pad', FIR_CASES)
def test_upfirdn2d_vs_reference_kernel(ref_fir, shape, taps, up, down, pad):
    B, C, H, W = shape
    x = synth.normal(shape, f'refk.fir.{H}.{W}').to(DEV)
    k = O.fir_kernel(taps, float(up * up)).to(DEV)
    # the reference op works on [major, in_h, in_w, minor] with minor = 1 (utils/op/upfirdn2d.py:97, 116-118)
    want = ref_fir.upfirdn2d(x.reshape(-1, H, W, 1), k, up, up, down, down, pad[0], pad[1], pad[2], pad[3])
    want = want.reshape(B, C, want.shape[1], want.shape[2])
    got = _lib.upfirdn2d_raw(x, k, (up, up), (down, down), pad)
    assert tuple(got.shape) == tuple(want.shape)                      # integer index path: exact
    e = rel_err(got, want)
    assert e < 1e-6, e                                                # same taps, same sums; only the order of the 16 additions may differ
    if pad[0] == pad[2] and pad[1] == pad[3]:                         # the oracle's restatement takes one (pad0, pad1) pair for both axes
        orc = O.upfirdn2d(x.cpu(), k.cpu(), up, down, (pad[0], pad[1]))
        assert tuple(orc.shape) == tuple(want.shape) and rel_err(orc, want) < 1e-6
