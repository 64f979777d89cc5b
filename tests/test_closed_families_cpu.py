"""Autograd algebra of the closed (any-order differentiable) families, checked on the CPU tier.

The families of op/linear.py (`_LinFwd/_LinDx/_LinDw`, `_Bmm`) are autograd Functions whose backward is built from each
other; what can go wrong in them is graph plumbing (a saved tensor without history, a transposition in the wrong branch), not
arithmetic.  Here the two kernel bindings they call (`_lib.small_gemm`, `_lib.small_gemm_batched`) are replaced by strided torch
restatements of the documented contract (te_hip.h: C(i,j) = alpha * sum_k A(i,k) B(k,j) through element strides) - test
infrastructure, monkeypatched for the duration of a test - so the Functions run in fp64 on the CPU and every derivative up to
second order is compared with torch.matmul.  The kernels themselves are compared with the oracle in the `-m gpu` tier."""
import pytest
import torch

from transeditor_amd import _lib
from transeditor_amd.op import linear as lin
from plain_torch import attention_core as plain_attention


def _strided(t, shape, strides):
    return torch.as_strided(t, shape, strides, t.storage_offset())


@pytest.fixture
def emulated_gemms(monkeypatch):
    def small_gemm(I, J, K, a, sai, sak, b, sbk, sbj, bias=None, residual=None, alpha=1.0, beta=1.0, act=0, want_pre=False,
                   rowsum_scale=None):
        assert bias is None and residual is None and act == 0
        return alpha * _strided(a, (I, K), (sai, sak)) @ _strided(b, (K, J), (sbk, sbj)), None, None

    def small_gemm_batched(c, a, b, bias, nz, za, zc, I, J, K, sai, sak, sbk, sbj, sci, scj, zb=0, zbias=0, b_tab=None,
                           bias_tab=None, alpha=1.0, beta=1.0, act=0):
        assert bias is None and b_tab is None and act == 0
        _strided(c, (nz, I, J), (zc, sci, scj)).copy_(alpha * _strided(a, (nz, I, K), (za, sai, sak)) @ _strided(b, (nz, K, J), (zb, sbk, sbj)))
        return c
    monkeypatch.setattr(_lib, 'small_gemm', small_gemm)
    monkeypatch.setattr(_lib, 'small_gemm_batched', small_gemm_batched)


def _second_order(f, ins, gout):
    """value, first gradients (recorded), and the gradient of a scalar of those: walks every branch of a closed family"""
    y = f(*ins)
    g1 = torch.autograd.grad(y, ins, gout, create_graph=True)
    probe = sum((g * torch.roll(g.detach(), 1, -1)).sum() + g.square().sum() for g in g1)
    return (y,) + tuple(g1) + tuple(torch.autograd.grad(probe, ins))


@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('views', [False, True])
def test_bmm_family_algebra(emulated_gemms, ta, tb, views):
    """all four transposition forms; `views`: both operands arrive as strided views (what a [1,G,M,D] -> [G,M,D] reshape of a
    permuted tensor is at batch 1) - the saved tensors must be the inputs themselves, with their history"""
    torch.manual_seed(0)
    Z, I, J, K = 3, 5, 4, 6
    a0 = torch.randn((Z, K, I) if ta else (Z, I, K), dtype=torch.float64)
    b0 = torch.randn((Z, J, K) if tb else (Z, K, J), dtype=torch.float64)
    gc = torch.randn(Z, I, J, dtype=torch.float64)
    if views:       # leaves are the transposed storage; the family sees non-contiguous views of them
        la, lb = a0.transpose(1, 2).contiguous().requires_grad_(True), b0.transpose(1, 2).contiguous().requires_grad_(True)
        pre = lambda t: t.transpose(1, 2)
    else:
        la, lb = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        pre = lambda t: t
    plain = lambda a, b: 0.7 * torch.matmul(pre(a).transpose(1, 2) if ta else pre(a), pre(b).transpose(1, 2) if tb else pre(b))
    ours = lambda a, b: lin._Bmm.apply(pre(a), pre(b), ta, tb, 0.7)
    want = _second_order(plain, (la, lb), gc)
    got = _second_order(ours, (la, lb), gc)
    for name, x, y in zip(('c', 'ga', 'gb', 'gga', 'ggb'), got, want):
        assert torch.allclose(x, y, rtol=1e-10, atol=1e-12), name


@pytest.mark.parametrize('views', [False, True])
def test_linear_trio_algebra(emulated_gemms, views):
    """y = alpha x W^T through _LinFwd (and, by differentiation, _LinDx / _LinDw); `views`: x is a row slice of a wider
    tensor (the per-layer latent `latent[:, i]`)"""
    torch.manual_seed(1)
    R, K, N = 6, 5, 4
    w = torch.randn(N, K, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(R, N, dtype=torch.float64)
    if views:
        lat = torch.randn(R, 3, K, dtype=torch.float64, requires_grad=True)
        pre = lambda t: t[:, 1]
    else:
        lat = torch.randn(R, K, dtype=torch.float64, requires_grad=True)
        pre = lambda t: t
    want = _second_order(lambda x, w: 0.3 * pre(x) @ w.t(), (lat, w), gy)
    got = _second_order(lambda x, w: lin._LinFwd.apply(pre(x), w, 0.3), (lat, w), gy)
    for name, x, y in zip(('y', 'gx', 'gw', 'ggx', 'ggw'), got, want):
        assert torch.allclose(x, y, rtol=1e-10, atol=1e-12), name


@pytest.mark.parametrize('N', [1, 2])
def test_attention_recorded_backward_expression(emulated_gemms, monkeypatch, N):
    """the expression the attention core differentiates when its backward is recorded (op/attention.py::_torch_expr: two
    members of the batched family + softmax) against the einsum restatement of model_spatial_query.py:888-894, to second order;
    N = 1 is the case where the head reshape is a view"""
    from transeditor_amd.op import attention as att
    monkeypatch.setattr(att, 'bmm', lambda a, b, ta=False, tb=False, alpha=1.0: lin._Bmm.apply(a, b, bool(ta), bool(tb), float(alpha)))
    torch.manual_seed(2)
    q, k, v = (torch.randn(N, 16, 128, dtype=torch.float64, requires_grad=True) for _ in range(3))
    go, gs = torch.randn(N, 16, 128, dtype=torch.float64), torch.randn(N, 4, 16, 16, dtype=torch.float64)

    def second(f):
        o, sim = f(q, k, v, 0.3, 4)
        g1 = torch.autograd.grad([o, sim], (q, k, v), [go, gs], create_graph=True)
        return (o, sim) + tuple(g1) + tuple(torch.autograd.grad(sum(t.square().sum() for t in g1), (q, k, v)))
    for name, x, y in zip(('o', 'sim', 'gq', 'gk', 'gv', 'ggq', 'ggk', 'ggv'), second(att._torch_expr), second(plain_attention)):
        assert torch.allclose(x, y, rtol=1e-9, atol=1e-11), name


# ------------------------------------------------------------------------------------------------ convolution trio + composite
import math

import torch.nn.functional as F

from transeditor_amd.op import chanscale, modconv


def _conv_ref(x, w, kind):
    """the four kinds as stock framework convolutions (op/modconv.py::_KIND; 'up' = transposed stride 2 -> 2H+1, 'down' = stride 2)"""
    if kind == '3x3':
        return F.conv2d(x, w, padding=1)
    if kind == '1x1':
        return F.conv2d(x, w)
    if kind == 'up':
        return F.conv_transpose2d(x, w.transpose(0, 1), stride=2)
    return F.conv2d(x, w, stride=2)


@pytest.fixture
def emulated_conv_ops(monkeypatch):
    """helper-level restatements (op/modconv.py docstrings): forward, data gradient, plain weight gradient of a kind; the
    per-channel scale pair; the fused bias + leaky-ReLU kernels (te_hip.h: out = lrelu(x + b) * scale; grad mode: (x + b) *
    slope(ref) * scale; backward pass: g * slope(ref) * scale and its sum over everything but the channel)."""
    def in_shape(g, w, kind):
        B, H, W = g.shape[0], g.shape[2], g.shape[3]
        return {'3x3': (B, w.shape[1], H, W), '1x1': (B, w.shape[1], H, W), 'up': (B, w.shape[1], (H - 1) // 2, (W - 1) // 2),
                'down': (B, w.shape[1], 2 * H + 1, 2 * W + 1)}[kind]

    ch = lambda t, sc: t if sc is None else t * sc[:, :, None, None]

    def fwd_raw(x, w, kind, isc=None, osc=None, bias=None, act=0, wscale=1.0, with_bwd_pack=False):
        """te_conv_f32's contract: osc * conv(isc * x, wscale w) (style scale at staging, demodulation in the epilogue)"""
        assert bias is None and not act and not with_bwd_pack
        return ch(_conv_ref(ch(x, isc), w * wscale, kind), osc)

    def dgrad_raw(g, w, kind, isc=None, osc=None, wscale=1.0, wp=None):
        """isc scales the channels of g (shaped like the convolution's output), osc the channels of the result"""
        with torch.enable_grad():
            x0 = torch.zeros(in_shape(g, w, kind), dtype=g.dtype, requires_grad=True)
            gx, = torch.autograd.grad(_conv_ref(x0, w.detach() * wscale, kind), x0, ch(g.detach(), isc))
        return ch(gx, osc)

    def wgrad_raw(g, x, kind, group=False):
        """per-sample correlation slabs [B, 1, Co, Ci, taps] of the UNMODULATED tensors"""
        assert kind != 'down'
        ks = 1 if kind == '1x1' else 3
        out = []
        for b in range(g.shape[0]):
            with torch.enable_grad():
                w0 = torch.zeros(g.shape[1], x.shape[1], ks, ks, dtype=g.dtype, requires_grad=True)
                gw, = torch.autograd.grad(_conv_ref(x[b:b + 1].detach(), w0, kind), w0, g[b:b + 1].detach())
            out.append(gw.reshape(g.shape[1], x.shape[1], ks * ks))
        return torch.stack(out).unsqueeze(1)

    def wgrad_reduce(slabs, w, wscale=1.0, isc=None, osc=None, want_w=True, want_isc=False, want_osc=False):
        """te_wgrad_reduce_f32's dW: wscale * sum_{b,s} osc[b,co] isc[b,ci] slab"""
        assert want_w and not want_isc and not want_osc
        sl = slabs.sum(1)
        if isc is not None:
            sl = sl * isc[:, None, :, None]
        if osc is not None:
            sl = sl * osc[:, :, None, None]
        return wscale * sl.sum(0), None, None

    def wgrad_plain(gy, x, kind, ksize, wscale):
        with torch.enable_grad():
            w0 = torch.zeros(gy.shape[1], x.shape[1], ksize, ksize, dtype=gy.dtype, requires_grad=True)
            gw, = torch.autograd.grad(_conv_ref(x.detach(), w0 * wscale, kind), w0, gy.detach())
        return gw

    slope = lambda ref, alpha: torch.where(ref > 0, torch.ones_like(ref), torch.full_like(ref, alpha))      # (in ref's precision)
    bshape = lambda x: (1, -1) + (1,) * (x.dim() - 2)

    def bias_act(x, b, ref, act, grad, alpha, scale):
        assert act == 3
        v = x if b is None else x + b.reshape(bshape(x))
        return (F.leaky_relu(v, alpha) if grad == 0 else v * slope(ref, alpha)) * scale

    def bias_act_bwd(g, ref, alpha, scale, want_bias=True):
        gi = g * slope(ref, alpha) * scale
        return gi, (gi.sum(dim=[0] + list(range(2, gi.dim()))) if want_bias else None)
    monkeypatch.setattr(modconv, '_fwd_raw', fwd_raw)
    monkeypatch.setattr(modconv, '_dgrad_raw', dgrad_raw)
    monkeypatch.setattr(modconv, '_wgrad_plain', wgrad_plain)
    monkeypatch.setattr(modconv, '_wgrad_raw', wgrad_raw)
    monkeypatch.setattr(_lib, 'wgrad_reduce', wgrad_reduce)
    monkeypatch.setattr(_lib, 'rgb_supported', lambda M, K, HW: False)
    monkeypatch.setattr(_lib, 'chan_scale', lambda x, s: x * s.reshape(*s.shape, *([1] * (x.dim() - 2))))
    monkeypatch.setattr(_lib, 'chan_dot', lambda a, b: (a * b).flatten(2).sum(2))
    monkeypatch.setattr(_lib, 'bias_act', bias_act)
    monkeypatch.setattr(_lib, 'bias_act_bwd', bias_act_bwd)
    monkeypatch.setattr(modconv, 'chan_scale', lambda x, s: chanscale._ChanScale.apply(x, s))       # (the wrapper asks for a GPU tensor)


@pytest.mark.parametrize('kind', ['3x3', '1x1', 'up', 'down'])
def test_conv_trio_algebra(emulated_conv_ops, kind):
    """y = conv(x, wscale w) through _ConvFwd and, by differentiation, _ConvDgrad / _ConvWgrad (each one's backward is the
    other two, with the same constant): value, both gradients, second differentiation, for the four kinds"""
    torch.manual_seed(3)
    ks = 1 if kind == '1x1' else 3
    x = torch.randn(2, 3, 7 if kind == 'down' else 4, 9 if kind == 'down' else 5, dtype=torch.float64, requires_grad=True)
    w = torch.randn(4, 3, ks, ks, dtype=torch.float64, requires_grad=True)
    ref = lambda x, w: _conv_ref(x, w * 0.37, kind)
    gy = torch.randn_like(ref(x, w))
    want = _second_order(ref, (x, w), gy)
    got = _second_order(lambda x, w: modconv.conv_core(x, w, kind, 0.37), (x, w), gy)
    for name, a, b in zip(('y', 'gx', 'gw', 'ggx', 'ggw'), got, want):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-12), name


@pytest.mark.parametrize('kind,demod', [('3x3', True), ('up', True), ('1x1', False), ('1x1', True)])
def test_closed_modulated_conv_family_algebra(emulated_conv_ops, monkeypatch, kind, demod):
    """the five-linear family of the modulated convolution (op/modconv.py::_MCFwd / _MCDgrad / _MCWgrad: style scale and
    demodulation INSIDE the convolution calls, the scale gradients as channel dots divided by the scale) against the broadcast
    expression: value, all four first gradients (recorded), and the gradient of a scalar of THOSE w.r.t. every input - which walks
    every backward branch of the three Functions (ModulatedConv2d.forward, model_spatial_query.py:296-337; ToRGB without
    demodulation :416-425)"""
    monkeypatch.setattr(modconv, '_closed_ok', lambda x: True)
    torch.manual_seed(6)
    B, Ci, Co = 2, 3, 4
    ks = 1 if kind == '1x1' else 3
    x = torch.randn(B, Ci, 4, 5, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, ks, ks, dtype=torch.float64, requires_grad=True)
    s = (1 + 0.3 * torch.randn(B, Ci, dtype=torch.float64)).requires_grad_(True)
    d = (1 + 0.3 * torch.randn(B, Co, dtype=torch.float64)).requires_grad_(True) if demod else None
    ins = (x, w, s) + ((d,) if demod else ())

    def plain(x, w, s, d=None):
        y = _conv_ref(x * s[:, :, None, None], w * 0.21, kind)
        return y if d is None else y * d[:, :, None, None]

    def ours(x, w, s, d=None):
        return modconv.modconv_closed(x, w, s, d, kind, 0.21)
    gy = torch.randn_like(plain(*ins))
    want = _second_order(plain, ins, gy)
    got = _second_order(ours, ins, gy)
    names = ['y'] + [f'g{i}' for i in range(len(ins))] + [f'G{i}' for i in range(len(ins))]
    for name, a, b in zip(names, got, want):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-11), name
    # an exactly-zero style scale does not poison the scale gradient (the division sees the smallest normal number instead)
    s0 = s.detach().clone()
    s0[0, 1] = 0.0
    s0.requires_grad_(True)
    y = ours(x, w, s0, d) if demod else ours(x, w, s0)
    gs, = torch.autograd.grad(y, s0, gy)
    assert torch.isfinite(gs).all()


@pytest.mark.parametrize('kind,act', [('3x3', True), ('up', False), ('down', True), ('1x1', 1.0)])
@pytest.mark.parametrize('closed', [False, True])
def test_modulated_conv_composite_algebra(emulated_conv_ops, monkeypatch, kind, act, closed):
    """the any-order composite of the modulated convolution (style scale -> trio -> demodulation scale -> bias + leaky ReLU,
    op/modconv.py::_composite) against the broadcast expression, through a path-length-style probe: gradients w.r.t. the input
    and the style with create_graph, then the gradient of their squares w.r.t. EVERY input (model_spatial_query.py:296-337 +
    train_spatial_query.py:92-105)"""
    monkeypatch.setattr(modconv, '_closed_ok', lambda x: closed)      # both routes of _composite: closed family / chan_scale + trio
    torch.manual_seed(4)
    B, Ci, Co = 2, 3, 4
    ks = 1 if kind == '1x1' else 3
    x = torch.randn(B, Ci, 7 if kind == 'down' else 4, 7 if kind == 'down' else 4, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, ks, ks, dtype=torch.float64, requires_grad=True)
    s = (1 + 0.3 * torch.randn(B, Ci, dtype=torch.float64)).requires_grad_(True)
    d = (1 + 0.3 * torch.randn(B, Co, dtype=torch.float64)).requires_grad_(True)
    bias = torch.randn(Co, dtype=torch.float64, requires_grad=True)
    ins = (x, w, s, d, bias)

    def plain(x, w, s, d, bias):
        y = _conv_ref(x * s[:, :, None, None], w * 0.21, kind) * d[:, :, None, None] + bias[None, :, None, None]
        return F.leaky_relu(y, 0.2) * (math.sqrt(2) if act is True else float(act)) if act else y

    def ours(x, w, s, d, bias):
        return modconv._composite(x, w, s, d, bias, act, kind, 0.21)
    gy = torch.randn_like(plain(*ins))

    def probe(f):
        y = f(*ins)
        gx, gs = torch.autograd.grad(y, (x, s), gy, create_graph=True)
        return (y, gx, gs) + tuple(torch.autograd.grad(gx.square().sum() + gs.square().sum(), ins, allow_unused=True))
    for name, a, b in zip(('y', 'gx', 'gs', 'Gx', 'Gw', 'Gs', 'Gd', 'Gbias'), probe(ours), probe(plain)):
        if b is None:            # (without an activation the probe does not depend on the bias)
            assert a is None or float(a.abs().max()) == 0, name
        else:
            assert torch.allclose(a, b, rtol=1e-9, atol=1e-11), name


# ------------------------------------------------------------------------------------------------ batched launches (pointer tables)
import ctypes


@pytest.fixture
def emulated_batched_gemm(monkeypatch, emulated_gemms):
    """te_small_gemm_batched_f32 with its per-z pointer tables (te_hip.h: operand z = b + b_tab[z] elements, bias + bias_tab[z]):
    the emulation resolves the tables through the CPU tensors' addresses, so the launches below really read separately
    allocated parameters the way the kernel does (fp32: the tables count 4-byte elements)."""
    def at(base, off, n):
        return torch.frombuffer((ctypes.c_float * n).from_address(base.data_ptr() + 4 * off), dtype=torch.float32)

    def small_gemm_batched(c, a, b, bias, nz, za, zc, I, J, K, sai, sak, sbk, sbj, sci, scj, zb=0, zbias=0, b_tab=None,
                           bias_tab=None, alpha=1.0, beta=1.0, act=0):
        assert a.dtype == torch.float32
        for z in range(nz):
            A = torch.as_strided(a, (I, K), (sai, sak), a.storage_offset() + z * za)
            if b_tab is not None:
                span = (K - 1) * sbk + (J - 1) * sbj + 1
                B = torch.as_strided(at(b, b_tab[z], span), (K, J), (sbk, sbj))
            else:
                B = torch.as_strided(b, (K, J), (sbk, sbj), b.storage_offset() + z * zb)
            y = alpha * (A @ B)
            if bias is not None:
                y = y + beta * (at(bias, bias_tab[z], J) if bias_tab is not None else
                                torch.as_strided(bias, (J,), (1,), bias.storage_offset() + z * zbias))
            if act == 3:
                y = F.leaky_relu(y, 0.2) * math.sqrt(2)
            else:
                assert act == 0
            torch.as_strided(c, (I, J), (sci, scj), c.storage_offset() + z * zc).copy_(y)
        return c
    slope = lambda ref, alpha: torch.where(ref > 0, torch.ones_like(ref), torch.full_like(ref, alpha))

    def bias_act_bwd(g, ref, alpha, scale, want_bias=True):
        gi = g * slope(ref, alpha) * scale
        return gi, (gi.sum(dim=[0] + list(range(2, gi.dim()))) if want_bias else None)
    monkeypatch.setattr(_lib, 'small_gemm_batched', small_gemm_batched)
    monkeypatch.setattr(_lib, 'bias_act_bwd', bias_act_bwd)


def _layers(n, K, J, seed):
    g = torch.Generator().manual_seed(seed)
    ws = [torch.randn(J, K, generator=g).requires_grad_(True) for _ in range(n)]          # separately allocated, like nn.Parameters
    bs = [torch.randn(J, generator=g).requires_grad_(True) for _ in range(n)]
    return ws, bs


def _compare(got, want, tol=2e-5):
    for i, (a, b) in enumerate(zip(got, want)):
        assert float((a - b).norm()) <= tol * float(b.norm()) + 1e-7, i


def test_shared_input_linears_plumbing(emulated_batched_gemm):
    """_SharedInputLinears (k / v and the query projections of the attention blocks as one launch): strides, pointer tables, the
    slab sum of dx, every dW / db, and the recorded backward, against per-layer F.linear"""
    ws, bs = _layers(3, 12, 5, 0)
    x = torch.randn(2, 4, 12, requires_grad=True)
    gys = [torch.randn(2, 4, 5) for _ in range(3)]
    ours = lambda: lin._SharedInputLinears.apply(x, 0.3, 0.7, 3, *ws, *bs)
    plain = lambda: [F.linear(x, w * 0.3, b * 0.7) for w, b in zip(ws, bs)]
    for a, b in zip(ours(), plain()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    _compare(torch.autograd.grad(ours(), [x] + ws + bs, gys), torch.autograd.grad(plain(), [x] + ws + bs, gys))
    g1, = torch.autograd.grad(ours(), x, gys, create_graph=True)
    g2, = torch.autograd.grad(plain(), x, gys, create_graph=True)
    _compare(torch.autograd.grad(g1.square().sum(), ws), torch.autograd.grad(g2.square().sum(), ws))


def test_batched_modulation_plumbing(emulated_batched_gemm):
    """_BatchedModulation (all style modulations of a pass, grouped by width): group-major gather, per-group launches with the
    latent's strides, scatter of d latent, dW / db, recorded backward"""
    from transeditor_amd.op import modulation
    K, widths = 8, [6, 6, 3, 6, 3]
    order = [i for _, ids in modulation._groups(widths) for i in ids]
    g = torch.Generator().manual_seed(1)
    ws = [torch.randn(wd, K, generator=g).requires_grad_(True) for wd in widths]
    bs = [torch.randn(wd, generator=g).requires_grad_(True) for wd in widths]
    lat = torch.randn(3, 4, K, requires_grad=True)
    index = [0, 1, 1, 3, 2]
    gys = [torch.randn(3, wd) for wd in widths]

    def ours():
        lat_g = lat.index_select(1, torch.tensor([index[i] for i in order]))
        outs = modulation._BatchedModulation.apply(lat_g, 0.5, 1.0, len(widths), *[ws[i] for i in order], *[bs[i] for i in order])
        res = [None] * len(widths)
        for pos, i in enumerate(order):
            res[i] = outs[pos]
        return res
    plain = lambda: [F.linear(lat[:, index[i]], ws[i] * 0.5, bs[i]) for i in range(len(widths))]
    for a, b in zip(ours(), plain()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    _compare(torch.autograd.grad(ours(), [lat] + ws + bs, gys), torch.autograd.grad(plain(), [lat] + ws + bs, gys))
    g1, = torch.autograd.grad(ours(), lat, gys, create_graph=True)
    g2, = torch.autograd.grad(plain(), lat, gys, create_graph=True)
    _compare(torch.autograd.grad(g1.square().sum(), ws), torch.autograd.grad(g2.square().sum(), ws))


def test_token_mlp_plumbing(emulated_batched_gemm):
    """_TokenMLP (model_spatial_query.py:626-646: token t through its own EqualLinear + fused lrelu, one launch): the in-place
    read of the [B, D, tokens] code, the [B, T, D] write, dx back into the code's layout (tokens >= T get zeros), dW / db"""
    from transeditor_amd.op import token_mlp as tm
    B, D, Cn, T = 3, 8, 6, 4
    ws, bs = _layers(T, D, D, 2)
    x = torch.randn(B, D, Cn, requires_grad=True)
    gy = torch.randn(B, T, D)
    ours = lambda: tm._TokenMLP.apply(x, 0.2, 0.5, T, *ws, *bs)
    plain = lambda: torch.stack([F.leaky_relu(F.linear(x[:, :, t], ws[t] * 0.2, bs[t] * 0.5), 0.2) * math.sqrt(2) for t in range(T)], 1)
    assert torch.allclose(ours(), plain(), rtol=1e-5, atol=1e-6)
    _compare(torch.autograd.grad(ours(), [x] + ws + bs, gy), torch.autograd.grad(plain(), [x] + ws + bs, gy))


# ------------------------------------------------------------------------------------------------ ResBlock composite (R1 route)
def _upfirdn2d_general(x, k, up, down, pad):
    """te_upfirdn2d_f32's contract (te_hip.h / upfirdn2d_kernel.cu:85-129) with per-axis factors and per-side pads: zero-insert,
    pad (negative = crop), TRUE convolution with k, keep every down-th sample.  up / down = (x, y), pad = (px0, px1, py0, py1)."""
    B, C, H, W = x.shape
    px0, px1, py0, py1 = pad
    z = x.new_zeros(B * C, 1, H * up[1], W * up[0])
    z[:, :, ::up[1], ::up[0]] = x.reshape(B * C, 1, H, W)
    z = F.pad(z, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    z = z[:, :, max(-py0, 0): z.shape[2] - max(-py1, 0), max(-px0, 0): z.shape[3] - max(-px1, 0)]
    y = F.conv2d(z, torch.flip(k, [0, 1]).to(x.dtype).reshape(1, 1, *k.shape))[:, :, ::down[1], ::down[0]]
    return y.reshape(B, C, y.shape[2], y.shape[3])


@pytest.mark.parametrize('stem', [False, True])
def test_resblock_composite_algebra(emulated_conv_ops, monkeypatch, stem):
    """the any-order form of the discriminator's ResBlock (op/resblock.py::resblock_composite: what R1 differentiates twice, with
    the from-RGB stem in front for the first block) against the reference's layer sequence (model_spatial_query.py:731-798:
    conv3x3 + lrelu, Blur(2,2) + stride-2 conv + lrelu, skip = Blur(1,1) + stride-2 1x1 conv, (out + skip) / sqrt(2)), through an
    R1-style probe: d pred / d input with create_graph, then the gradient of its square w.r.t. every parameter"""
    from oracle import te_oracle as O
    from transeditor_amd.op import resblock as rb
    monkeypatch.setattr(_lib, 'upfirdn2d_raw',
                        lambda x, k, up, down, pad, bias=None, act=0, alpha=0.2, scale=1.0: _upfirdn2d_general(x, k, up, down, pad))
    torch.manual_seed(5)
    dt = torch.float64
    Cin, C1, C2 = (3 if stem else 4), 4, 6
    k = O.fir_kernel((1, 3, 3, 1)).to(dt)
    x = torch.randn(2, Cin, 8, 8, dtype=dt, requires_grad=True)
    P = [torch.randn(*s, dtype=dt, requires_grad=True) for s in [(C1, C1, 3, 3), (C1,), (C2, C1, 3, 3), (C2,), (C2, C1, 1, 1)]]
    S = [torch.randn(*s, dtype=dt, requires_grad=True) for s in [(C1, 3, 1, 1), (C1,)]] if stem else []
    s1, s2, ss, s0, gain = 0.3, 0.25, 0.5, 0.6, 1 / math.sqrt(2)
    lrelu = lambda t: F.leaky_relu(t, 0.2) * math.sqrt(2)

    def plain(x, w1, b1, w2, b2, ws, *st):
        if st:
            x = lrelu(F.conv2d(x, st[0] * s0) + st[1][None, :, None, None])
        out = lrelu(F.conv2d(x, w1 * s1, padding=1) + b1[None, :, None, None])
        out = lrelu(F.conv2d(O.upfirdn2d(out, k, pad=(2, 2)), w2 * s2, stride=2) + b2[None, :, None, None])
        skip = F.conv2d(O.upfirdn2d(x, k, pad=(1, 1)), ws * ss, stride=2)
        return (out + skip) / math.sqrt(2)

    def ours(x, w1, b1, w2, b2, ws, *st):
        return rb.resblock_composite(x, w1, b1, w2, b2, ws, k, k, s1, s2, ss, (2, 2), (1, 1), gain, *(st + ((s0,) if st else ())))
    gy = torch.randn_like(plain(x, *P, *S))

    def probe(f):
        y = f(x, *P, *S)
        gx, = torch.autograd.grad(y, x, gy, create_graph=True)
        return (y, gx) + tuple(torch.autograd.grad(gx.square().sum(), P + S))
    for i, (a, b) in enumerate(zip(probe(ours), probe(plain))):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-11), i
