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
