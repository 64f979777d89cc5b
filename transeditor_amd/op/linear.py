"""EqualLinear as ONE fused launch (te_small_gemm_f32): y = act(alpha * x W^T + beta * bias) + residual.

Reference: EqualLinear.forward, model_spatial_query.py:213-221 (F.linear(input, weight * scale, bias * lr_mul), optional
fused leaky-ReLU) and the residual / GELU around it in AttentionBlock.forward (:920-936).  alpha = scale, beta = lr_mul,
so the parameters are consumed as stored (no `weight * scale` / `bias * lr_mul` launches).  Backward: dx = alpha g W and
dW = alpha g^T x on the same kernel, db = beta * sum(g); a recorded backward (create_graph) goes through `_closed_expr`:
the GEMM as a trio of te_small_gemm launches closed under differentiation (no library GEMM), bias / activation through the
twice-differentiable K1 op (leaky-ReLU) or torch's elementwise GELU.
"""
import math

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib
from .modconv import _STATE as _modconv_state

_ACT = {None: 0, 'gelu': 1, 'lrelu': 3}
MAX_K = 1024          # widest reduction of the single-pass kernel; beyond it the split-K form (discriminator's 8192-wide linear)


def _ksplit(K):
    """number of K chunks for the split-K form: chunks of <= 1024 that are multiples of 8, or 0 if K does not split"""
    for S in range(-(-K // MAX_K), 65):
        if K % S == 0 and (K // S) % 8 == 0:
            return S
    return 0


def _gemm(I, J, K, a, sai, sak, b, sbk, sbj, alpha):
    """alpha * A[I,K] B[K,J] through the strided small-GEMM kernel (split-K form for wide reductions)"""
    if K > MAX_K:
        S = _ksplit(K)
        if S:
            return _lib.small_gemm_splitk(I, J, K, S, a, sai, sak, b, sbk, sbj, alpha=alpha)[0]
        # a reduction that does not split evenly: whole 1024-wide chunks through the split-K form, the tail (< 1024) through the
        # single-pass kernel with the first part as its residual
        K1 = (K // MAX_K) * MAX_K
        c = _lib.small_gemm_splitk(I, J, K1, K1 // MAX_K, a, sai, sak, b, sbk, sbj, alpha=alpha)[0]
        at = a.as_strided((1,), (1,), a.storage_offset() + K1 * sak)          # (operands are addressed from their first element)
        bt = b.as_strided((1,), (1,), b.storage_offset() + K1 * sbk)
        return _lib.small_gemm(I, J, K - K1, at, sai, sak, bt, sbk, sbj, residual=c, alpha=alpha)[0]
    return _lib.small_gemm(I, J, K, a, sai, sak, b, sbk, sbj, alpha=alpha)[0]


def _rows2(x):
    """a 2-D operand with unit inner stride is used through its row stride; anything else is densified"""
    return x if (x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= x.shape[1]) else x.contiguous()


# The bilinear map y = alpha * x W^T as a trio closed under differentiation (each one's backward is the other two), like the
# convolution trio of op/modconv.py: whatever is built from it differentiates to any order on te_small_gemm_f32 alone.  The
# recorded backward of every EqualLinear (path-length regulariser: the 20 style modulations; `--spatial_regu`: the mapping
# and attention stacks) goes through it instead of F.linear / torch.mm (library GEMMs).
class _LinFwd(Function):
    @staticmethod
    def forward(ctx, x, w, alpha):                     # x [R,K], w [N,K] -> [R,N]
        ctx.save_for_backward(x, w)         # the INPUTS (with their history: the backward may itself be recorded), not dense copies
        ctx.alpha = alpha
        x, w = _rows2(x), w.contiguous()           # (a row slice of a wider tensor - latent[:, i] - is read in place)
        return _gemm(x.shape[0], w.shape[0], w.shape[1], x, x.stride(0), 1, w, 1, w.shape[1], alpha)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = _LinDx.apply(gy, w, ctx.alpha) if ctx.needs_input_grad[0] else None
        gw = _LinDw.apply(gy, x, ctx.alpha) if ctx.needs_input_grad[1] else None
        return gx, gw, None


class _LinDx(Function):
    @staticmethod
    def forward(ctx, g, w, alpha):                     # g [R,N], w [N,K] -> alpha g W  [R,K]
        ctx.save_for_backward(g, w)
        ctx.alpha = alpha
        g, w = g.contiguous(), w.contiguous()
        return _gemm(g.shape[0], w.shape[1], w.shape[0], g, g.stride(0), 1, w, w.shape[1], 1, alpha)

    @staticmethod
    def backward(ctx, ggx):
        g, w = ctx.saved_tensors
        g_g = _LinFwd.apply(ggx, w, ctx.alpha) if ctx.needs_input_grad[0] else None
        g_w = _LinDw.apply(g, ggx, ctx.alpha) if ctx.needs_input_grad[1] else None
        return g_g, g_w, None


class _LinDw(Function):
    @staticmethod
    def forward(ctx, g, x, alpha):                     # g [R,N], x [R,K] -> alpha g^T x  [N,K]
        ctx.save_for_backward(g, x)
        ctx.alpha = alpha
        g, x = g.contiguous(), _rows2(x)
        return _gemm(g.shape[1], x.shape[1], g.shape[0], g, 1, g.shape[1], x, x.stride(0), 1, alpha)

    @staticmethod
    def backward(ctx, ggw):
        g, x = ctx.saved_tensors
        g_g = _LinFwd.apply(x, ggw, ctx.alpha) if ctx.needs_input_grad[0] else None
        g_x = _LinDx.apply(g, ggw, ctx.alpha) if ctx.needs_input_grad[1] else None
        return g_g, g_x, None


class _Bmm(Function):
    """C_z = alpha * op_a(A_z) op_b(B_z) over a leading batch dimension, on te_small_gemm_batched_f32; a, b are [Z, ., .] and
    `ta` / `tb` say which of them is read transposed.  The family is closed under differentiation (every gradient below is
    another _Bmm), so the recorded backward of the attention core (q k^T, sim v), of the token-wise mapping network and of
    anything else built from batched products stays on our kernel to any order - no library GEMM (torch.bmm / matmul)."""

    @staticmethod
    def forward(ctx, a, b, ta, tb, alpha):
        # the INPUTS are saved, not their dense copies: the backward may itself be recorded, and a copy made in here has no
        # history (a [1,G,M,D] -> [G,M,D] reshape of a permuted tensor is a strided VIEW: batch 1 arrives non-contiguous)
        ctx.save_for_backward(a, b)
        a, b = a.contiguous(), b.contiguous()
        Z, p, q = a.shape
        _, r, t = b.shape
        I, K = (q, p) if ta else (p, q)
        K2, J = (t, r) if tb else (r, t)
        if K != K2 or b.shape[0] != Z:
            raise ValueError(f'te_hip: batched product of {tuple(a.shape)} (transposed={ta}) and {tuple(b.shape)} (transposed={tb})')
        c = torch.empty(Z, I, J, device=a.device, dtype=a.dtype)
        sai, sak = (1, q) if ta else (q, 1)
        sbk, sbj = (1, t) if tb else (t, 1)
        _lib.small_gemm_batched(c, a, b, None, Z, p * q, I * J, I, J, K, sai, sak, sbk, sbj, J, 1, zb=r * t, alpha=alpha)
        ctx.cfg = (ta, tb, alpha)
        return c

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        ta, tb, alpha = ctx.cfg
        ga = gb = None
        if ctx.needs_input_grad[0]:      # d op_a(A) = gC op_b(B)^T
            ga = _Bmm.apply(b, gc, tb, True, alpha) if ta else _Bmm.apply(gc, b, False, not tb, alpha)
        if ctx.needs_input_grad[1]:      # d op_b(B) = op_a(A)^T gC
            gb = _Bmm.apply(gc, a, True, ta, alpha) if tb else _Bmm.apply(a, gc, not ta, False, alpha)
        return ga, gb, None, None, None


def bmm(a, b, ta=False, tb=False, alpha=1.0):
    """alpha * op_a(a) @ op_b(b) for [Z, ., .] fp32 tensors on the GPU, differentiable to any order on our own kernel"""
    if not (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32):
        raise RuntimeError('te_hip: expected fp32 tensors on the GPU (no CPU path exists)')
    return _Bmm.apply(a, b, bool(ta), bool(tb), float(alpha))


def _closed_expr(x, weight, bias, alpha, beta, act, residual):
    """the same function as the fused launch, built from any-order differentiable pieces that run on our kernels"""
    N, K = weight.shape
    y = _LinFwd.apply(x.reshape(-1, K), weight, alpha).reshape(*x.shape[:-1], N)
    if act == 'lrelu':
        from .fused_act import fused_leaky_relu
        y2 = y.reshape(-1, N)
        y = fused_leaky_relu(y2, None if bias is None else (bias * beta if beta != 1.0 else bias)).reshape(y.shape)
    else:
        if bias is not None:
            y = torch.add(y, bias, alpha=beta)
        if act == 'gelu':
            y = F.gelu(y)
    return y if residual is None else y + residual


def _rows(x, K):
    """[R, K] view of x with unit inner stride; a strided row slice (latent[:, i]) is used in place, without a copy."""
    if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= K:
        return x
    return x.reshape(-1, K).contiguous()


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, alpha, beta, act):
        N, K = weight.shape
        x2 = _rows(x, K)
        res2 = None if residual is None else residual.reshape(-1, N).contiguous()
        if K > MAX_K:       # wide reduction: K chunks over the grid + fixed-order second pass (deterministic)
            y, pre = _lib.small_gemm_splitk(x2.shape[0], N, K, _ksplit(K), x2, x2.stride(0), 1, weight.contiguous(), 1, K, bias, res2,
                                            alpha, beta, _ACT[act], want_pre=(act == 'gelu'))
        else:
            y, pre, _ = _lib.small_gemm(x2.shape[0], N, K, x2, x2.stride(0), 1, weight.contiguous(), 1, K, bias, res2, alpha, beta,
                                        _ACT[act], want_pre=(act == 'gelu'))
        ctx.save_for_backward(x, weight, bias, residual, pre if act == 'gelu' else (y if act == 'lrelu' else None))
        ctx.cfg = (alpha, beta, act)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, residual, aux = ctx.saved_tensors
        alpha, beta, act = ctx.cfg
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():
            with torch.enable_grad():
                al = [None if t is None else t.view_as(t) for t in (x, weight, bias, residual)]
                y = _closed_expr(al[0], al[1], al[2], alpha, beta, act, al[3])
                ins = [t for t, n in zip(al, need[:4]) if n and t is not None]
                gs = iter(torch.autograd.grad(y, ins, gy, create_graph=True, allow_unused=True))
            return tuple(next(gs) if (n and t is not None) else None for t, n in zip(al, need[:4])) + (None, None, None)
        N, K = weight.shape
        g = gy.reshape(-1, N).contiguous()
        # the skip connection hands on the DENSE gradient: a strided one (the permute in front of adjust_style makes the first)
        # would otherwise be densified again by both linear layers of every block upstream
        g_res = g.view(gy.shape) if (residual is not None and need[3]) else None
        if act == 'gelu':
            g = torch.ops.aten.gelu_backward(g, aux)
        elif act == 'lrelu':       # residual is not combined with lrelu anywhere in the model
            g = g * torch.where(aux > 0, math.sqrt(2), 0.2 * math.sqrt(2))
        R = g.shape[0]
        gx = gw = gb = None
        if need[0]:     # dx[R,K] = alpha * g[R,N] W[N,K]
            gx = _lib.small_gemm(R, K, N, g, N, 1, weight.contiguous(), K, 1, alpha=alpha)[0].reshape(x.shape)
        want_b = bias is not None and need[2]
        if need[1]:     # dW[N,K] = alpha * g^T[N,R] x[R,K];  db = beta * row sums of g^T come out of the same launch
            x2 = _rows(x, K)
            if R <= MAX_K:
                gw, _, gb = _lib.small_gemm(N, K, R, g, 1, N, x2, x2.stride(0), 1, alpha=alpha,
                                            rowsum_scale=beta if want_b else None)
            elif _ksplit(R):       # tall reductions (adjust_style: 8192 rows): the split-K form of the same kernel
                gw, _ = _lib.small_gemm_splitk(N, K, R, _ksplit(R), g, 1, N, x2, x2.stride(0), 1, alpha=alpha)
            else:
                # a row count that does not split evenly: whole 1024-row chunks through the split-K form, the remaining
                # rows (< 1024) through the single-pass kernel with the first part as its residual (no library GEMM)
                R1 = (R // MAX_K) * MAX_K
                gw, _ = _lib.small_gemm_splitk(N, K, R1, R1 // MAX_K, g, 1, N, x2, x2.stride(0), 1, alpha=alpha)
                gw = _lib.small_gemm(N, K, R - R1, g[R1:], 1, N, x2[R1:], x2.stride(0), 1, residual=gw, alpha=alpha)[0]
        if want_b and gb is None:
            gb = g.sum(0)
            gb = gb * beta if beta != 1.0 else gb
        return gx, gw, gb, g_res, None, None, None


def linear_fused(x, weight, bias=None, alpha=1.0, beta=1.0, act=None, residual=None):
    """act in (None, 'gelu', 'lrelu'); reductions wider than MAX_K run the split-K form of the same kernel."""
    if not (x.is_cuda and x.dtype == torch.float32):
        raise RuntimeError(f'te_hip: expected an fp32 tensor on the GPU, got {x.dtype} {x.device} (no CPU / library path exists)')
    if weight.shape[1] > MAX_K and not _ksplit(weight.shape[1]):
        raise RuntimeError(f'te_hip: linear layer with {weight.shape[1]} inputs: reductions wider than {MAX_K} must split into '
                           f'chunks of <= {MAX_K} that are multiples of 8 (no library fallback)')
    if _modconv_state.second_order and torch.is_grad_enabled():
        # the caller announced a create_graph backward (regulariser steps): build the differentiable form right away
        return _closed_expr(x, weight, bias, alpha, beta, act, residual)
    return _Linear.apply(x, weight, bias, residual, alpha, beta, act)


# ------------------------------------------------------------------------------------------------ several layers, one input
class _SharedInputLinears(Function):
    """y_z = alpha * x W_z^T + beta * b_z for nz <= 16 EqualLinear layers of equal shape that read the SAME input, as one
    batched launch (te_small_gemm_batched_f32 with a zero operand stride for x): the key / value projections of an
    attention block (model_spatial_query.py:889-890) and the query projections of all blocks after the first, which every
    block applies to the same P code (:675-678).  Backward: one batched launch for the nz weight gradients, one for the
    nz input-gradient slabs + their sum, bias gradients as one reduction."""

    @staticmethod
    def forward(ctx, x, alpha, beta, nz, *params):
        weights, biases = params[:nz], params[nz:]
        J, K = weights[0].shape
        x2 = x.reshape(-1, K).contiguous()
        R = x2.shape[0]
        buf = torch.empty(nz, R, J, device=x.device, dtype=x.dtype)
        base_w, base_b = weights[0].data_ptr(), biases[0].data_ptr()
        _lib.small_gemm_batched(buf, x2, weights[0], biases[0], nz, 0, R * J, R, J, K, K, 1, 1, K, J, 1,
                                b_tab=[(w.data_ptr() - base_w) // 4 for w in weights],
                                bias_tab=[(b.data_ptr() - base_b) // 4 for b in biases], alpha=alpha, beta=beta, act=0)
        ctx.save_for_backward(x, *params)
        ctx.cfg = (alpha, beta, nz)
        return tuple(buf[z].reshape(*x.shape[:-1], J) for z in range(nz))

    @staticmethod
    def backward(ctx, *gs):
        x, *params = ctx.saved_tensors
        alpha, beta, nz = ctx.cfg
        weights, biases = params[:nz], params[nz:]
        need = ctx.needs_input_grad
        J, K = weights[0].shape
        if torch.is_grad_enabled():              # recorded backward: layer by layer through the closed trio
            with torch.enable_grad():
                xa = x.view_as(x)
                pa = [p.view_as(p) for p in params]
                ys = [_closed_expr(xa, pa[z], pa[nz + z], alpha, beta, None, None) for z in range(nz)]
                ins = [t for t, f in zip([xa] + pa, [need[0]] + list(need[4:])) if f]
                res = iter(torch.autograd.grad(ys, ins, list(gs), create_graph=True, allow_unused=True))
            gx = next(res) if need[0] else None
            return (gx, None, None, None) + tuple(next(res) if f else None for f in need[4:])
        x2 = x.reshape(-1, K).contiguous()
        R = x2.shape[0]
        g = torch.stack([t.reshape(R, J) for t in gs])                      # [nz, R, J]
        gx = None
        if need[0]:          # slab_z[r, k] = alpha * sum_j g[z, r, j] W_z[j, k];  dx = sum_z slab_z
            base_w = weights[0].data_ptr()
            slabs = torch.empty(nz, R, K, device=g.device, dtype=g.dtype)
            _lib.small_gemm_batched(slabs, g, weights[0], None, nz, R * J, R * K, R, K, J, J, 1, K, 1, K, 1,
                                    b_tab=[(w.data_ptr() - base_w) // 4 for w in weights], alpha=alpha)
            gx = (slabs.sum(dim=0) if nz > 1 else slabs[0]).reshape(x.shape)
        gws, gbs = [None] * nz, [None] * nz
        if any(need[4:4 + nz]):      # dW_z[j, k] = alpha * sum_r g[z, r, j] x[r, k]   (reduction over the rows: split when tall)
            if R <= MAX_K:
                gw = torch.empty(nz, J, K, device=g.device, dtype=g.dtype)
                _lib.small_gemm_batched(gw, g, x2, None, nz, R * J, J * K, J, K, R, 1, J, K, 1, K, 1, zb=0, alpha=alpha)
                gws = [gw[z] if need[4 + z] else None for z in range(nz)]
            else:
                gws = [_LinDw.apply(g[z], x2, alpha) if need[4 + z] else None for z in range(nz)]
        if any(need[4 + nz:]):
            gb = g.sum(dim=1)
            gb = gb * beta if beta != 1.0 else gb
            gbs = [gb[z] if need[4 + nz + z] else None for z in range(nz)]
        return (gx, None, None, None) + tuple(gws) + tuple(gbs)


def shared_input_linears(x, mods):
    """[m(x) for m in mods] for EqualLinear modules of equal shape without activation, as one launch when the batched
    kernel applies (fp32 on the GPU, <= 16 layers, reduction <= MAX_K, not inside second_order())."""
    m0 = mods[0]
    ok = (x.is_cuda and x.dtype == torch.float32 and 1 < len(mods) <= 16 and m0.weight.shape[1] <= MAX_K
          and not _modconv_state.second_order
          and all(m.activation is None and m.bias is not None and m.weight.shape == m0.weight.shape and m.scale == m0.scale
                  and m.lr_mul == m0.lr_mul and m.weight.is_contiguous() and m.bias.is_contiguous() for m in mods))
    if not ok:
        return [m(x) for m in mods]
    return list(_SharedInputLinears.apply(x, float(m0.scale), float(m0.lr_mul), len(mods), *[m.weight for m in mods],
                                          *[m.bias for m in mods]))
