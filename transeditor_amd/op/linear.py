"""EqualLinear as ONE fused launch (te_small_gemm_f32): y = act(alpha * x W^T + beta * bias) + residual.

Reference: EqualLinear.forward, model_spatial_query.py:213-221 (F.linear(input, weight * scale, bias * lr_mul), optional
fused leaky-ReLU) and the residual / GELU around it in AttentionBlock.forward (:920-936).  alpha = scale, beta = lr_mul,
so the parameters are consumed as stored (no `weight * scale` / `bias * lr_mul` launches).  Backward: dx = alpha g W and
dW = alpha g^T x on the same kernel, db = beta * sum(g); a recorded backward (create_graph) goes through the
equivalent torch expression.
"""
import math

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib

_ACT = {None: 0, 'gelu': 1, 'lrelu': 3}
MAX_K = 1024          # widest reduction of the single-pass kernel; beyond it the split-K form (discriminator's 8192-wide linear)


def _ksplit(K):
    """number of K chunks for the split-K form: chunks of <= 1024 that are multiples of 8, or 0 if K does not split"""
    for S in range(-(-K // MAX_K), 65):
        if K % S == 0 and (K // S) % 8 == 0:
            return S
    return 0


def _torch_expr(x, weight, bias, alpha, beta, act, residual):
    y = F.linear(x, weight * alpha, None if bias is None else bias * beta)
    if act == 'gelu':
        y = F.gelu(y)
    elif act == 'lrelu':
        y = F.leaky_relu(y, 0.2) * math.sqrt(2)
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
                y = _torch_expr(al[0], al[1], al[2], alpha, beta, act, al[3])
                ins = [t for t, n in zip(al, need[:4]) if n and t is not None]
                gs = iter(torch.autograd.grad(y, ins, gy, create_graph=True, allow_unused=True))
            return tuple(next(gs) if (n and t is not None) else None for t, n in zip(al, need[:4])) + (None, None, None)
        N, K = weight.shape
        g = gy.reshape(-1, N).contiguous()
        g_res = gy if (residual is not None and need[3]) else None
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
                gw = torch.mm(g.t(), x2 if x2.is_contiguous() else x2.contiguous())
                gw = gw * alpha if alpha != 1.0 else gw
        if want_b and gb is None:
            gb = g.sum(0)
            gb = gb * beta if beta != 1.0 else gb
        return gx, gw, gb, g_res, None, None, None


def linear_fused(x, weight, bias=None, alpha=1.0, beta=1.0, act=None, residual=None):
    """act in (None, 'gelu', 'lrelu'); reductions wider than MAX_K run the split-K form of the same kernel."""
    if (weight.shape[1] > MAX_K and not _ksplit(weight.shape[1])) or not (x.is_cuda and x.dtype == torch.float32):
        if not x.is_cuda:
            raise RuntimeError('te_hip: expected a contiguous fp32 tensor on the GPU (no CPU path exists)')
        return _torch_expr(x, weight, bias, alpha, beta, act, residual)
    return _Linear.apply(x, weight, bias, residual, alpha, beta, act)
