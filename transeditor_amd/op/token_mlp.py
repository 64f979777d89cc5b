"""The token-wise mapping network as one launch per pass (te_small_gemm_batched_f32).

Reference: Generator.forward, model_spatial_query.py:626-646 — token i of the (pixel-normalised) code goes through its
own EqualLinear(512, 512, lr_mul, 'fused_lrelu') (:547-566): 16 F.linear + 16 fused_leaky_relu launches and 32 slice
copies per network.  Here: forward = one batched GEMM with the bias + leaky-ReLU epilogue, reading the [B, D, tokens]
code in place and writing [B, tokens, D]; backward = one te_bias_act_bwd pass (activation gradient + all bias
gradients), one batched GEMM for dx (written straight into the [B, D, tokens] layout) and one for the 16 dW.  A recorded
backward (create_graph) differentiates the equivalent torch expression.
"""
import math

import torch
from torch.autograd import Function

from .. import _lib
from .fused_act import fused_leaky_relu
from .linear import bmm


def _torch_expr(x, weights, biases, scale, lr_mul):
    """x [B, D, C] -> [B, T, D]"""
    T = len(weights)
    W = torch.stack(list(weights)) * scale                              # [T, out, in]
    bias = torch.cat(list(biases)) * lr_mul
    y = bmm(x[:, :, :T].permute(2, 0, 1), W, False, True)               # [T, B, out]  (closed family: any order on our kernel)
    B, D = y.shape[1], y.shape[2]
    return fused_leaky_relu(y.permute(1, 0, 2).reshape(B, T * D), bias).view(B, T, D)


def _offsets(ts):
    """element offsets of separately allocated per-token operands relative to the first one"""
    base = ts[0].data_ptr()
    return [(t.data_ptr() - base) // 4 for t in ts]


def _kernel_ready(x, params):
    """the batched launch reads every weight / bias through a raw offset: each must be a dense fp32 tensor on x's device"""
    return all(t.is_contiguous() and t.dtype == torch.float32 and t.device == x.device for t in params)


class _TokenMLP(Function):
    @staticmethod
    def forward(ctx, x, scale, lr_mul, T, *params):
        weights, biases = params[:T], params[T:]
        x_in, x = x, x.contiguous()       # saved: the input itself (its history matters when the backward is recorded), not a copy
        B, D, Cn = x.shape
        N = weights[0].shape[0]
        y = torch.empty(B, T, N, device=x.device, dtype=x.dtype)
        # A_t(b, k) = x[b, k, t];  B_t(k, j) = W_t[j, k];  C_t(b, j) = y[b, t, j]
        _lib.small_gemm_batched(y, x, weights[0], biases[0], T, 1, N, B, N, D, D * Cn, Cn, 1, D, T * N, 1,
                                b_tab=_offsets(weights), bias_tab=_offsets(biases), alpha=scale, beta=lr_mul, act=3)
        ctx.save_for_backward(x_in, y, *params)
        ctx.cfg = (scale, lr_mul, T)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, *params = ctx.saved_tensors
        scale, lr_mul, T = ctx.cfg
        weights, biases = params[:T], params[T:]
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():
            with torch.enable_grad():
                xa = x.view_as(x)
                pa = [p.view_as(p) for p in params]
                out = _torch_expr(xa, pa[:T], pa[T:], scale, lr_mul)
                ins = [t for t, n in zip([xa] + pa, [need[0]] + list(need[4:])) if n]
                gs = iter(torch.autograd.grad(out, ins, gy, create_graph=True, allow_unused=True))
            gx = next(gs) if need[0] else None
            return (gx, None, None, None) + tuple(next(gs) if n else None for n in need[4:])
        x = x.contiguous()
        B, D, Cn = x.shape
        N = weights[0].shape[0]
        # activation gradient and the 16 bias gradients in one pass over [B, T*N]
        gp, gb = _lib.bias_act_bwd(gy.reshape(B, T * N), y.view(B, T * N), 0.2, math.sqrt(2), want_bias=True)
        gx = gw = None
        if need[0]:      # gx[b, k, t] = scale * sum_j gp[b, t, j] W_t[j, k]   (tokens >= T get no gradient)
            gx = torch.zeros_like(x) if T < Cn else torch.empty_like(x)
            _lib.small_gemm_batched(gx, gp, weights[0], None, T, N, 1, B, D, N, T * N, 1, D, 1, D * Cn, Cn,
                                    b_tab=_offsets(weights), alpha=scale)
        if any(need[4:4 + T]):   # gW_t[j, k] = scale * sum_b gp[b, t, j] x[b, k, t]
            gw = torch.empty(T, N, D, device=x.device, dtype=x.dtype)
            _lib.small_gemm_batched(gw, gp, x, None, T, N, N * D, N, D, B, 1, T * N, D * Cn, Cn, D, 1, zb=1, alpha=scale)
        gws = tuple(gw[t] if (gw is not None and need[4 + t]) else None for t in range(T))
        gb = gb * lr_mul if lr_mul != 1.0 else gb
        gbs = tuple(gb[t * N:(t + 1) * N] if need[4 + T + t] else None for t in range(T))
        return (gx, None, None, None) + gws + gbs


def token_mlp(x, weights, biases, scale, lr_mul):
    """x [B, D, C] (already pixel-normalised) -> [B, T, D]: y[b, t] = lrelu(scale * W_t x[b, :, t] + lr_mul * b_t) * sqrt(2)."""
    T = len(weights)
    if T > 16 or not x.is_cuda:
        if not x.is_cuda:
            raise RuntimeError('te_hip: expected an fp32 tensor on the GPU (no CPU path exists)')
        return _torch_expr(x, weights, biases, scale, lr_mul)
    if not _kernel_ready(x, list(weights) + list(biases)):
        raise RuntimeError('te_hip: token_mlp needs contiguous fp32 weights / biases on the input\'s device '
                           '(call .contiguous() on reassigned parameters)')
    return _TokenMLP.apply(x, scale, lr_mul, T, *weights, *biases)
