"""Demodulation coefficient d[b,co] = rsqrt(sum_{ci,k} (wscale * w[co,ci,k] * s[b,ci])^2 + eps) as one launch.

Reference: ModulatedConv2d.forward, model_spatial_query.py:300-304 (there on B materialised weight copies).  The
shared-weight formulation only needs this [B,Co] coefficient; te_demod_fwd/bwd replace the pow / sum / mm / add / rsqrt
chain and its backward.  A recorded backward (create_graph) differentiates the equivalent torch expression.
"""
import torch
from torch.autograd import Function

from .. import _lib
from .linear import _LinFwd


def _torch_expr(w, s, wscale, eps):
    """the same function for the recorded backward; the [B,Ci] x [Ci,Co] product on the closed small-GEMM trio"""
    wsq = (w * wscale).pow(2).sum(dim=(2, 3))
    if s.is_cuda and s.dtype == torch.float32:
        return torch.rsqrt(_LinFwd.apply(s.pow(2), wsq, 1.0) + eps)
    return torch.rsqrt(s.pow(2) @ wsq.t() + eps)


class _Demod(Function):
    @staticmethod
    def forward(ctx, w, s, wscale, eps):
        w3 = w.reshape(w.shape[0], w.shape[1], -1).contiguous()
        d, wsq = _lib.demod_fwd(w3, s.contiguous(), wscale, eps)
        ctx.save_for_backward(w, s, d, wsq)          # (w, s: the inputs themselves - their history matters to a recorded backward)
        ctx.cfg = (wscale, eps)
        return d

    @staticmethod
    def backward(ctx, gd):
        w, s, d, wsq = ctx.saved_tensors
        wscale, eps = ctx.cfg
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():
            with torch.enable_grad():
                wa, sa = w.view_as(w), s.view_as(s)
                ins = [t for t, n in zip((wa, sa), need[:2]) if n]
                gs = iter(torch.autograd.grad(_torch_expr(wa, sa, wscale, eps), ins, gd, create_graph=True))
            return tuple(next(gs) if n else None for n in need[:2]) + (None, None)
        w3 = w.reshape(w.shape[0], w.shape[1], -1).contiguous()
        gw, gs = _lib.demod_bwd(gd, d, w3, wsq, s.contiguous(), wscale, want_w=need[0], want_s=need[1])
        return (gw.reshape(w.shape) if gw is not None else None), gs, None, None


def demod(w, s, wscale, eps=1e-8):
    """w [Co,Ci,k,k] raw (unscaled) weight, s [B,Ci] style scale -> d [B,Co]"""
    if s.shape[0] > 64 or w.shape[1] > 8192:
        return _torch_expr(w, s, wscale, eps)
    return _Demod.apply(w, s, wscale, eps)
