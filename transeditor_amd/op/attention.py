"""Attention core (softmax(scale * q k^T) v per head) on the te_attn kernels.

Reference: Attention.forward, model_spatial_query.py:888-894.  q [N,M,G*D], k/v [N,L,G*D]
token-major; returns (o [N,M,G*D], sim [N,G,M,L]).  QK^T and sim.V run on fp32 MFMA.
First-order backward is a kernel; a backward that is itself recorded (create_graph) goes
through the equivalent torch expression so higher orders stay available.
"""
import torch
from torch.autograd import Function

from .. import _lib
from .linear import bmm


def _torch_expr(q, k, v, scale, groups):
    """the same function for the recorded backward: the two batched products on the closed small-GEMM family
    (op/linear.py::_Bmm, any order on our kernel), softmax and the head reshapes as framework elementwise ops"""
    N, M, C = q.shape
    L, D = k.shape[1], C // groups
    qh = q.reshape(N, M, groups, D).permute(0, 2, 1, 3).reshape(N * groups, M, D)
    kh = k.reshape(N, L, groups, D).permute(0, 2, 1, 3).reshape(N * groups, L, D)
    vh = v.reshape(N, L, groups, D).permute(0, 2, 1, 3).reshape(N * groups, L, D)
    sim = torch.softmax(bmm(qh, kh, False, True, scale), dim=2)                      # [N*G, M, L]
    o = bmm(sim, vh).reshape(N, groups, M, D).permute(0, 2, 1, 3).reshape(N, M, C)
    return o, sim.reshape(N, groups, M, L)


class _AttnCore(Function):
    @staticmethod
    def forward(ctx, q, k, v, scale, groups):
        o, sim = _lib.attn_fwd(q, k, v, scale, groups)
        ctx.save_for_backward(q, k, v, sim)
        ctx.scale, ctx.groups = scale, groups
        ctx.set_materialize_grads(False)      # an unused `similarity` output must not cost a zero-fill per block
        return o, sim

    @staticmethod
    def backward(ctx, go, gsim):
        q, k, v, sim = ctx.saved_tensors
        if torch.is_grad_enabled():
            with torch.enable_grad():
                q, k, v = q.view_as(q), k.view_as(k), v.view_as(v)   # aliases: partial derivatives only
                o2, sim2 = _torch_expr(q, k, v, ctx.scale, ctx.groups)
                outs, gouts = [o2], [go if go is not None else torch.zeros_like(o2)]
                if gsim is not None:
                    outs.append(sim2)
                    gouts.append(gsim)
                gq, gk, gv = torch.autograd.grad(outs, (q, k, v), gouts, create_graph=True, allow_unused=True)
            return gq, gk, gv, None, None
        if go is None:
            go = torch.zeros_like(q)
        gq, gk, gv = _lib.attn_bwd(go, gsim, q.contiguous(), k.contiguous(), v.contiguous(), sim, ctx.scale, ctx.groups)
        return gq, gk, gv, None, None


def attention_core(q, k, v, scale, groups=4):
    return _AttnCore.apply(q.contiguous(), k.contiguous(), v.contiguous(), scale, groups)
