"""Minibatch standard deviation + channel concat of the discriminator as one launch (te_minibatch_stddev_*).

Reference: Discriminator.forward, model_spatial_query.py:844-852 (view / var / sqrt / mean / repeat / cat).  A recorded
backward (create_graph: the R1 regulariser differentiates D twice) goes through the equivalent torch expression.
"""
import torch
from torch.autograd import Function

from .. import _lib

EPS = 1e-8


def _torch_expr(out, group, feat=1, chunks=1):
    if chunks > 1:
        return torch.cat([_torch_expr(o, group, feat) for o in out.chunk(chunks)], 0)
    batch, channel, height, width = out.shape
    sd = out.view(group, -1, feat, channel // feat, height, width)
    sd = torch.sqrt(sd.var(0, unbiased=False) + EPS).mean([2, 3, 4], keepdims=True).squeeze(2)
    return torch.cat([out, sd.repeat(group, 1, height, width)], 1)


class _Stddev(Function):
    @staticmethod
    def forward(ctx, x, group, chunks):
        ctx.save_for_backward(x)          # the input itself (history for a recorded backward); the binding makes its own dense copy
        ctx.group, ctx.chunks = group, chunks
        return _lib.minibatch_stddev_fwd(x, group, EPS, chunks)

    @staticmethod
    def backward(ctx, gy):
        x, = ctx.saved_tensors
        if torch.is_grad_enabled():
            with torch.enable_grad():
                xa = x.view_as(x)
                gx, = torch.autograd.grad(_torch_expr(xa, ctx.group, 1, ctx.chunks), xa, gy, create_graph=True)
            return gx, None, None
        return _lib.minibatch_stddev_bwd(gy, x.contiguous(), ctx.group, EPS, ctx.chunks), None, None


def minibatch_stddev(out, group=4, feat=1, second_order=False, chunks=1):
    """out [B, C, H, W] -> [B, C + feat, H, W]: the input with its minibatch-stddev channel appended.  `chunks` > 1: the
    batch holds `chunks` independent minibatches of equal size laid end to end (Discriminator.forward(..., chunks=))."""
    if out.shape[0] % chunks:
        raise ValueError(f'batch of {out.shape[0]} does not split into {chunks} equal minibatches')
    group = min(out.shape[0] // chunks, group)
    if not (out.is_cuda and out.dtype == torch.float32):
        raise RuntimeError('te_hip: expected an fp32 tensor on the GPU (no CPU path exists)')
    if feat != 1 or group > 4 or (out.shape[0] // chunks) % group or (second_order and torch.is_grad_enabled()):
        return _torch_expr(out, group, feat, chunks)   # shapes outside the kernel / a forward known to be differentiated twice
    return _Stddev.apply(out, group, chunks)
