"""F.layer_norm(x, x.size()[1:]) (no affine) of the attention blocks on te_layer_norm_fwd/bwd.

Reference: AttentionBlock.forward, model_spatial_query.py:924 / 931.  A recorded backward (create_graph) differentiates
the torch expression; shapes the kernel does not cover (rows longer than 16384 elements, N % 4 != 0) use it directly.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib

EPS = 1e-5


class _SampleLayerNorm(Function):
    @staticmethod
    def forward(ctx, x):
        x2 = x.reshape(x.shape[0], -1).contiguous()
        y, stats = _lib.layer_norm_fwd(x2, EPS)
        ctx.save_for_backward(x, y, stats)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        x, y, stats = ctx.saved_tensors
        if torch.is_grad_enabled():
            with torch.enable_grad():
                xa = x.view_as(x)
                gx, = torch.autograd.grad(F.layer_norm(xa, xa.shape[1:], eps=EPS), xa, g, create_graph=True)
            return gx
        return _lib.layer_norm_bwd(g.reshape(y.shape).contiguous(), y, stats).view(x.shape)


def sample_layer_norm(x):
    """normalise every sample x[b] over all of its elements (eps 1e-5, no affine)"""
    n = x[0].numel() if x.shape[0] > 0 else 0
    if not (x.is_cuda and x.dtype == torch.float32) or n == 0 or not _lib.layer_norm_supported(x.shape[0], n):
        if not x.is_cuda:
            raise RuntimeError('te_hip: expected an fp32 tensor on the GPU (no CPU path exists)')
        return F.layer_norm(x, x.shape[1:], eps=EPS)
    return _SampleLayerNorm.apply(x)


def _pixel_norm_expr(x, dim):
    return x * torch.rsqrt(x.pow(2).mean(dim=dim, keepdim=True) + 1e-8)


class _PixelNorm(Function):
    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        y, r = _lib.pixel_norm_fwd(xc, 1e-8)
        ctx.save_for_backward(x, y, r)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, r = ctx.saved_tensors
        if torch.is_grad_enabled():
            with torch.enable_grad():
                xa = x.view_as(x)
                gx, = torch.autograd.grad(_pixel_norm_expr(xa, 1), xa, g, create_graph=True)
            return gx
        return _lib.pixel_norm_bwd(g.contiguous(), y, r)


def pixel_norm(x, dim):
    """PixelNorm.forward (model_spatial_query.py:80-81); the kernel covers the configuration every script uses
    (3-D codes [B, D, C], dim = 1), anything else is the torch expression."""
    if not x.is_cuda:
        raise RuntimeError('te_hip: expected an fp32 tensor on the GPU (no CPU path exists)')
    if (x.dtype == torch.float32 and x.dim() == 3 and dim == 1
            and _lib.pixel_norm_supported(x.shape[0], x.shape[1], x.shape[2])):
        return _PixelNorm.apply(x)
    return _pixel_norm_expr(x, dim)
