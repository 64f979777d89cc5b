"""Blur (upfirdn2d, up=down=1) with bias + leaky-ReLU fused into the FIR kernel's epilogue.

Replaces the reference's two passes `self.blur(out)` (model_spatial_query.py:321) followed by
`FusedLeakyReLU` (:401) on the upsampling StyledConv layers with one read of the (2H+1)^2
tensor and one write of the (2H)^2 result.  Backward: one te_bias_act_bwd pass (gradient of
the activation + bias gradient) and one adjoint FIR pass; a recorded backward (create_graph)
falls back to the composition of the two twice-differentiable ops.
"""
import torch
from torch.autograd import Function

from .. import _lib
from .fused_act import fused_leaky_relu
from .upfirdn2d import upfirdn2d, _geometry, flipped_taps


class _BlurBiasAct(Function):
    @staticmethod
    def forward(ctx, x, kernel, bias, pad):
        pad4 = (pad[0], pad[1], pad[0], pad[1])
        out = _lib.upfirdn2d_raw(x, kernel, (1, 1), (1, 1), pad4, bias=bias.contiguous(), act=3, alpha=0.2,
                                 scale=2 ** 0.5)
        ctx.save_for_backward(x, kernel, bias, out)
        ctx.pad = pad
        return out

    @staticmethod
    def backward(ctx, g):
        x, kernel, bias, out = ctx.saved_tensors
        pad = ctx.pad
        if torch.is_grad_enabled():
            with torch.enable_grad():
                xa, ba = x.view_as(x), bias.view_as(bias)          # aliases: partial derivatives only
                y = fused_leaky_relu(upfirdn2d(xa, kernel, pad=pad), ba)
                gx, gb = torch.autograd.grad(y, (xa, ba), g, create_graph=True, allow_unused=True)
            return gx, None, gb, None
        pad4 = (pad[0], pad[1], pad[0], pad[1])
        _, g_pad = _geometry(x.shape[2:], kernel.shape, (1, 1), (1, 1), pad4)
        if tuple(kernel.shape) == (4, 4) and min(g_pad) >= 0:
            # activation gradient applied while the adjoint FIR stages its input tile; bias gradient from per-tile sums
            gx, gb = _lib.blur_actgrad(g, out, flipped_taps(kernel), g_pad, 0.2, 2 ** 0.5)
        else:
            gpre, gb = _lib.bias_act_bwd(g, out, 0.2, 2 ** 0.5, want_bias=True)
            gx = _lib.upfirdn2d_raw(gpre, flipped_taps(kernel), (1, 1), (1, 1), g_pad)
        return gx, None, gb, None


def blur_bias_act(x, kernel, bias, pad):
    return _BlurBiasAct.apply(x, kernel, bias, tuple(pad))
