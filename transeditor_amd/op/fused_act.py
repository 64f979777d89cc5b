"""fused_leaky_relu / FusedLeakyReLU on the te_bias_act kernels.

Call surface and gradient structure follow utils/op/fused_act.py:18-90 of the reference:
forward saves the OUTPUT; backward is gi = g * slope(out) * scale with the bias gradient the
sum over all dims but 1; the backward is itself differentiable (grad-grad = the same slope
mask applied to ggi + ggb[c]).  Unlike the reference, gi and the bias reduction come out of a
single kernel pass (te_bias_act_bwd_f32).
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import _lib


class _LReluBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, has_bias, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale, ctx.has_bias = negative_slope, scale, has_bias
        gi, gb = _lib.bias_act_bwd(grad_output, out, negative_slope, scale, want_bias=has_bias)
        if not has_bias:
            gb = grad_output.new_zeros(0)
        return gi, gb

    @staticmethod
    def backward(ctx, ggi, ggb):
        out, = ctx.saved_tensors
        if ggi is None:
            ggi = torch.zeros_like(out)
        b = ggb.contiguous() if (ctx.has_bias and ggb is not None) else None
        ggo = _lib.bias_act(ggi, b, out, 3, 1, ctx.negative_slope, ctx.scale)
        return ggo, None, None, None, None


class _LRelu(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = _lib.bias_act(input, bias.contiguous() if bias is not None else None, None, 3, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale, ctx.has_bias = negative_slope, scale, bias is not None
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        gi, gb = _LReluBackward.apply(grad_output, out, ctx.has_bias, ctx.negative_slope, ctx.scale)
        return gi, (gb if ctx.has_bias else None), None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return _LRelu.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
