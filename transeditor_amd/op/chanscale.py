"""x * s[:, :, None, None] and its adjoint as a pair of kernels closed under differentiation (te_chan_scale_f32 /
te_chan_dot_f32).  The any-order composite of the modulated convolution (op/modconv.py) applies the style scale and the
demodulation to activations this way; when a backward is recorded (path-length regulariser, model_spatial_query.py:299-304 +
train_spatial_query.py:92-109) the framework's broadcast multiply, its backward multiply and the reduction for the scale's
gradient are each one pass here instead of two or three."""
from torch.autograd import Function

from .. import _lib


class _ChanScale(Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.save_for_backward(x, s)
        return _lib.chan_scale(x, s)

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        gx = _ChanScale.apply(g, s) if ctx.needs_input_grad[0] else None
        gs = _ChanDot.apply(g, x) if ctx.needs_input_grad[1] else None
        return gx, gs


class _ChanDot(Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _lib.chan_dot(a, b)

    @staticmethod
    def backward(ctx, c):
        a, b = ctx.saved_tensors
        ga = _ChanScale.apply(b, c) if ctx.needs_input_grad[0] else None
        gb = _ChanScale.apply(a, c) if ctx.needs_input_grad[1] else None
        return ga, gb


USE_KERNELS = True      # False: the framework's broadcast expression (A/B measurements)


def chan_scale(x, s):
    """x [B,C,H,W] (or [B,C,...]) times s [B,C]"""
    if USE_KERNELS and x.is_cuda and x.dtype == s.dtype and s.shape == x.shape[:2]:
        return _ChanScale.apply(x, s)
    return x * s.reshape(*s.shape, *([1] * (x.ndim - 2)))
