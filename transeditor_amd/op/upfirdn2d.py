"""upfirdn2d on the te_upfirdn2d kernels; call surface of utils/op/upfirdn2d.py:143-148.

Gradient structure (utils/op/upfirdn2d.py:101-112, 29-40, 66-81): the adjoint is the same op
with up/down swapped, the taps flipped and g_pad0 = k - pad0 - 1,
g_pad1 = in*up - out*down + pad0 - up + 1; the adjoint's adjoint is the forward op again, so
the op is differentiable to any order.  The FIR taps are a constant buffer (no gradient).
"""
import torch
from torch.autograd import Function

from .. import _lib


def _geometry(in_hw, k_hw, up, down, pad):
    (ih, iw), (kh, kw) = in_hw, k_hw
    px0, px1, py0, py1 = pad
    oh = (ih * up[1] + py0 + py1 - kh) // down[1] + 1
    ow = (iw * up[0] + px0 + px1 - kw) // down[0] + 1
    g_pad = (kw - px0 - 1, iw * up[0] - ow * down[0] + px0 - up[0] + 1,
             kh - py0 - 1, ih * up[1] - oh * down[1] + py0 - up[1] + 1)
    return (oh, ow), g_pad


_FLIPPED = {}


def flipped_taps(kernel):
    """torch.flip(kernel, [0, 1]) memoised per FIR buffer (a constant of the module): the adjoint passes would
    otherwise launch one flip kernel per layer per step.  The entry keeps the source tensor alive, so its id stays valid."""
    ent = _FLIPPED.get(id(kernel))
    if ent is None or ent[0] is not kernel or ent[1] != kernel._version:
        if len(_FLIPPED) > 256:
            _FLIPPED.clear()
        ent = (kernel, kernel._version, torch.flip(kernel, [0, 1]).contiguous())
        _FLIPPED[id(kernel)] = ent
    return ent[2]


class _UpFirDnAdjoint(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, kernel_flipped, up, down, pad, g_pad, in_hw):
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad)
        gi = _lib.upfirdn2d_raw(grad_output, kernel_flipped, down, up, g_pad)
        assert gi.shape[2:] == tuple(in_hw), (gi.shape, in_hw)
        return gi

    @staticmethod
    def backward(ctx, gg_input):
        kernel, = ctx.saved_tensors
        up, down, pad = ctx.cfg
        return _UpFirDn.apply(gg_input, kernel, up, down, pad), None, None, None, None, None, None, None


class _UpFirDn(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        _, g_pad = _geometry(input.shape[2:], kernel.shape, up, down, pad)
        ctx.save_for_backward(kernel, flipped_taps(kernel))
        ctx.cfg = (up, down, pad, g_pad, tuple(input.shape[2:]))
        return _lib.upfirdn2d_raw(input, kernel, up, down, pad)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, flipped = ctx.saved_tensors
        up, down, pad, g_pad, in_hw = ctx.cfg
        gi = _UpFirDnAdjoint.apply(grad_output, kernel, flipped, up, down, pad, g_pad, in_hw)
        return gi, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    return _UpFirDn.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
