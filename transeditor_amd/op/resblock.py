"""The discriminator's ResBlock as ONE autograd node (reference: ResBlock.forward, model_spatial_query.py:780-798;
ConvLayer :731-777).

    y1  = lrelu(conv3x3(x, w1) + b1) * sqrt(2)                      conv1
    yb  = blur(y1, pad (2,2))                                       conv2[0]
    y2  = lrelu(conv3x3_stride2(yb, w2) + b2) * sqrt(2) / sqrt(2)   conv2[1:]   (the block's 1/sqrt(2) folded into the gain)
    xs  = blur(x, pad (1,1))[:, :, ::2, ::2]                        skip[0] + the stride of skip[1]: one down-sampling FIR pass
    out = conv1x1(xs, ws / sqrt(2)) + y2                            skip[1], the sum rides in the convolution's epilogue

Why a node of its own ("ResBlock-level autograd node"): as separate autograd nodes the block costs one
elementwise add forward (y2 + skip), one backward (the two gradients that meet at x: 3 x the input tensor through HBM) and
a full-resolution bias_act_bwd pass for conv1.  Here

* the forward sum is the residual epilogue of the 1x1 convolution (te_conv_res_f32);
* backward: the adjoint blur applies conv1's leaky-ReLU gradient in ITS epilogue and emits the bias-gradient partials
  (te_blur_gradact_f32), and the data gradient of conv1 adds the skip branch's gradient in its epilogue - no add kernels,
  no separate activation-gradient pass at the block's input resolution.

A backward that is itself recorded (R1: create_graph) re-derives everything through the any-order composite of the same
pieces (`_composite`), like the other fused nodes; `ResBlock.forward` takes that route directly under `second_order()`.
"""
import math

import torch
from torch.autograd import Function

from .. import _lib
from .modconv import _bwd_pack_kind, _composite, _dgrad_raw, _wgrad_plain, packed, packed2
from .upfirdn2d import _geometry, flipped_taps, upfirdn2d

_SQRT2 = math.sqrt(2.0)


def resblock_composite(x, w1, b1, w2, b2, ws, k_main, k_skip, s1, s2, ss, pad_main, pad_skip, gain):
    """the same function from any-order differentiable pieces"""
    y1 = _composite(x, w1, None, None, b1, True, '3x3', s1)
    yb = upfirdn2d(y1, k_main, pad=pad_main)
    y2 = _composite(yb, w2, None, None, b2, _SQRT2 * gain, 'down', s2)
    xs = upfirdn2d(x, k_skip, down=2, pad=pad_skip)
    return y2 + _composite(xs, ws, None, None, None, False, '1x1', ss * gain)


class _ResBlock(Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ws, k_main, k_skip, s1, s2, ss, pad_main, pad_skip, gain):
        x = x.contiguous()
        need = ctx.needs_input_grad
        front = need[0] or need[1] or need[2]                 # anything upstream of the blur wants a gradient
        H, W = x.shape[2], x.shape[3]
        # conv1 (+ its data-gradient packing when dx will be asked for)
        if need[0]:
            wp1, wp1b = packed2(w1, _lib.PACK_FWD, _bwd_pack_kind('3x3'), s1)
        else:
            wp1, wp1b = packed(w1, _lib.PACK_FWD, s1), None
        y1 = _lib.conv(x, wp1, _lib.CONV_3X3, w1.shape[0], H, W, None, None, b1, 3)
        pm = (pad_main[0], pad_main[1], pad_main[0], pad_main[1])
        yb = _lib.upfirdn2d_raw(y1, k_main, (1, 1), (1, 1), pm)
        if yb.shape[2] % 2 == 0 or yb.shape[3] % 2 == 0:
            raise RuntimeError(f'resblock: blurred size {tuple(yb.shape[2:])} is not (2h+1)x(2w+1)')
        h, w_ = (yb.shape[2] - 1) // 2, (yb.shape[3] - 1) // 2
        if front:
            wp2, wp2b = packed2(w2, _lib.PACK_FWD, _bwd_pack_kind('down'), s2)
        else:
            wp2, wp2b = packed(w2, _lib.PACK_FWD, s2), None
        act2 = 3 if abs(_SQRT2 * gain - _SQRT2) < 1e-6 else 4
        if act2 == 4 and abs(_SQRT2 * gain - 1.0) > 1e-6:
            raise RuntimeError(f'resblock: leaky-ReLU gain {_SQRT2 * gain} is not one the kernels fuse (sqrt(2) or 1)')
        y2 = _lib.conv(yb, wp2, _lib.CONV_S2, w2.shape[0], h, w_, None, None, b2, act2)
        ps = (pad_skip[0], pad_skip[1], pad_skip[0], pad_skip[1])
        xs = _lib.upfirdn2d_raw(x, k_skip, (1, 1), (2, 2), ps)
        if xs.shape[2:] != y2.shape[2:]:
            raise RuntimeError(f'resblock: branch sizes differ {tuple(xs.shape[2:])} vs {tuple(y2.shape[2:])}')
        if need[0]:
            wps, wpsb = packed2(ws, _lib.PACK_FWD, _bwd_pack_kind('1x1'), ss * gain)
        else:
            wps, wpsb = packed(ws, _lib.PACK_FWD, ss * gain), None
        out = _lib.conv(xs, wps, _lib.CONV_1X1, ws.shape[0], xs.shape[2], xs.shape[3], None, None, None, 0, res=y2)
        ctx.save_for_backward(x, w1, b1, w2, b2, ws, k_main, k_skip, y1, yb, y2, xs)
        ctx.packs = (wp1b, wp2b, wpsb)
        ctx.cfg = (s1, s2, ss, tuple(pad_main), tuple(pad_skip), gain, pm, ps)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w1, b1, w2, b2, ws, k_main, k_skip, y1, yb, y2, xs = ctx.saved_tensors
        s1, s2, ss, pad_main, pad_skip, gain, pm, ps = ctx.cfg
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():          # double backward requested: differentiate the composite (partial derivatives
            with torch.enable_grad():        # w.r.t. fresh aliases, everything stays connected for the next order)
                al = [t.view_as(t) for t in (x, w1, b1, w2, b2, ws)]
                y = resblock_composite(*al, k_main, k_skip, s1, s2, ss, pad_main, pad_skip, gain)
                ins = [t for t, n in zip(al, need[:6]) if n]
                gs = iter(torch.autograd.grad(y, ins, g, create_graph=True, allow_unused=True))
            return tuple(next(gs) if n else None for n in need[:6]) + (None,) * 8
        wp1b, wp2b, wpsb = ctx.packs
        g = g.contiguous()
        front = need[0] or need[1] or need[2]
        gx = gw1 = gb1 = gw2 = gb2 = gws = None
        # ---- main branch, from the tail: activation gradient of conv2 (quarter resolution), its data / weight gradient
        g2, gb2 = _lib.bias_act_bwd(g, y2, 0.2, _SQRT2 * gain, want_bias=need[4])
        if need[3]:
            gw2 = _wgrad_plain(g2, yb, 'down', 3, s2)
        g1 = None
        if front:
            g_yb = _dgrad_raw(g2, w2, 'down', wscale=s2, wp=wp2b)
            _, g_pad = _geometry(y1.shape[2:], k_main.shape, (1, 1), (1, 1), pm)
            if tuple(k_main.shape) == (4, 4) and g_yb.shape[3] >= 4:
                # adjoint blur with conv1's leaky-ReLU gradient in the epilogue + bias-gradient partials: one pass
                g1, gb1 = _lib.blur_gradact(g_yb, y1, flipped_taps(k_main), g_pad, 0.2, _SQRT2)
            else:
                g1 = _lib.upfirdn2d_raw(g_yb, flipped_taps(k_main), (1, 1), (1, 1), g_pad)
                g1, gb1 = _lib.bias_act_bwd(g1, y1, 0.2, _SQRT2, want_bias=need[2])
            if not need[2]:
                gb1 = None
            if need[1]:
                gw1 = _wgrad_plain(g1, x, '3x3', 3, s1)
        # ---- skip branch
        if need[5]:
            gws = _wgrad_plain(g, xs, '1x1', 1, ss * gain)
        if need[0]:
            g_xs = _dgrad_raw(g, ws, '1x1', wscale=ss * gain, wp=wpsb)
            _, gp_s = _geometry(x.shape[2:], k_skip.shape, (1, 1), (2, 2), ps)
            gx_b = _lib.upfirdn2d_raw(g_xs, flipped_taps(k_skip), (2, 2), (1, 1), gp_s)
            # data gradient of conv1 + the skip branch's gradient in its epilogue
            gx = _lib.conv(g1, wp1b, _lib.CONV_3X3, w1.shape[1], x.shape[2], x.shape[3], None, None, None, 0, res=gx_b)
        return (gx, gw1, gb1, gw2, gb2, gws) + (None,) * 8


def resblock(x, w1, b1, w2, b2, ws, k_main, k_skip, s1, s2, ss, pad_main, pad_skip, gain):
    return _ResBlock.apply(x, w1, b1, w2, b2, ws, k_main, k_skip, float(s1), float(s2), float(ss), tuple(pad_main),
                           tuple(pad_skip), float(gain))
