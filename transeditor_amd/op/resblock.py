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
from .modconv import (_composite, _dgrad_raw, _wgrad_plain, bwd_kinds, cache_of, fwd_kinds, keep_cache, packed, packed2,
                      plain_1x1_kinds)
from .upfirdn2d import _geometry, flipped_taps, upfirdn2d

_SQRT2 = math.sqrt(2.0)


def resblock_composite(x, w1, b1, w2, b2, ws, k_main, k_skip, s1, s2, ss, pad_main, pad_skip, gain, w0=None, b0=None, s0=1.0):
    """the same function from any-order differentiable pieces (w0 / b0: the from-RGB stem in front, x is the image then)"""
    if w0 is not None:
        x = _composite(x, w0, None, None, b0, True, '1x1', s0)
    y1 = _composite(x, w1, None, None, b1, True, '3x3', s1)
    yb = upfirdn2d(y1, k_main, pad=pad_main)
    y2 = _composite(yb, w2, None, None, b2, _SQRT2 * gain, 'down', s2)
    xs = upfirdn2d(x, k_skip, down=2, pad=pad_skip)
    return y2 + _composite(xs, ws, None, None, None, False, '1x1', ss * gain)


class _ResBlock(Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ws, k_main, k_skip, s1, s2, ss, pad_main, pad_skip, gain, w0, b0, s0):
        x = x.contiguous()
        need = ctx.needs_input_grad
        keep_cache(ctx)
        # Optional from-RGB stem in front (the discriminator's ConvLayer(3, C, 1), :815, + its first ResBlock as one node):
        # x is the image, the block's input is lrelu(conv1x1(x, w0) + b0) * sqrt(2).  The stem's activation gradient then
        # rides in the epilogue of conv1's data gradient (te_conv_res_f32's mask stage) instead of a pass of its own.
        ctx.stem = w0 is not None
        img = None
        need_x = need[0]
        if ctx.stem:
            img = x
            x = _lib.rgb_expand(img, w0.reshape(w0.shape[0], 3).t(), b0, 3, s0)
            need_x = need[0] or need[14] or need[15]          # the block's input gradient feeds dimg, dw0, db0
        front = need_x or need[1] or need[2]                  # anything upstream of the blur wants a gradient
        H, W = x.shape[2], x.shape[3]
        # conv1 (+ its data-gradient packing when dx will be asked for)
        pk1, ck1 = fwd_kinds('3x3', x.shape[0], w1, H, W)       # (the Winograd form where it applies: op/modconv.py)
        if need_x:
            wp1, wp1b = packed2(w1, pk1, bwd_kinds('3x3', x.shape[0], w1, H, W)[0], s1)
        else:
            wp1, wp1b = packed(w1, pk1, s1), None
        y1 = _lib.conv(x, wp1, ck1, w1.shape[0], H, W, None, None, b1, 3)
        pm = (pad_main[0], pad_main[1], pad_main[0], pad_main[1])
        yb = _lib.upfirdn2d_raw(y1, k_main, (1, 1), (1, 1), pm)
        if yb.shape[2] % 2 == 0 or yb.shape[3] % 2 == 0:
            raise RuntimeError(f'resblock: blurred size {tuple(yb.shape[2:])} is not (2h+1)x(2w+1)')
        h, w_ = (yb.shape[2] - 1) // 2, (yb.shape[3] - 1) // 2
        pk2, ck2 = fwd_kinds('down', x.shape[0], w2, h, w_)     # (the split-bf16 form of the strided convolution where it applies)
        if front:
            wp2, wp2b = packed2(w2, pk2, bwd_kinds('down', x.shape[0], w2, h, w_)[0], s2)     # (the layout _dgrad_raw's launch will ask for)
        else:
            wp2, wp2b = packed(w2, pk2, s2), None
        act2 = 3 if abs(_SQRT2 * gain - _SQRT2) < 1e-6 else 4
        if act2 == 4 and abs(_SQRT2 * gain - 1.0) > 1e-6:
            raise RuntimeError(f'resblock: leaky-ReLU gain {_SQRT2 * gain} is not one the kernels fuse (sqrt(2) or 1)')
        y2 = _lib.conv(yb, wp2, ck2, w2.shape[0], h, w_, None, None, b2, act2)
        ps = (pad_skip[0], pad_skip[1], pad_skip[0], pad_skip[1])
        xs = _lib.upfirdn2d_raw(x, k_skip, (1, 1), (2, 2), ps)
        if xs.shape[2:] != y2.shape[2:]:
            raise RuntimeError(f'resblock: branch sizes differ {tuple(xs.shape[2:])} vs {tuple(y2.shape[2:])}')
        # the skip branch's 1x1 product (+ the main branch as its residual) and, in backward, its data gradient: the split-bf16 kernel
        # of csrc/p1s6.hip where it applies (round 6)
        pks, cks = plain_1x1_kinds(x.shape[0], ws, xs.shape[2], xs.shape[3])
        if need_x:
            wps, wpsb = packed2(ws, pks, plain_1x1_kinds(x.shape[0], ws, xs.shape[2], xs.shape[3], dgrad=True)[0], ss * gain)
        else:
            wps, wpsb = packed(ws, pks, ss * gain), None
        out = _lib.conv(xs, wps, cks, ws.shape[0], xs.shape[2], xs.shape[3], None, None, None, 0, res=y2)
        ctx.save_for_backward(x, w1, b1, w2, b2, ws, k_main, k_skip, y1, yb, y2, xs, img, w0, b0)
        ctx.packs = (wp1b, wp2b, wpsb)
        ctx.cfg = (s1, s2, ss, tuple(pad_main), tuple(pad_skip), gain, pm, ps, s0, need_x)
        return out

    @staticmethod
    def backward(ctx, g):
        with cache_of(ctx):
            return _ResBlock._backward(ctx, g)

    @staticmethod
    def _backward(ctx, g):
        x, w1, b1, w2, b2, ws, k_main, k_skip, y1, yb, y2, xs, img, w0, b0 = ctx.saved_tensors
        s1, s2, ss, pad_main, pad_skip, gain, pm, ps, s0, need_x = ctx.cfg
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():          # double backward requested: differentiate the composite (partial derivatives
            with torch.enable_grad():        # w.r.t. fresh aliases, everything stays connected for the next order)
                src = (img if ctx.stem else x, w1, b1, w2, b2, ws)
                al = [t.view_as(t) for t in src]
                st = [w0.view_as(w0), b0.view_as(b0)] if ctx.stem else [None, None]
                y = resblock_composite(*al, k_main, k_skip, s1, s2, ss, pad_main, pad_skip, gain, st[0], st[1], s0)
                flags = list(need[:6]) + ([need[14], need[15]] if ctx.stem else [])
                ins = [t for t, n in zip(al + (st if ctx.stem else []), flags) if n]
                gs = iter(torch.autograd.grad(y, ins, g, create_graph=True, allow_unused=True))
            first = tuple(next(gs) if n else None for n in need[:6])
            tail = (next(gs) if need[14] else None, next(gs) if need[15] else None) if ctx.stem else (None, None)
            return first + (None,) * 8 + tail + (None,)
        wp1b, wp2b, wpsb = ctx.packs
        g = g.contiguous()
        front = need_x or need[1] or need[2]
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
        gw0 = gb0 = None
        if need_x:
            g_xs = _lib.conv(g, wpsb, plain_1x1_kinds(x.shape[0], ws, xs.shape[2], xs.shape[3], dgrad=True)[1], ws.shape[1],
                             xs.shape[2], xs.shape[3])
            _, gp_s = _geometry(x.shape[2:], k_skip.shape, (1, 1), (2, 2), ps)
            gx_b = _lib.upfirdn2d_raw(g_xs, flipped_taps(k_skip), (2, 2), (1, 1), gp_s)
            # data gradient of conv1 + the skip branch's gradient in its epilogue (+ the stem's activation gradient)
            gx = _lib.conv(g1, wp1b, bwd_kinds('3x3', x.shape[0], w1, x.shape[2], x.shape[3])[1], w1.shape[1], x.shape[2], x.shape[3],
                           None, None, None, 0, res=gx_b, mask_ref=x if ctx.stem else None, mask_gain=_SQRT2)
        if ctx.stem:
            gpre, gx = gx, None                               # gradient w.r.t. the stem's pre-activation
            M = w0.shape[0]
            if need[14] or need[15]:                          # dW0 [M,3] and db0 [M] from one pass over gpre
                sl = _lib.rgb_wgrad_sum_slabs(img, gpre).sum(dim=(0, 1))          # [4, M]
                if need[14]:
                    gw0 = sl[:3].t().reshape(w0.shape)
                    gw0 = gw0 * s0 if s0 != 1.0 else gw0
                if need[15]:
                    gb0 = sl[3].contiguous()
            if need[0]:
                gx = _lib.rgb_fwd(gpre, w0.reshape(M, 3).t().contiguous(), None, None, s0)
        return (gx, gw1, gb1, gw2, gb2, gws) + (None,) * 8 + (gw0, gb0, None)


def resblock(x, w1, b1, w2, b2, ws, k_main, k_skip, s1, s2, ss, pad_main, pad_skip, gain, stem=None):
    """stem = (w0 [C,3,1,1], b0 [C], s0): x is the 3-channel image and the from-RGB layer runs inside the node"""
    w0, b0, s0 = stem if stem is not None else (None, None, 1.0)
    x = x.contiguous()        # out here (a differentiable op): the node saves its input, and a copy made inside it would have no history
    return _ResBlock.apply(x, w1, b1, w2, b2, ws, k_main, k_skip, float(s1), float(s2), float(ss), tuple(pad_main),
                           tuple(pad_skip), float(gain), w0, b0, float(s0))
