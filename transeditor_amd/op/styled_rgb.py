"""A plain StyledConv and the ToRGB that reads its output as ONE autograd node (reference: StyledConv.forward,
model_spatial_query.py:395-403, ModulatedConv2d.forward :296-337, ToRGB.forward :416-425; Generator.forward :702-714).

    a   = lrelu( d * conv3x3(s * x, scale * w) + b ) * sqrt(2)        d = demodulation of (w, s)
    rgb = conv1x1(sr * a, scale_r * wr) + br                          (the caller adds the up-sampled skip image)

Why one node: `a` has two consumers (the next layer and ToRGB), so with separate nodes the backward pays te_rgb_dgrad_f32
(writes an activation-sized tensor), the framework's gradient-accumulation add (reads two, writes one) and then the
activation-gradient pass of the convolution (reads two, writes one): 28 bytes per activation element.  Here the ToRGB data
gradient is three multiply-adds per element inside the activation-gradient pass (te_bias_act_bwd_rgb_f32): 12 bytes per
element, and the last layer - whose activation feeds ToRGB only - reads no upstream gradient at all.

Everything else is the fused modulated-convolution backward of op/modconv.py (one data-gradient convolution, one
correlation pass, the slab reducer, the demodulation chain).  A recorded backward (create_graph) differentiates the
any-order composite of the same two layers; under `second_order()` the model does not build this node.
"""
import torch
from torch.autograd import Function

from .. import _lib
from .modconv import _DEFAULT_GRAPH, _composite, _dgrad_raw, _fwd_raw, _wgrad_raw, cache_of, keep_cache

_SQRT2 = 2 ** 0.5


def supported(x, w, wr):
    """3x3 / pad 1 modulated convolution into a ToRGB the streaming kernels cover"""
    HW = x.shape[2] * x.shape[3]
    return (x.is_cuda and x.dtype == torch.float32 and w.shape[2] == 3 and wr.shape[0] == 3 and wr.shape[1] == w.shape[0]
            and x.shape[0] <= 64 and w.shape[1] <= 8192 and _lib.rgb_supported(3, w.shape[0], HW))


class _ModConvRGB(Function):
    @staticmethod
    def forward(ctx, x, w, s, bias, wr, sr, br, wscale, eps, wscale_r):
        w3 = w.reshape(w.shape[0], w.shape[1], -1)
        d, wsq = _lib.demod_fwd(w3, s, wscale, eps)
        keep_cache(ctx)
        if ctx.needs_input_grad[0]:
            a, wp_bwd = _fwd_raw(x, w, '3x3', s, d, bias, 3, wscale, with_bwd_pack=True)
        else:
            a, wp_bwd = _fwd_raw(x, w, '3x3', s, d, bias, 3, wscale), None
        rgb = _lib.rgb_fwd(a, wr.reshape(3, wr.shape[1]), sr, br, wscale_r)
        ctx.save_for_backward(x, w, s, bias, wr, sr, br, d, wsq, a)
        ctx.wp_bwd = wp_bwd
        ctx.cfg = (wscale, eps, wscale_r)
        ctx.set_materialize_grads(False)        # an unused output must not cost a zero-filled activation-sized gradient
        return a, rgb

    @staticmethod
    def backward(ctx, ga, grgb):
        with cache_of(ctx):
            return _ModConvRGB._backward(ctx, ga, grgb)

    @staticmethod
    def _backward(ctx, ga, grgb):
        x, w, s, bias, wr, sr, br, d, wsq, a = ctx.saved_tensors
        wscale, eps, wscale_r = ctx.cfg
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():
            # recorded backward: the composite of the two layers w.r.t. fresh aliases (partial derivatives only)
            with torch.enable_grad():
                from .style import demod as _demod
                al = [t.view_as(t) for t in (x, w, s, bias, wr, sr, br)]
                ac = _composite(al[0], al[1], al[2], _demod(al[1], al[2], wscale, eps), al[3], True, '3x3', wscale)
                rc = _composite(ac, al[4], al[5], None, al[6], False, '1x1', wscale_r)
                outs, gouts = [], []
                if ga is not None:
                    outs.append(ac)
                    gouts.append(ga)
                if grgb is not None:
                    outs.append(rc)
                    gouts.append(grgb)
                nd = list(need[:7])
                if _DEFAULT_GRAPH.skip_w:
                    nd[1] = nd[4] = False
                ins = [t for t, n in zip(al, nd) if n]
                gs = iter(torch.autograd.grad(outs, ins, gouts, create_graph=True, allow_unused=True))
            return tuple(next(gs) if n else None for n in nd) + (None, None, None)
        Co = w.shape[0]
        gwr = gsr = gbr = None
        if grgb is not None:
            grgb = grgb.contiguous()
            if need[6]:
                gbr = grgb.sum(dim=(0, 2, 3))
            if need[4] or need[5]:
                sl = _lib.rgb_wgrad_slabs(grgb, a)
                gwr, gsr, _ = _lib.wgrad_reduce(sl, wr.reshape(3, Co, 1), wscale_r, sr, None, want_w=need[4], want_isc=need[5])
                if gwr is not None:
                    gwr = gwr.reshape(wr.shape)
        if ga is None and grgb is None:
            return (None,) * 10
        # activation gradient (+ bias gradient) with the ToRGB data gradient folded in
        if grgb is not None and _lib.bias_act_bwd_rgb_supported(a.shape):
            g, gb = _lib.bias_act_bwd_rgb(ga, a, grgb, wr.reshape(3, Co), sr, wscale_r, 0.2, _SQRT2, want_bias=need[3])
        else:
            gt = ga
            if grgb is not None:
                gr = _lib.rgb_dgrad(grgb, wr.reshape(3, Co), sr, Co, wscale_r)
                gt = gr if gt is None else gt + gr
            g, gb = _lib.bias_act_bwd(gt, a, 0.2, _SQRT2, want_bias=need[3])
        # the modulated convolution's backward: one data-gradient convolution, one correlation pass, one reducer
        gx = _dgrad_raw(g, w, '3x3', isc=d, osc=s, wscale=wscale, wp=ctx.wp_bwd) if need[0] else None
        gw = gs = None
        if need[1] or need[2]:
            slabs = _wgrad_raw(g, x, '3x3')
            w3 = w.reshape(Co, w.shape[1], -1)
            gw, gs, gd = _lib.wgrad_reduce(slabs, w3, wscale, s, d, want_w=need[1], want_isc=need[2], want_osc=True)
            _lib.demod_bwd(gd, d, w3, wsq, s, wscale, into=(gw, gs))      # chain through d = demod(w, s), accumulated in place
            if gw is not None:
                gw = gw.reshape(w.shape)
        return gx, gw, gs, gb, gwr, gsr, gbr, None, None, None


def styled_conv_rgb(x, w, s, bias, wr, sr, br, wscale, eps, wscale_r):
    """-> (a, rgb); w [Co,Ci,3,3], s [B,Ci], bias [Co]; wr [3,Co,1,1], sr [B,Co], br [3]"""
    return _ModConvRGB.apply(x, w, s.contiguous(), bias, wr, sr.contiguous(), br, float(wscale), float(eps), float(wscale_r))
