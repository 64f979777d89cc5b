"""Modulated convolution on the te_conv / te_wgrad kernels (reference: ModulatedConv2d.forward,
model_spatial_query.py:296-337).

Formulation: with ONE shared weight w (already multiplied by the
equalised-lr scale), style scale s[b,ci] and demodulation d[b,co],

    y[b,co] = d[b,co] * conv(s[b,ci] * x[b,ci], w)[co]        (+ bias, leaky-ReLU fused)

instead of the reference's B materialised weight copies + grouped convolution.

Two layers of autograd functions:

* ``conv_core`` / ``_ConvDgrad`` / ``_ConvWgrad`` — the plain bilinear convolution trio, closed
  under differentiation (each one's backward is expressed with the other two), so anything
  built on them is differentiable to any order (R1 / path-length regularisers).
* ``modconv`` (``_ModConvFused``) — the fast path: scales, bias and activation fused into the
  kernels, a hand-written backward that produces dx, dW, ds, dd from ONE data-gradient conv and
  ONE correlation pass (per-sample slabs + te_wgrad_reduce).  When its backward is itself being
  recorded (create_graph=True) it re-derives the gradients through the differentiable trio,
  so double backward stays exact.

kinds: '3x3' (stride 1, pad 1), '1x1', 'up' (3x3 transposed stride 2 -> (2H+1)x(2W+1); the blur
that follows in the reference, model_spatial_query.py:318-321, is a separate upfirdn2d), 'down' (3x3 stride 2,
pad 0 over a (2H+1)x(2W+1) input -> HxW: the discriminator's blurred downsampling conv, :744-768; it is the
adjoint kernel pair of 'up').
"""
import contextlib

import os

import torch
from torch.autograd import Function

from .. import _lib
from .chanscale import chan_scale
from .fused_act import fused_leaky_relu

# Hints for the regulariser steps (R1, path length), which differentiate twice.  Both are optional: without them the
# fused Function recomputes its forward through the differentiable pieces when its backward turns out to be recorded.
#
# Threading: forward passes may run concurrently from several Python threads (the reference's metrics wrap the generator in
# nn.DataParallel; te_hip.h promises thread-safe entry points), so `second_order` and the frozen-weight cache are
# THREAD-LOCAL.  `skip_w` cannot be: it is read inside backward functions, which the autograd engine runs on its own
# per-device thread.  It therefore lives on a small token object (`_Graph`) that every `second_order()` context creates and
# every node built inside it keeps (`ctx.graph`); `no_weight_grads()` flips the flag on the calling thread's most recent
# token (and on the process-wide default token that nodes built outside any `second_order()` context share).
import threading


class _Graph:
    __slots__ = ('skip_w',)

    def __init__(self):
        self.skip_w = False


_DEFAULT_GRAPH = _Graph()


class _Local(threading.local):
    """per-thread state with dict-style access (`_STATE['second_order']`); every thread starts from the defaults"""

    def __init__(self):
        self.second_order = False
        self.second_wrt = None               # second_order(wrt=...): where the recorded backward will stop
        self.graph = _DEFAULT_GRAPH          # token of the current / most recent second_order() context of this thread
        self.frozen_on = False
        self.frozen_cache = None
        self.pack_cache = None               # dict while a packed_weights_cache() context is open on this thread

    def __getitem__(self, k):
        if k == 'skip_w':
            return self.graph.skip_w or _DEFAULT_GRAPH.skip_w
        return getattr(self, k)

    def __eq__(self, other):                 # (tests compare against a plain dict)
        return {'second_order': self.second_order, 'skip_w': self['skip_w']} == other


_STATE = _Local()


def current_graph():
    """token the nodes created right now belong to (see the threading note above)"""
    return _STATE.graph if _STATE.second_order else _DEFAULT_GRAPH


@contextlib.contextmanager
def second_order(wrt=None):
    """Forward passes inside this context are built from the any-order differentiable pieces right away (the caller knows
    a create_graph backward follows), which saves the forward recomputation inside the recorded backward.
    `wrt='latent'`: the recorded backward will stop at the generator's W+ latent (path-length regulariser,
    train_spatial_query.py:92-105 with `return_latents=True`), so everything that PRODUCES the latent - mapping networks,
    attention blocks, adjust_style - is differentiated once only and keeps its fused first-order nodes
    (Generator.forward suspends the hint for that section)."""
    old = (_STATE.second_order, _STATE.graph, _STATE.second_wrt)
    _STATE.second_order, _STATE.graph, _STATE.second_wrt = True, _Graph(), wrt
    try:
        yield _STATE.graph
    finally:
        _STATE.second_order = old[0]         # (the token stays current for the no_weight_grads() that follows the forward)
        _STATE.second_wrt = old[2]
        if old[0]:
            _STATE.graph = old[1]


@contextlib.contextmanager
def latent_producer_section():
    """Generator.forward wraps the part that produces the latent in this: under second_order(wrt='latent') the hint is off
    inside (see there); otherwise nothing changes."""
    if not (_STATE.second_order and _STATE.second_wrt == 'latent'):
        yield
        return
    _STATE.second_order = False
    try:
        yield
    finally:
        _STATE.second_order = True


@contextlib.contextmanager
def no_weight_grads():
    """Around `autograd.grad(..., inputs=<activations / latents>, create_graph=True)`: the first-order weight gradients
    of the convolutions are not among the requested inputs, so their correlation passes are skipped (autograd cannot tell
    a Python Function which of its outputs are consumed).  Do NOT use around a backward that accumulates into weights.
    Applies to the graph built by this thread's most recent `second_order()` forward and to nodes built outside any such
    context (those share one process-wide token: hint-less double backward is not meant to run concurrently)."""
    toks = {id(t): t for t in (_STATE.graph, _DEFAULT_GRAPH)}.values()
    old = [(t, t.skip_w) for t in toks]
    for t in toks:
        t.skip_w = True
    try:
        yield
    finally:
        for t, o in old:
            t.skip_w = o


# Frozen-weight cache (inference-only pipeline): inside `frozen_weights()` a forward that needs no gradient
# reuses the packed weight layout and the demodulation squares wsq[Co,Ci] of every layer instead of re-deriving them from
# the [Co,Ci,k,k] parameter on every call.  Opt-in because weights can be rewritten behind autograd's back (the reference's
# accumulate() goes through `.data`, train_spatial_query.py:60-61, which does not touch the version counter): whoever opens
# the context promises the weights stay put (inference.GeneratorSampler); an entry is still re-validated against the
# parameter's version counter and storage address.  Thread-local: a sampler in one thread does not switch the training
# forward of another thread to cached weights.
@contextlib.contextmanager
def frozen_weights(cache):
    """`cache`: a dict owned by the caller (lives as long as the weights it describes)."""
    old = (_STATE.frozen_on, _STATE.frozen_cache)
    _STATE.frozen_on, _STATE.frozen_cache = True, cache
    try:
        yield
    finally:
        _STATE.frozen_on, _STATE.frozen_cache = old


# Packed-weight cache for a TRAINING loop that owns every write to the weights (train_step.TrainStep): inside
# `packed_weights_cache(d)` the packed layouts of a weight tensor are kept in `d` and reused as long as the tensor's version
# counter and storage address are unchanged.  One iteration runs the discriminator three times and the generator twice
# between optimiser steps (train_spatial_query.py:173-224), the regulariser steps pack the same weight once per autograd
# node: ~45 of ~110 packing launches per iteration disappear.  Opt-in and thread-local for the same reason as the
# frozen-weight cache: a write through `.data` does not move the version counter (FusedAdam / MultiTensorEMA do bump it).
@contextlib.contextmanager
def packed_weights_cache(cache):
    old = _STATE.pack_cache
    _STATE.pack_cache = cache
    try:
        yield
    finally:
        _STATE.pack_cache = old


@contextlib.contextmanager
def cache_of(ctx):
    """Backward functions run on the autograd engine's device thread, where the forward thread's `packed_weights_cache()` is
    not visible (thread-local): a node keeps the cache it was built under (`ctx.pack_cache`, set by `keep_cache`) and re-opens
    it around its backward, so the layouts packed there - the data-gradient layout of the closed trio in the recorded R1 /
    path-length backward - are cached like the forward's.  Entries stay validated by version counter and storage address."""
    c = getattr(ctx, 'pack_cache', None)
    if c is None or _STATE.pack_cache is not None:
        yield
        return
    with packed_weights_cache(c):
        yield


def keep_cache(ctx):
    ctx.pack_cache = _STATE.pack_cache


def _pack_key(w, pack_kind, wscale):
    return (w.data_ptr(), tuple(w.shape), pack_kind, float(wscale))


def _cacheable(w):
    """only model parameters (or views of them) are cached: an intermediate tensor (the `ggw` of a double backward) may share
    address, shape and version 0 with an earlier one"""
    if _STATE.pack_cache is None:
        return None
    base = w._base if w._base is not None else w
    return base if isinstance(base, torch.nn.Parameter) else None


def refresh_packed_weights(cache):
    """Re-derive every stale layout held by `cache` in ONE launch (te_conv_pack_weights_multi_f32), into new buffers.  Called right after an
    optimiser step: the layers that use the weights next then hit the cache instead of issuing ~60 small packing launches per
    iteration.  An entry whose parameter is gone / resized is dropped; entries of untouched parameters cost nothing."""
    jobs, fresh = [], []
    for key, (ver, wp, base) in list(cache.items()):
        ptr, shape, pack_kind, wscale = key
        if base._version == ver:
            continue
        n = 1
        for d in shape:
            n *= d
        if not (base.data_ptr() == ptr and base.numel() == n and base.is_contiguous() and len(shape) == 4):
            del cache[key]                # a partial view of the parameter (or a resized one): repacked at its next use
            continue
        # a FRESH buffer per stale layout: the old one may still be held by an autograd node (ctx.wp_bwd / ctx.packs of a graph
        # that outlives the optimiser step - retain_graph, a custom step order); rewriting it in place would make that graph
        # differentiate through the NEW weights without an error
        jobs.append((torch.empty_like(wp), base.detach().view(shape), pack_kind, wscale))      # (ModulatedConv2d.weight is [1,Co,Ci,k,k]: same storage)
        fresh.append((key, base))
    _lib.conv_pack_multi(jobs)
    for (key, base), job in zip(fresh, jobs):
        cache[key] = (base._version, job[0], base)
    return len(jobs)


def packed(w, pack_kind, wscale=1.0):
    """packed layout `pack_kind` of w (through the cache when one is open)"""
    base = _cacheable(w)
    if base is None:
        return _lib.conv_pack(w, pack_kind, wscale)
    cache, key = _STATE.pack_cache, _pack_key(w, pack_kind, wscale)
    ent = cache.get(key)
    if ent is None or ent[0] != w._version or ent[2] is not base:
        ent = cache[key] = (w._version, _lib.conv_pack(w, pack_kind, wscale), base)      # (keeps the storage alive: the key stays valid)
    return ent[1]


def packed2(w, kind_a, kind_b, wscale=1.0):
    """two layouts of w; one launch when both have to be produced"""
    base = _cacheable(w)
    if base is None:
        return _lib.conv_pack2(w, kind_a, kind_b, wscale)
    cache = _STATE.pack_cache
    ka, kb = _pack_key(w, kind_a, wscale), _pack_key(w, kind_b, wscale)
    ea, eb = cache.get(ka), cache.get(kb)
    va = ea is not None and ea[0] == w._version and ea[2] is base
    vb = eb is not None and eb[0] == w._version and eb[2] is base
    if va and vb:
        return ea[1], eb[1]
    if not va and not vb:
        wa, wb = _lib.conv_pack2(w, kind_a, kind_b, wscale)
        cache[ka], cache[kb] = (w._version, wa, base), (w._version, wb, base)
        return wa, wb
    return packed(w, kind_a, wscale), packed(w, kind_b, wscale)


def _frozen_entry(w, kind, wscale, want_wsq, pack_kind=_lib.PACK_FWD):
    base = w._base if w._base is not None else w
    key = (id(base), w.data_ptr(), tuple(w.shape), kind, float(wscale), pack_kind)
    stamp = (base._version, base.data_ptr())
    ent = _STATE.frozen_cache.get(key)
    if ent is None or ent['stamp'] != stamp:
        ent = {'stamp': stamp, 'wp': _lib.conv_pack(w, pack_kind, wscale), 'wsq': None, 'keep': base}
        _STATE.frozen_cache[key] = ent
    if want_wsq and ent['wsq'] is None:
        w3 = w.reshape(w.shape[0], w.shape[1], -1)
        ent['wsq'] = (w3 * wscale).square().sum(dim=2).contiguous()
    return ent


def _modconv_frozen(x, w, isc, osc, bias, act, kind, wscale, demod_eps):
    """forward only, weights taken from the frozen cache (no autograd node is created)"""
    H, W = _lowres_hw(kind, False, x)
    pk, ck = fwd_kinds(kind, x.shape[0], w, H, W)
    ent = _frozen_entry(w, kind, wscale, demod_eps is not None, pk)
    if demod_eps is not None:
        osc = _lib.demod_from_wsq(ent['wsq'], isc, demod_eps)
    if kind == '1x1' and osc is None and not act and _lib.rgb_supported(w.shape[0], w.shape[1], x.shape[2] * x.shape[3]):
        return _lib.rgb_fwd(x, w.reshape(w.shape[0], w.shape[1]), isc, bias, wscale)
    return _lib.conv(x, ent['wp'], ck, w.shape[0], H, W, isc, osc, bias, _act_code(act))


_KIND = {'3x3': _lib.CONV_3X3, '1x1': _lib.CONV_1X1, 'up': _lib.CONV_T2, 'down': _lib.CONV_S2}


_SQRT2 = 2 ** 0.5


def _act_gain(act):
    """`act` is False, True (the reference's lrelu * sqrt(2)) or the gain itself (sqrt(2) or 1.0)."""
    return _SQRT2 if act is True else float(act)


def _act_code(act):
    if not act:
        return 0
    g = _act_gain(act)
    if abs(g - _SQRT2) < 1e-6:
        return 3
    if abs(g - 1.0) < 1e-6:
        return 4
    raise RuntimeError(f'modconv: leaky-ReLU gain {g} is not one the kernels fuse (sqrt(2) or 1)')


def _lowres_hw(kind, is_output, t):
    """low-resolution (H, W) of the problem from the conv input (is_output=False) or output (True)."""
    H, W = t.shape[2], t.shape[3]
    if (kind == 'up' and is_output) or (kind == 'down' and not is_output):
        if H % 2 == 0 or W % 2 == 0:
            raise RuntimeError(f"kind '{kind}': high-resolution side must be (2H+1)x(2W+1), got {H}x{W}")
        return (H - 1) // 2, (W - 1) // 2
    return H, W


def _bwd_pack_kind(kind):
    return _lib.PACK_SWAP if kind in ('up', 'down') else _lib.PACK_DGRAD


# 3x3 / stride 1 launches that the 1-D Winograd F(2,3) kernel covers (csrc/wino.hip: >= 32x32 images, K % 8 == 0, M % 128 == 0 - the
# launches that carry the FLOPs of both networks) use it: same result to fp32 round-off, 2/3 of the MFMAs.  The choice is a pure
# function of the problem shape, so the forward (which packs the data-gradient layout ahead) and the backward agree on it.
USE_WINOGRAD = True      # False: the direct kernel everywhere (A/B measurements)
# the Winograd form on the bf16 matrix pipe (TE_CONV_3X3W6: three-piece split, fp32-equivalent results) where it applies;
# TE_SPLIT_BF16=0 (or False here): fp32 matrix instructions everywhere (A/B measurements)
USE_SPLIT_BF16 = os.environ.get('TE_SPLIT_BF16', '1') != '0'


def fwd_kinds(kind, B, w, H, W):
    """(weight pack kind, convolution kind code) of the forward launch; H, W = low-resolution size"""
    if kind == '3x3' and USE_WINOGRAD:
        if USE_SPLIT_BF16 and (W >= 32 or USE_SPLIT_W16) and _lib.wino6_ok(B, w.shape[1], w.shape[0], H, W):
            return _lib.PACK_W6FWD, _lib.CONV_3X3W6
        if _lib.wino_ok(B, w.shape[1], w.shape[0], H, W):
            return _lib.PACK_WFWD, _lib.CONV_3X3W
    if kind == 'down' and USE_SPLIT_BF16 and USE_SPLIT_S2 and _lib.s2s6_ok(B, w.shape[1], w.shape[0], H, W):
        return _lib.PACK_S6FWD, _lib.CONV_S2S6          # the stride-2 convolution on the bf16 matrix pipe (csrc/s2s6.hip)
    if kind == 'up' and USE_SPLIT_BF16 and USE_SPLIT_T2 and _lib.t2s6_ok(B, w.shape[1], w.shape[0], H, W):
        return _lib.PACK_T6FWD, _lib.CONV_T2S6          # the transposed stride-2 convolution likewise (csrc/t2s6.hip)
    return _lib.PACK_FWD, _KIND[kind]


# TE_SPLIT_1X1=0: the 1x1 launches stay on the fp32 matrix instructions (A/B)
USE_SPLIT_1X1 = os.environ.get('TE_SPLIT_1X1', '1') != '0'


def plain_1x1_kinds(B, w, H, W, dgrad=False):
    """(pack kind, convolution kind) of a PLAIN 1x1 product - no scales, bias or activation, at most a residual: the skip branch of the
    discriminator's ResBlocks (op/resblock.py) - forward (Ci -> Co) or data gradient (Co -> Ci): the split-bf16 kernel of csrc/p1s6.hip
    (round 6) where it applies, the fp32 kernel elsewhere"""
    K, M = (w.shape[0], w.shape[1]) if dgrad else (w.shape[1], w.shape[0])
    if USE_SPLIT_BF16 and USE_SPLIT_1X1 and _lib.p1s6_ok(B, K, M, H, W):
        return (_lib.PACK_P6DGRAD if dgrad else _lib.PACK_P6FWD), _lib.CONV_1X1S6
    return (_lib.PACK_DGRAD if dgrad else _lib.PACK_FWD), _lib.CONV_1X1


# TE_SPLIT_W16=0: the 16-column 3x3 launches (two samples side by side in a tile row of the split Winograd kernel, round 6) stay on the fp32 kernel (A/B)
USE_SPLIT_W16 = os.environ.get('TE_SPLIT_W16', '1') != '0'
# TE_SPLIT_S2=0: the stride-2 launches stay on the fp32 matrix instructions while the 3x3 stride-1 ones keep the split form (A/B)
USE_SPLIT_S2 = os.environ.get('TE_SPLIT_S2', '1') != '0'
USE_SPLIT_T2 = os.environ.get('TE_SPLIT_T2', '1') != '0'


def bwd_kinds(kind, B, w, H, W):
    """the same for the data gradient (a convolution from Co to Ci channels)"""
    if kind == '3x3' and USE_WINOGRAD:
        if USE_SPLIT_BF16 and (W >= 32 or USE_SPLIT_W16) and _lib.wino6_ok(B, w.shape[0], w.shape[1], H, W):
            return _lib.PACK_W6DGRAD, _lib.CONV_3X3W6
        if _lib.wino_ok(B, w.shape[0], w.shape[1], H, W):
            return _lib.PACK_WDGRAD, _lib.CONV_3X3W
    if kind == 'up' and USE_SPLIT_BF16 and USE_SPLIT_S2 and _lib.s2s6_ok(B, w.shape[0], w.shape[1], H, W):
        return _lib.PACK_S6SWAP, _lib.CONV_S2S6         # adjoint of the transposed kind = the strided one, from Co to Ci channels
    if kind == 'down' and USE_SPLIT_BF16 and USE_SPLIT_T2 and _lib.t2s6_ok(B, w.shape[0], w.shape[1], H, W):
        return _lib.PACK_T6SWAP, _lib.CONV_T2S6         # adjoint of the strided kind = the transposed one, from Co to Ci channels
    ck = {'up': _lib.CONV_S2, 'down': _lib.CONV_T2}.get(kind)          # adjoint of the transposed / strided kind
    return _bwd_pack_kind(kind), (ck if ck is not None else _KIND[kind])


def _fwd_raw(x, w, kind, isc=None, osc=None, bias=None, act=0, wscale=1.0, with_bwd_pack=False):
    """with_bwd_pack: also return the data-gradient packing of w (one launch packs both layouts)."""
    H, W = _lowres_hw(kind, False, x)
    pk, ck = fwd_kinds(kind, x.shape[0], w, H, W)
    if with_bwd_pack:
        wp, wpb = packed2(w, pk, bwd_kinds(kind, x.shape[0], w, H, W)[0], wscale)
        return _lib.conv(x, wp, ck, w.shape[0], H, W, isc, osc, bias, act), wpb
    if _STATE.pack_cache is not None and torch.is_grad_enabled() is False:
        # (a no-grad forward inside a training loop: the same weights meet a backward later in the iteration)
        wp, _ = packed2(w, pk, bwd_kinds(kind, x.shape[0], w, H, W)[0], wscale)
    else:
        wp = packed(w, pk, wscale)
    return _lib.conv(x, wp, ck, w.shape[0], H, W, isc, osc, bias, act)


def _dgrad_raw(g, w, kind, isc=None, osc=None, wscale=1.0, wp=None):
    """data gradient: g is shaped like the conv OUTPUT; returns a tensor shaped like the conv input.
    isc scales the channels of g ([B,Co]), osc the channels of the result ([B,Ci]).  wp: w already packed for it."""
    H, W = _lowres_hw(kind, True, g)
    pk, ck = bwd_kinds(kind, g.shape[0], w, H, W)          # ('up': the strided kind is its adjoint; 'down': the transposed one)
    if wp is None:
        wp = packed(w, pk, wscale)
    return _lib.conv(g, wp, ck, w.shape[1], H, W, isc, osc)


def _wgrad_raw(g, x, kind, group=False):
    """per-sample correlation slabs [B, S, Co, Ci, taps] ('down': [B, S, Ci, Co, taps], see _slab_sum).  group=True: the caller
    only sums them (plain gradient), so several samples may share a slab (_lib.wgrad_slabs)."""
    if kind == 'down':     # same correlation as 'up' with the roles of the two tensors swapped
        H, W = g.shape[2], g.shape[3]
        return _lib.wgrad_slabs(x, g, _lib.CONV_T2, H, W, group)
    H, W = x.shape[2], x.shape[3]
    return _lib.wgrad_slabs(g, x, _KIND[kind], H, W, group)


def _slab_sum(slabs, kind):
    """plain (unmodulated) weight gradient [Co, Ci, taps] from the slabs"""
    gw = slabs.sum(dim=(0, 1))
    return gw.transpose(0, 1) if kind == 'down' else gw


def _wgrad_plain(gy, x, kind, ksize, wscale):
    """wscale * sum of the correlation slabs -> [Co, Ci, k, k] (te_wgrad_reduce_f32 without modulation; 'down' slabs come
    transposed and go through the framework's reduction)"""
    slabs = _wgrad_raw(gy, x, kind, group=True)
    Co, Ci = gy.shape[1], x.shape[1]
    if kind == 'down':
        gw = _slab_sum(slabs, kind)
        gw = gw * wscale if wscale != 1.0 else gw
    else:
        # (the reducer reads a weight operand only for the style / demodulation gradients, which are not asked for here)
        gw, _, _ = _lib.wgrad_reduce(slabs, slabs[0, 0], wscale, None, None, want_w=True)
    return gw.reshape(Co, Ci, ksize, ksize)


# The plain convolution y = conv(x, wscale * w) and its two gradients: a trio closed under differentiation (each one's
# backward is the other two, with the same constant wscale), so anything built on it differentiates to any order.
class _ConvFwd(Function):
    @staticmethod
    def forward(ctx, x, w, kind, wscale):
        ctx.save_for_backward(x, w)
        ctx.kind, ctx.wscale = kind, wscale
        ctx.graph = current_graph()
        keep_cache(ctx)
        return _fwd_raw(x, w, kind, wscale=wscale)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        skip_w = ctx.graph.skip_w or _DEFAULT_GRAPH.skip_w
        with cache_of(ctx):
            gx = _ConvDgrad.apply(gy, w, ctx.kind, ctx.wscale) if ctx.needs_input_grad[0] else None
            gw = _ConvWgrad.apply(gy, x, ctx.kind, w.shape[2], ctx.wscale) if (ctx.needs_input_grad[1] and not skip_w) else None
        return gx, gw, None, None


class _ConvDgrad(Function):
    @staticmethod
    def forward(ctx, gy, w, kind, wscale):
        ctx.save_for_backward(gy, w)
        ctx.kind, ctx.wscale = kind, wscale
        keep_cache(ctx)
        return _dgrad_raw(gy, w, kind, wscale=wscale)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        with cache_of(ctx):
            g_gy = _ConvFwd.apply(ggx, w, ctx.kind, ctx.wscale) if ctx.needs_input_grad[0] else None
            g_w = _ConvWgrad.apply(gy, ggx, ctx.kind, w.shape[2], ctx.wscale) if ctx.needs_input_grad[1] else None
        return g_gy, g_w, None, None


class _ConvWgrad(Function):
    @staticmethod
    def forward(ctx, gy, x, kind, ksize, wscale):
        ctx.save_for_backward(gy, x)
        ctx.kind, ctx.wscale = kind, wscale
        keep_cache(ctx)
        return _wgrad_plain(gy, x, kind, ksize, wscale)

    @staticmethod
    def backward(ctx, ggw):
        gy, x = ctx.saved_tensors
        with cache_of(ctx):
            g_gy = _ConvFwd.apply(x, ggw, ctx.kind, ctx.wscale) if ctx.needs_input_grad[0] else None
            g_x = _ConvDgrad.apply(gy, ggw, ctx.kind, ctx.wscale) if ctx.needs_input_grad[1] else None
        return g_gy, g_x, None, None, None


def set_split_bf16(on):
    """ONE switch for every split-bf16 kernel: the convolution kinds (gated here, by USE_SPLIT_BF16) and the weight-gradient kernels
    (gated in the library, te_wgrad_split_bf16).  Assigning `modconv.USE_SPLIT_BF16 = x` does the same (module property below), so an
    A/B that flips the Python flag can no longer leave the weight gradient on the other arithmetic.  TE_SPLIT_WGRAD=0 in the
    environment keeps the weight gradient alone on the fp32 kernels."""
    on = bool(on)
    globals()['USE_SPLIT_BF16'] = on
    _lib.wgrad_split(1 if (on and os.environ.get('TE_SPLIT_WGRAD', '1') != '0') else 0)
    return on


def conv_core(x, w, kind='3x3', wscale=1.0):
    """Plain convolution y = conv(x, wscale * w) (w [Co,Ci,k,k]), differentiable to any order."""
    return _ConvFwd.apply(x, w, kind, float(wscale))


# The MODULATED convolution y = d (.) conv(s (.) x, wscale w) as a family closed under differentiation.  It is one
# five-linear form  P(gy, x, w, s, d) = sum gy[b,co,p (+) t] x[b,ci,p] w[co,ci,t] s[b,ci] d[b,co]  seen from its five
# arguments: y = dP/dgy, the data gradient dP/dx, the weight gradient dP/dw, and the two scale gradients
#     dP/ds[b,ci] = sum_p x * (dP/dx) / s ,      dP/dd[b,co] = sum_p gy * (dP/dgy) / d
# (the sums are te_chan_dot_f32 passes, the divisions [B,C]-sized).  The derivative of dP/da with respect to b, contracted with
# a cotangent c, is dP/db with a := c - so the three Functions below name each other in their backwards and anything built on
# them differentiates to any order.  Unlike the chan_scale -> conv_core -> chan_scale composite of rounds 2-3, the style scale
# and the demodulation ride INSIDE the convolution kernels (staging / epilogue, the arguments the fused first-order path has
# always used): per layer and order, three activation-sized elementwise passes and their autograd accumulations disappear
# (path-length step: 161 chan_scale launches and ~5 ms per step).
_SCALE_FLOOR = 1e-20


def _nonzero(s):
    """The scale gradients of the closed family divide by the scale (dP/ds = sum_p x * (s U) / s with s U = the data gradient the
    kernel returns).  Scales of magnitude below 1e-20 - exact zeros included - are moved to +-1e-20 BEFORE they enter the family
    (kernels and divisions see the same value): the product s U then stays a normal number for every |U| > 1e-18, so the quotient is
    sum_p x U to fp32 accuracy instead of 0 / garbage (the former guard, 1.2e-38, let s U flush to zero), and the outputs move by less
    than 1e-20 |x w| - far below half an ulp of anything they are added to.  Both scales are guarded (the demodulation coefficient
    is rsqrt(... + 1e-8) > 0 in the model, but modconv_closed takes any osc).  What the division form cannot remove: under
    create_graph, d(dP/ds)/ds is two terms of size |dP/ds| / |s| that cancel analytically; their round-off, ~6e-8 |dP/ds| / |s|, is
    what a channel with a small style scale adds to a second-order gradient (pinned second-order parity: tests/test_gpu_timed_second_order.py)."""
    if s is None:
        return None
    return torch.where(s.abs() < _SCALE_FLOOR, torch.where(s < 0, -_SCALE_FLOOR, _SCALE_FLOOR).to(s.dtype), s)


def _rgb_ok(w, d, x_like, kind):
    return (kind == '1x1' and d is None and w.shape[0] == 3
            and _lib.rgb_supported(3, w.shape[1], x_like.shape[2] * x_like.shape[3]))


def _chan_dot(a, b):
    from .chanscale import _ChanDot
    return _ChanDot.apply(a.contiguous(), b.contiguous())


class _MCFwd(Function):
    @staticmethod
    def forward(ctx, x, w, s, d, kind, wscale):
        ctx.kind, ctx.wscale = kind, wscale
        ctx.graph = current_graph()
        keep_cache(ctx)
        if _rgb_ok(w, d, x, kind):          # ToRGB: HBM-bound streaming kernel
            y = _lib.rgb_fwd(x, w.reshape(3, w.shape[1]), s.contiguous(), None, wscale)
        else:
            y = _fwd_raw(x, w, kind, s.contiguous(), None if d is None else d.contiguous(), None, 0, wscale)
        ctx.save_for_backward(x, w, s, d, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, s, d, y = ctx.saved_tensors
        need = ctx.needs_input_grad
        skip_w = ctx.graph.skip_w or _DEFAULT_GRAPH.skip_w
        with cache_of(ctx):
            gx = _MCDgrad.apply(gy, w, s, d, ctx.kind, ctx.wscale) if (need[0] or need[2]) else None
            gw = _MCWgrad.apply(gy, x, s, d, ctx.kind, w.shape[2], ctx.wscale) if (need[1] and not skip_w) else None
        gs = _chan_dot(gx, x) / s if need[2] else None
        gd = _chan_dot(gy, y) / d if (d is not None and need[3]) else None
        return (gx if need[0] else None), gw, gs, gd, None, None


class _MCDgrad(Function):
    @staticmethod
    def forward(ctx, gy, w, s, d, kind, wscale):       # s (.) dgrad(d (.) gy, wscale w), shaped like the convolution's input
        ctx.kind, ctx.wscale = kind, wscale
        keep_cache(ctx)
        if _rgb_ok(w, d, gy, kind):
            gx = _lib.rgb_dgrad(gy, w.reshape(3, w.shape[1]), s.contiguous(), w.shape[1], wscale)
        else:
            gx = _dgrad_raw(gy, w, kind, isc=None if d is None else d.contiguous(), osc=s.contiguous(), wscale=wscale)
        ctx.save_for_backward(gy, w, s, d, gx)
        return gx

    @staticmethod
    def backward(ctx, ggx):
        gy, w, s, d, gx = ctx.saved_tensors
        need = ctx.needs_input_grad
        with cache_of(ctx):
            y2 = _MCFwd.apply(ggx, w, s, d, ctx.kind, ctx.wscale) if (need[0] or (d is not None and need[3])) else None
            gw = _MCWgrad.apply(gy, ggx, s, d, ctx.kind, w.shape[2], ctx.wscale) if need[1] else None
        gs = _chan_dot(ggx, gx) / s if need[2] else None
        gd = _chan_dot(gy, y2) / d if (d is not None and need[3]) else None
        return (y2 if need[0] else None), gw, gs, gd, None, None


class _MCWgrad(Function):
    @staticmethod
    def forward(ctx, gy, x, s, d, kind, ksize, wscale):    # wscale * sum_b d_b s_b (correlation slab of sample b)  [Co,Ci,k,k]
        ctx.kind, ctx.wscale = kind, wscale
        keep_cache(ctx)
        ctx.save_for_backward(gy, x, s, d)
        Co, Ci = gy.shape[1], x.shape[1]
        gyc, xc = gy.contiguous(), x.contiguous()
        if kind == '1x1' and d is None and Co == 3 and _lib.rgb_supported(3, Ci, x.shape[2] * x.shape[3]):
            slabs = _lib.rgb_wgrad_slabs(gyc, xc)
        else:
            slabs = _wgrad_raw(gyc, xc, kind)
        gw, _, _ = _lib.wgrad_reduce(slabs, slabs[0, 0], wscale, s.contiguous(), None if d is None else d.contiguous(), want_w=True)
        return gw.reshape(Co, Ci, ksize, ksize)

    @staticmethod
    def backward(ctx, ggw):
        gy, x, s, d = ctx.saved_tensors
        need = ctx.needs_input_grad
        with cache_of(ctx):
            y2 = _MCFwd.apply(x, ggw, s, d, ctx.kind, ctx.wscale) if (need[0] or (d is not None and need[3])) else None
            gx = _MCDgrad.apply(gy, ggw, s, d, ctx.kind, ctx.wscale) if (need[1] or need[2]) else None
        gs = _chan_dot(gx, x) / s if need[2] else None
        gd = _chan_dot(gy, y2) / d if (d is not None and need[3]) else None
        return (y2 if need[0] else None), (gx if need[1] else None), gs, gd, None, None, None


def _closed_ok(x):
    return x.is_cuda and x.dtype == torch.float32


USE_CLOSED_MODCONV = True      # False: the chan_scale -> conv_core -> chan_scale composite of rounds 2-3 (A/B measurements)


def modconv_closed(x, w, isc, osc, kind='3x3', wscale=1.0):
    """osc (.) conv(isc (.) x, wscale * w), differentiable to any order with the scales inside the convolution kernels"""
    return _MCFwd.apply(x, w, _nonzero(isc), _nonzero(osc), kind, float(wscale))


def _composite(x, w, isc, osc, bias, act, kind, wscale=1.0):
    """The same function as the fused kernel, built from any-order differentiable pieces (the equalised-lr constant rides
    in the weight packing of the trio: no `w * wscale` pass, no scaling pass for its gradient)."""
    if USE_CLOSED_MODCONV and isc is not None and kind in ('3x3', 'up', '1x1') and _closed_ok(x):
        y = modconv_closed(x, w, isc, osc, kind, wscale)
        if act:
            return fused_leaky_relu(y, bias, 0.2, _act_gain(act))
        return y if bias is None else y + bias[None, :, None, None]
    if isc is not None:
        x = chan_scale(x, isc)
    y = conv_core(x, w, kind, wscale)
    if osc is not None:
        y = chan_scale(y, osc)
    if act:
        return fused_leaky_relu(y, bias, 0.2, _act_gain(act))
    if bias is not None:
        y = y + bias[None, :, None, None]
    return y


class _ModConvFused(Function):
    @staticmethod
    def forward(ctx, x, w, isc, osc, bias, act, kind, wscale, demod_eps):
        # demod_eps != None: osc is the demodulation coefficient of (w, isc), computed here (te_demod_fwd_f32) and
        # differentiated here (its dW / d style are accumulated into the convolution's own in the backward)
        ctx.demod = None
        keep_cache(ctx)
        if demod_eps is not None:
            w3 = w.reshape(w.shape[0], w.shape[1], -1)
            osc, wsq = _lib.demod_fwd(w3, isc, wscale, demod_eps)
            ctx.demod = (demod_eps, wsq)
        # ToRGB (1x1 to 3 channels, no demodulation / activation) is HBM-bound: dedicated streaming kernels
        ctx.rgb = (kind == '1x1' and osc is None and not act
                   and _lib.rgb_supported(w.shape[0], w.shape[1], x.shape[2] * x.shape[3]))
        if ctx.rgb:
            out = _lib.rgb_fwd(x, w.reshape(w.shape[0], w.shape[1]), isc, bias, wscale)
            ctx.save_for_backward(x, w, isc, osc, bias, None)
            ctx.act, ctx.kind, ctx.wscale = act, kind, wscale
            return out
        # from-RGB stem (1x1 FROM 3 channels, unmodulated: the discriminator's first layer, :815): write-bound, the same
        # streaming kernels with the operands' roles exchanged instead of a 3-of-16-channels MFMA stage
        ctx.stem = (kind == '1x1' and w.shape[1] == 3 and isc is None and osc is None
                    and _lib.rgb_supported(3, w.shape[0], x.shape[2] * x.shape[3]))
        if ctx.stem:
            out = _lib.rgb_expand(x, w.reshape(w.shape[0], 3).t(), bias, _act_code(act), wscale)
            ctx.save_for_backward(x, w, isc, osc, bias, out if act else None)
            ctx.act, ctx.kind, ctx.wscale = act, kind, wscale
            return out
        ctx.wp_bwd = None
        if ctx.needs_input_grad[0]:         # a backward will want dx: pack the data-gradient layout in the same launch
            out, ctx.wp_bwd = _fwd_raw(x, w, kind, isc, osc, bias, _act_code(act), wscale, with_bwd_pack=True)
        else:
            out = _fwd_raw(x, w, kind, isc, osc, bias, _act_code(act), wscale)
        ctx.save_for_backward(x, w, isc, osc, bias, out if act else None)
        ctx.act, ctx.kind, ctx.wscale = act, kind, wscale
        return out

    @staticmethod
    def backward(ctx, g):
        with cache_of(ctx):
            return _ModConvFused._backward(ctx, g)

    @staticmethod
    def _backward(ctx, g):
        x, w, isc, osc, bias, out = ctx.saved_tensors
        act, kind, wscale = ctx.act, ctx.kind, ctx.wscale
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():
            # double backward requested: differentiate the composite built from the closed trio
            # The inputs may depend on each other in the OUTER graph (osc = demod is a function of isc = style
            # scale): differentiate w.r.t. fresh aliases so each returned gradient is the PARTIAL derivative,
            # while everything stays connected to the original tensors for the next differentiation.
            with torch.enable_grad():
                al = [None if t is None else t.view_as(t) for t in (x, w, isc, osc, bias)]
                if ctx.demod is not None:          # osc is a function of (w, isc) inside this node
                    from .style import demod as _demod
                    al[3] = _demod(al[1], al[2], wscale, ctx.demod[0])
                    need = tuple(need[:3]) + (False,) + tuple(need[4:])
                y = _composite(al[0], al[1], al[2], al[3], al[4], act, kind, wscale)
                if _DEFAULT_GRAPH.skip_w:        # (a node of the fused path was built outside second_order(): default token)
                    need = (need[0], False) + tuple(need[2:])
                ins = [t for t, n in zip(al, need[:5]) if n and t is not None]
                gs = iter(torch.autograd.grad(y, ins, g, create_graph=True, allow_unused=True))
            return tuple(next(gs) if (n and t is not None) else None
                         for t, n in zip(al, need[:5])) + (None, None, None, None)
        g = g.contiguous()
        g_bias = None
        if act:
            g, g_bias = _lib.bias_act_bwd(g, out, 0.2, _act_gain(act), want_bias=bias is not None)
        elif bias is not None and need[4]:
            g_bias = g.sum(dim=(0, 2, 3))
        if getattr(ctx, 'stem', False):
            gx = _lib.rgb_fwd(g, w.reshape(w.shape[0], 3).t().contiguous(), None, None, wscale) if need[0] else None
            gw = None
            if need[1]:
                gw = _lib.rgb_wgrad_slabs(x, g).sum(dim=(0, 1)).reshape(3, w.shape[0]).t().reshape(w.shape)
                gw = gw * wscale if wscale != 1.0 else gw
            return gx, gw, None, None, g_bias, None, None, None, None
        if ctx.rgb:
            gx = _lib.rgb_dgrad(g, w.reshape(w.shape[0], w.shape[1]), isc, x.shape[1], wscale) if need[0] else None
        else:
            gx = _dgrad_raw(g, w, kind, isc=osc, osc=isc, wscale=wscale, wp=ctx.wp_bwd) if need[0] else None
        gw = gisc = gosc = None
        if need[1] or (need[2] and isc is not None) or (need[3] and osc is not None):
            slabs = _lib.rgb_wgrad_slabs(g, x) if ctx.rgb else _wgrad_raw(g, x, kind)
            if kind == 'down' and isc is None and osc is None:        # the discriminator's convolutions
                gw = _slab_sum(slabs, kind).reshape(w.shape) if need[1] else None
                if gw is not None and wscale != 1.0:
                    gw = gw * wscale
                return gx, gw, None, None, g_bias, None, None, None, None
            if kind == 'down':       # modulated (ModulatedConv2d(downsample=True)): slabs come as [B,S,Ci,Co,taps]
                slabs = slabs.sum(dim=1).transpose(1, 2).contiguous().unsqueeze(1)
            w3 = w.reshape(w.shape[0], w.shape[1], -1)
            dm = ctx.demod is not None
            gw, gisc, gosc = _lib.wgrad_reduce(slabs, w3, wscale, isc, osc,
                                               want_w=need[1], want_isc=need[2] and isc is not None,
                                               want_osc=(need[3] or dm) and osc is not None)
            if dm:       # chain through d = demod(w, isc): its dW / d style are added to gw / gisc in place
                if gw is not None or gisc is not None:
                    _lib.demod_bwd(gosc, osc, w3, ctx.demod[1], isc, wscale, into=(gw, gisc))
                gosc = None
            if gw is not None:
                gw = gw.reshape(w.shape)
        return gx, gw, gisc, gosc, g_bias, None, None, None, None


def modconv(x, w, isc=None, osc=None, bias=None, act=False, kind='3x3', wscale=1.0, demod_eps=None):
    """out = [lrelu*sqrt2]( osc[b,co] * conv(isc[b,ci] * x, wscale * w) + bias[co] )  — fused kernels.
    `act`: False, True (gain sqrt(2), the reference's FusedLeakyReLU) or the gain itself (sqrt(2) or 1.0).
    `demod_eps` (instead of `osc`): osc = rsqrt(sum (wscale w isc)^2 + eps), the StyleGAN2 demodulation
    (model_spatial_query.py:300-304), evaluated and differentiated inside the same autograd node.
    `wscale` is the equalised-lr constant: the parameter is consumed as stored, its gradient comes back scaled."""
    if demod_eps is not None and (osc is not None or isc is None or isc.shape[0] > 64 or w.shape[1] > 8192):
        if osc is not None or isc is None:
            raise RuntimeError('modconv: demod_eps needs isc and excludes an explicit osc')
        from .style import demod as _demod
        osc, demod_eps = _demod(w, isc, float(wscale), demod_eps), None      # shapes the demod kernels do not cover
    if _STATE.second_order and torch.is_grad_enabled():
        if demod_eps is not None:
            from .style import demod as _demod
            osc = _demod(w, isc, float(wscale), demod_eps)
        return _composite(x, w, isc, osc, bias, act, kind, float(wscale))
    isc = isc.contiguous() if isc is not None else None
    osc = osc.contiguous() if osc is not None else None
    if _STATE.frozen_on and not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, w, isc, osc, bias))):
        return _modconv_frozen(x, w, isc, osc, bias, act, kind, float(wscale), demod_eps)
    return _ModConvFused.apply(x, w, isc, osc, bias, act, kind, float(wscale), demod_eps)


# `modconv.USE_SPLIT_BF16 = flag` goes through set_split_bf16 (reads stay plain global lookups)
import sys as _sys
import types as _types


class _ModconvModule(_types.ModuleType):
    @property
    def USE_SPLIT_BF16(self):
        return self.__dict__['USE_SPLIT_BF16']

    @USE_SPLIT_BF16.setter
    def USE_SPLIT_BF16(self, on):
        set_split_bf16(on)


_sys.modules[__name__].__class__ = _ModconvModule
