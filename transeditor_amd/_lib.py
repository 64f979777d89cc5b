"""ctypes binding of libte_hip.so (C ABI: include/te_hip.h) + thin tensor-level wrappers.

The product path has NO fallback: if the shared object is missing, was built for another
architecture, or a tensor is not a contiguous fp32 CUDA(HIP) tensor, these wrappers raise.
PyTorch is used here only for device memory and the current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libte_hip.so')
_lib = None

CONV_3X3, CONV_T2, CONV_S2, CONV_1X1, CONV_3X3W, CONV_3X3W6, CONV_S2S6, CONV_T2S6, CONV_1X1S6 = 0, 1, 2, 3, 4, 5, 6, 7, 8
(PACK_FWD, PACK_DGRAD, PACK_SWAP, PACK_WFWD, PACK_WDGRAD, PACK_W6FWD, PACK_W6DGRAD, PACK_S6FWD, PACK_S6SWAP, PACK_T6FWD,
 PACK_T6SWAP, PACK_P6FWD, PACK_P6DGRAD) = range(13)

_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGNATURES = {
    'te_version': (C.c_int, []),
    'te_last_error_string': (C.c_char_p, []),
    'te_arch': (C.c_char_p, []),
    'te_bias_act_f32': (C.c_int, [_P, _P, _P, _P, _I, _I, _F, _F, _L, _L, _L, _P]),
    'te_bias_act_f16': (C.c_int, [_P, _P, _P, _P, _I, _I, _F, _F, _L, _L, _L, _P]),
    'te_bias_act_f64': (C.c_int, [_P, _P, _P, _P, _I, _I, _F, _F, _L, _L, _L, _P]),
    'te_bias_act_bwd_ws_floats': (C.c_int64, [_L, _L, _L]),
    'te_bias_act_bwd_f32': (C.c_int, [_P, _P, _P, _P, _P, _F, _F, _L, _L, _L, _P]),
    'te_bias_act_bwd_rgb_supported': (C.c_int, [_L, _L, _L]),
    'te_bias_act_bwd_rgb_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _L, _L, _L, _P]),
    'te_upfirdn2d_f32': (C.c_int, [_P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                                   _P, _L, _I, _F, _F, _P]),
    'te_upfirdn2d_f16': (C.c_int, [_P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'te_upfirdn2d_f64': (C.c_int, [_P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'te_conv_packed_numel': (C.c_int64, [_I, _I, _I, _I]),
    'te_conv_pack_weights_f32': (C.c_int, [_P, _P, _F, _I, _I, _I, _I, _P]),
    'te_conv_pack_weights2_f32': (C.c_int, [_P, _I, _P, _I, _P, _F, _I, _I, _I, _P]),
    'te_conv_pack_weights_multi_f32': (C.c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P]),
    'te_conv_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'te_conv_splitk_count': (C.c_int, [_I, _I, _I, _I, _I, _I]),
    'te_conv_wino_supported': (C.c_int, [_I, _I, _I, _I, _I]),
    'te_conv_wino6_supported': (C.c_int, [_I, _I, _I, _I, _I]),
    'te_conv_wino6_form': (C.c_int, [_I]),
    'te_conv_s2s6_form': (C.c_int, [_I]),
    'te_conv_t2s6_form': (C.c_int, [_I]),
    'te_conv_t2s6_ws_floats': (C.c_int64, [_I, _I, _I]),
    'te_conv_s2s6_supported': (C.c_int, [_I, _I, _I, _I, _I]),
    'te_conv_t2s6_supported': (C.c_int, [_I, _I, _I, _I, _I]),
    'te_conv_p1s6_supported': (C.c_int, [_I, _I, _I, _I, _I]),
    'te_conv_ws_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'te_conv_res_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I, _P]),
    'te_wgrad_slab_count': (C.c_int, [_I, _I, _I, _I, _I, _I]),
    'te_wgrad_pair_form': (C.c_int, [_I, _I, _I, _I, _I]),
    'te_wgrad_split_supported': (C.c_int, [_I, _I, _I, _I, _I]),
    'te_wgrad_split_bf16': (C.c_int, [_I]),
    'te_wgrad_t2_wide': (C.c_int, [_I]),
    'te_wgrad_f32': (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'te_wgrad_group_plan': (C.c_int, [_I, _I, _I, _I, _I, _I, _P, _P]),
    'te_wgrad_group_f32': (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'te_wgrad_reduce_ws_floats': (C.c_int64, [_I, _I, _I, _I, _I, _I, _I, _I]),
    'te_wgrad_reduce_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _P]),
    'te_rgb_supported': (C.c_int, [_I, _I, _I]),
    'te_rgb_fwd_f32': (C.c_int, [_P, _P, _P, _P, _P, _F, _I, _I, _I, _P]),
    'te_rgb_dgrad_f32': (C.c_int, [_P, _P, _P, _P, _F, _I, _I, _I, _P]),
    'te_rgb_wgrad_sum_f32': (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'te_rgb_expand_f32': (C.c_int, [_P, _P, _P, _P, _I, _F, _I, _I, _I, _P]),
    'te_rgb_wgrad_slab_count': (C.c_int, [_I, _I, _I]),
    'te_rgb_wgrad_f32': (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'te_blur_actgrad_tiles': (C.c_int, [_I] * 8),
    'te_blur_actgrad_f32': (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    'te_blur_gradact_f32': (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    'te_small_gemm_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _L, _L, _L, _L, _F, _F, _I, _P]),
    'te_small_gemm_batched_f32': (C.c_int, [_P, _P, _P, _P, _I, _L, _L, _L, _L, _P, _P, _I, _I, _I, _L, _L, _L, _L, _L, _L,
                                             _F, _F, _I, _P]),
    'te_layer_norm_supported': (C.c_int, [_L, _I]),
    'te_layer_norm_fwd_f32': (C.c_int, [_P, _P, _P, _L, _I, _F, _P]),
    'te_layer_norm_bwd_f32': (C.c_int, [_P, _P, _P, _P, _L, _I, _P]),
    'te_pixel_norm_supported': (C.c_int, [_L, _I, _I]),
    'te_pixel_norm_fwd_f32': (C.c_int, [_P, _P, _P, _L, _I, _I, _F, _P]),
    'te_pixel_norm_bwd_f32': (C.c_int, [_P, _P, _P, _P, _L, _I, _I, _P]),
    'te_demod_fwd_f32': (C.c_int, [_P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _P]),
    'te_demod_bwd_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P]),
    'te_attn_fwd_f32': (C.c_int, [_P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P]),
    'te_attn_bwd_f32': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P]),
    'te_small_gemm_splitk_f32': (C.c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _F, _F, _I, _P]),
    'te_minibatch_stddev_fwd_f32': (C.c_int, [_P, _P, _I, _I, _I, _I, _F, _P]),
    'te_minibatch_stddev_bwd_f32': (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _F, _P]),
    'te_mt_adam_f32': (C.c_int, [_P, _P, _I, _I, _I, C.c_double, C.c_double, C.c_double, C.c_double, _I, _P]),
    'te_mt_ema_f32': (C.c_int, [_P, _P, _I, _I, _I, C.c_double, _P]),
    'te_chan_scale_f32': (C.c_int, [_P, _P, _P, _L, _L, _P]),
    'te_chan_dot_f32': (C.c_int, [_P, _P, _P, _L, _L, _P]),
}
EXPORTS = tuple(_SIGNATURES)


def lib():
    """Load (once) and return the CDLL.  Raises RuntimeError if the extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -m transeditor_amd.build` '
                               '(there is no CPU / PyTorch fallback for the TransEditor hot path)')
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            if not hasattr(L, name):
                raise RuntimeError(f'{LIB_PATH} does not export {name}')
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.te_arch() != b'gfx950':
            raise RuntimeError('libte_hip.so was not built for gfx950')
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what} failed (code {rc}): {lib().te_last_error_string().decode()}')


_cur_dev = getattr(torch._C, '_cuda_getDevice', torch.cuda.current_device)


def _on_current_device(t):
    """The kernels are enqueued on the CURRENT device's current stream (what the reference's ops do,
    fused_bias_act_kernel.cu:90, upfirdn2d_kernel.cu:192).  Unlike the reference, a tensor that lives on another GPU is
    refused instead of being read through a foreign pointer on the wrong stream."""
    if t.device.index != _cur_dev():          # (a CUDA tensor exists, so the runtime is initialised)
        raise RuntimeError(f'te_hip: tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}; '
                           f'call torch.cuda.set_device (one process per GPU) or wrap the call in torch.cuda.device(...)')


def _ptr(t):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f'te_hip: expected a contiguous fp32 tensor on the GPU, got {t.dtype} {t.device} '
                           f'contiguous={t.is_contiguous()} (no CPU path exists)')
    _on_current_device(t)
    return t.data_ptr()


def _raw(t):
    """device pointer of a tensor addressed through explicit strides (no contiguity requirement)"""
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f'te_hip: expected an fp32 tensor on the GPU, got {t.dtype} {t.device} (no CPU path exists)')
    _on_current_device(t)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# --------------------------------------------------------------------------------------------- K1
_OTHER = {torch.float16: 'f16', torch.float64: 'f64'}       # K1 / K2 also exist in the reference's other two dispatch types


def _ptr_as(t, dtype):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise RuntimeError(f'te_hip: expected a contiguous {dtype} tensor on the GPU, got {t.dtype} {t.device} '
                           f'contiguous={t.is_contiguous()} (no CPU path exists)')
    _on_current_device(t)
    return t.data_ptr()


def bias_act(x, b, ref, act, grad, alpha, scale):
    """out = act(x + b[channel]) * scale; channel = dim 1 (step_b = prod(shape[2:]))."""
    x = x.contiguous()
    if x.dtype in _OTHER:
        out = torch.empty_like(x)
        step_b = 1
        for d in x.shape[2:]:
            step_b *= d
        fn = getattr(lib(), 'te_bias_act_' + _OTHER[x.dtype])
        _check(fn(_ptr_as(out, x.dtype), _ptr_as(x, x.dtype), _ptr_as(b, x.dtype), _ptr_as(ref, x.dtype), act, grad, alpha, scale,
                  x.numel(), step_b, b.numel() if b is not None else 1, _stream()), 'te_bias_act_' + _OTHER[x.dtype])
        return out
    out = torch.empty_like(x)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    _check(lib().te_bias_act_f32(_ptr(out), _ptr(x), _ptr(b), _ptr(ref), act, grad, alpha, scale,
                                 x.numel(), step_b, b.numel() if b is not None else 1, _stream()), 'te_bias_act_f32')
    return out


def _bias_grad_buffers(want_bias, outer, Cn, inner, device):
    """(gb, workspace): the bias gradient is WRITTEN by a fixed-order second pass over per-block partials (no atomics, so no
    zero fill and bit-reproducible); it owns its storage (it becomes a parameter's .grad)."""
    if not want_bias:
        return None, None
    n = lib().te_bias_act_bwd_ws_floats(outer, Cn, inner)
    return (torch.empty(Cn, device=device, dtype=torch.float32),
            torch.empty(n, device=device, dtype=torch.float32) if n > 0 else None)


def bias_act_bwd(g, ref, alpha, scale, want_bias=True):
    g = g.contiguous()
    if g.dtype in _OTHER:       # the reference's two steps (fused_act.py:18-38): the kernel in grad mode, then the bias sum
        gi = bias_act(g, None, ref, 3, 1, alpha, scale)
        return gi, (gi.sum(dim=[0] + list(range(2, gi.ndim))) if want_bias else None)
    gi = torch.empty_like(g)
    Cn = g.shape[1]
    inner = 1
    for d in g.shape[2:]:
        inner *= d
    gb, ws = _bias_grad_buffers(want_bias, g.shape[0], Cn, inner, g.device)
    _check(lib().te_bias_act_bwd_f32(_ptr(gi), _ptr(gb), _ptr(ws), _ptr(g), _ptr(ref), alpha, scale, g.shape[0], Cn, inner,
                                     _stream()), 'te_bias_act_bwd_f32')
    return gi, gb


def bias_act_bwd_rgb_supported(shape):
    inner = 1
    for d in shape[2:]:
        inner *= d
    return bool(lib().te_bias_act_bwd_rgb_supported(shape[0], shape[1], inner))


def bias_act_bwd_rgb(g, ref, grgb, wrgb, srgb, wscale, alpha, scale, want_bias=True):
    """activation gradient with a ToRGB data gradient folded in (see te_hip.h); g may be None.  -> (gi, gb | None)"""
    ref, grgb = ref.contiguous(), grgb.contiguous()
    g = g.contiguous() if g is not None else None
    gi = torch.empty_like(ref)
    Cn = ref.shape[1]
    inner = 1
    for d in ref.shape[2:]:
        inner *= d
    gb, ws = _bias_grad_buffers(want_bias, ref.shape[0], Cn, inner, ref.device)
    _check(lib().te_bias_act_bwd_rgb_f32(_ptr(gi), _ptr(gb), _ptr(ws), _ptr(g), _ptr(ref), _ptr(grgb), _ptr(wrgb.contiguous()),
                                         _ptr(srgb.contiguous()) if srgb is not None else None, wscale, alpha, scale,
                                         ref.shape[0], Cn, inner, _stream()), 'te_bias_act_bwd_rgb_f32')
    return gi, gb


# --------------------------------------------------------------------------------------------- K2
def upfirdn2d_raw(x, k, up, down, pad, bias=None, act=0, alpha=0.2, scale=1.0):
    """x [B,C,H,W] -> [B,C,H',W'];  pad = (px0, px1, py0, py1);  up/down = (x, y)."""
    x = x.contiguous()
    B, Cn, H, W = x.shape
    kh, kw = k.shape
    px0, px1, py0, py1 = pad
    oh = (H * up[1] + py0 + py1 - kh) // down[1] + 1
    ow = (W * up[0] + px0 + px1 - kw) // down[0] + 1
    if oh <= 0 or ow <= 0:
        raise RuntimeError(f'upfirdn2d: empty output {oh}x{ow}')
    out = torch.empty(B, Cn, oh, ow, device=x.device, dtype=x.dtype)
    if x.dtype in _OTHER:
        if bias is not None or act:
            raise RuntimeError('upfirdn2d: the fused bias / activation epilogue exists in fp32 only')
        fn = getattr(lib(), 'te_upfirdn2d_' + _OTHER[x.dtype])
        _check(fn(_ptr_as(out, x.dtype), _ptr_as(x, x.dtype), _ptr_as(k.to(x.dtype).contiguous(), x.dtype), B * Cn, H, W, 1, kh, kw,
                  up[0], up[1], down[0], down[1], px0, px1, py0, py1, _stream()), 'te_upfirdn2d_' + _OTHER[x.dtype])
        return out
    _check(lib().te_upfirdn2d_f32(_ptr(out), _ptr(x), _ptr(k.contiguous()), B * Cn, H, W, 1, kh, kw, up[0], up[1],
                                  down[0], down[1], px0, px1, py0, py1, _ptr(bias),
                                  bias.numel() if bias is not None else 1, act, alpha, scale, _stream()),
           'te_upfirdn2d_f32')
    return out


def blur_actgrad(g, ref, k_flipped, pad, alpha, scale):
    """backward of blur + bias + lrelu in one pass: returns (gx, gbias).  g / ref [B,C,H,W]; pad = (px0, px1, py0, py1)."""
    g, ref = g.contiguous(), ref.contiguous()
    B, Cn, H, W = g.shape
    kh, kw = k_flipped.shape
    px0, px1, py0, py1 = pad
    tiles = lib().te_blur_actgrad_tiles(H, W, kh, kw, px0, px1, py0, py1)
    if tiles <= 0:
        raise RuntimeError(f'te_blur_actgrad_tiles failed ({tiles})')
    gx = torch.empty(B, Cn, H + py0 + py1 - kh + 1, W + px0 + px1 - kw + 1, device=g.device, dtype=g.dtype)
    partial = torch.empty(B, Cn, tiles, device=g.device, dtype=g.dtype)
    _check(lib().te_blur_actgrad_f32(_ptr(gx), _ptr(partial), _ptr(g), _ptr(ref), _ptr(k_flipped.contiguous()), B * Cn, H, W,
                                     kh, kw, px0, px1, py0, py1, alpha, scale, _stream()), 'te_blur_actgrad_f32')
    return gx, partial.sum(dim=(0, 2))


def blur_gradact(g, ref, k_flipped, pad, alpha, scale):
    """backward of 'bias + lrelu -> blur' in one pass: (upfirdn2d(g, k_flipped, pad) * slope(ref), gbias).  ref is shaped
    like the result."""
    g, ref = g.contiguous(), ref.contiguous()
    B, Cn, H, W = g.shape
    kh, kw = k_flipped.shape
    px0, px1, py0, py1 = pad
    oh, ow = H + py0 + py1 - kh + 1, W + px0 + px1 - kw + 1
    if tuple(ref.shape) != (B, Cn, oh, ow):
        raise RuntimeError(f'blur_gradact: ref {tuple(ref.shape)} is not the shape of the adjoint blur output {(B, Cn, oh, ow)}')
    tiles = lib().te_blur_actgrad_tiles(H, W, kh, kw, px0, px1, py0, py1)
    if tiles <= 0:
        raise RuntimeError(f'te_blur_actgrad_tiles failed ({tiles})')
    gx = torch.empty(B, Cn, oh, ow, device=g.device, dtype=g.dtype)
    partial = torch.empty(B, Cn, tiles, device=g.device, dtype=g.dtype)
    _check(lib().te_blur_gradact_f32(_ptr(gx), _ptr(partial), _ptr(g), _ptr(ref), _ptr(k_flipped.contiguous()), B * Cn, H, W,
                                     kh, kw, px0, px1, py0, py1, alpha, scale, _stream()), 'te_blur_gradact_f32')
    return gx, partial.sum(dim=(0, 2))


# --------------------------------------------------------------------------------------------- F1
def conv_pack(w, kind_pack, wscale=1.0):
    """w [Co,Ci,k,k] (model layout) -> packed Wp[tap][Kp][Mp]."""
    w = w.contiguous()
    Co, Ci, ks, _ = w.shape
    n = lib().te_conv_packed_numel(kind_pack, Co, Ci, ks)
    wp = torch.empty(n, device=w.device, dtype=w.dtype)
    _check(lib().te_conv_pack_weights_f32(_ptr(wp), _ptr(w), wscale, kind_pack, Co, Ci, ks, _stream()),
           'te_conv_pack_weights_f32')
    return wp


def conv_pack2(w, kind_a, kind_b, wscale=1.0):
    """both packed layouts of w in one launch -> (wp_a, wp_b)"""
    w = w.contiguous()
    Co, Ci, ks, _ = w.shape
    wa = torch.empty(lib().te_conv_packed_numel(kind_a, Co, Ci, ks), device=w.device, dtype=w.dtype)
    wb = torch.empty(lib().te_conv_packed_numel(kind_b, Co, Ci, ks), device=w.device, dtype=w.dtype)
    _check(lib().te_conv_pack_weights2_f32(_ptr(wa), kind_a, _ptr(wb), kind_b, _ptr(w), wscale, Co, Ci, ks, _stream()),
           'te_conv_pack_weights2_f32')
    return wa, wb


def conv_pack_multi(jobs):
    """jobs: [(wp, w, kind_pack, wscale)] with w [Co,Ci,k,k] dense and wp an already allocated packed buffer of the right
    size: every layout is (re)written in one launch per 64 jobs."""
    n = len(jobs)
    if not n:
        return
    for wp, w, _, _ in jobs:
        if not (w.is_contiguous() and wp.is_contiguous()):
            raise RuntimeError('te_hip: conv_pack_multi needs dense weights and buffers')
    arr = lambda ty, vals: (ty * n)(*vals)
    _check(lib().te_conv_pack_weights_multi_f32(
        n, arr(C.c_void_p, [_ptr(j[0]) for j in jobs]), arr(C.c_void_p, [_ptr(j[1]) for j in jobs]),
        arr(C.c_float, [float(j[3]) for j in jobs]), arr(C.c_int, [int(j[2]) for j in jobs]),
        arr(C.c_int, [j[1].shape[0] for j in jobs]), arr(C.c_int, [j[1].shape[1] for j in jobs]),
        arr(C.c_int, [j[1].shape[2] for j in jobs]), _stream()), 'te_conv_pack_weights_multi_f32')


def wino_ok(B, K, M, H, W):
    """does TE_CONV_3X3W (1-D Winograd F(2,3): 2/3 of the MFMAs of the direct 3x3 kernel) cover this problem?"""
    return bool(lib().te_conv_wino_supported(B, K, M, H, W))


def s2s6_ok(B, K, M, H, W):
    """does TE_CONV_S2S6 (the stride-2 convolution on the bf16 matrix pipe, three-piece split) cover this problem?  H, W = output size"""
    return bool(lib().te_conv_s2s6_supported(B, K, M, H, W))


def t2s6_ok(B, K, M, H, W):
    """does TE_CONV_T2S6 (the transposed stride-2 convolution on the bf16 matrix pipe) cover this problem?  H, W = input (low-res) size"""
    return bool(lib().te_conv_t2s6_supported(B, K, M, H, W))


def p1s6_ok(B, K, M, H, W):
    """does TE_CONV_1X1S6 (the 1x1 convolution on the bf16 matrix pipe) cover this problem?"""
    return bool(lib().te_conv_p1s6_supported(B, K, M, H, W))


def wino6_form(form=-1):
    """kernel form of TE_CONV_3X3W6: 2 = two-image (default; M % 128 == 0, else ping-pong), 1 = ping-pong, 0 = block-phase (bit-identical
    results); returns the previous value (-1: query only)"""
    return int(lib().te_conv_wino6_form(form))


def s2s6_form(form=-1):
    """kernel form of TE_CONV_S2S6: 1 = two-image (default; M % 128 == 0 and a block per CU, else ping-pong), 2 = two-image wherever
    M % 128 == 0, 0 = ping-pong (bit-identical results); returns the previous value (-1: query only)"""
    return int(lib().te_conv_s2s6_form(form))


def t2s6_form(form=-1):
    """kernel form of TE_CONV_T2S6: as s2s6_form"""
    return int(lib().te_conv_t2s6_form(form))


def wino6_ok(B, K, M, H, W):
    """does TE_CONV_3X3W6 (the Winograd form on the bf16 matrix pipe, three-piece split, fp32-equivalent) cover this problem?"""
    return bool(lib().te_conv_wino6_supported(B, K, M, H, W))


def conv_out_shape(kind, B, M, H, W):
    if kind in (CONV_T2, CONV_T2S6):
        return (B, M, 2 * H + 1, 2 * W + 1)
    return (B, M, H, W)


def conv(x, wp, kind, M, H, W, isc=None, osc=None, bias=None, act=0, res=None, mask_ref=None, mask_gain=1.0):
    """H, W = LOW-resolution size (see te_hip.h).  x [B,K,Hin,Win].  res: residual added after the activation; mask_ref:
    leaky-ReLU gradient mask (saved output of the layer this data gradient lands on) applied last."""
    x = x.contiguous()
    B, K = x.shape[0], x.shape[1]
    out = torch.empty(conv_out_shape(kind, B, M, H, W), device=x.device, dtype=x.dtype)
    if res is not None and tuple(res.shape) != tuple(out.shape):
        raise RuntimeError(f'te_hip: residual {tuple(res.shape)} does not match the convolution output {tuple(out.shape)}')
    if mask_ref is not None and tuple(mask_ref.shape) != tuple(out.shape):
        raise RuntimeError(f'te_hip: mask reference {tuple(mask_ref.shape)} does not match the convolution output {tuple(out.shape)}')
    S = lib().te_conv_splitk_count(kind, B, K, M, H, W)
    if S < 1:
        raise RuntimeError(f'te_conv_splitk_count failed ({S})')
    # small images split the channel loop over the grid: per-split slabs + fixed-order sum (deterministic, graph-capturable)
    ws = torch.empty((S,) + tuple(out.shape), device=x.device, dtype=x.dtype) if S > 1 else None
    if kind == CONV_T2S6:          # scratch for the last input column (body kernel -> edge kernel; te_hip.h)
        ws = torch.empty(B * K * H, device=x.device, dtype=x.dtype)
    _check(lib().te_conv_res_f32(_ptr(out), _ptr(ws), _ptr(x), _ptr(wp), _ptr(isc), _ptr(osc), _ptr(bias),
                                 _ptr(res.contiguous()) if res is not None else None,
                                 _ptr(mask_ref.contiguous()) if mask_ref is not None else None, mask_gain, act, kind, B, K, M, H, W,
                                 _stream()),
           'te_conv_res_f32')
    return out


def wgrad_slabs(g, x, kind, H, W, group=False):
    """correlation slabs [B, S, Co, Ci, taps]; group=True (PLAIN gradient only - nothing per sample is derived from the slabs):
    [B / NB, S, Co, Ci, taps] with NB samples per slab where the plan finds that worthwhile (small images, big weights)"""
    g, x = g.contiguous(), x.contiguous()
    B, Co, Ci = g.shape[0], g.shape[1], x.shape[1]
    taps = 1 if kind == CONV_1X1 else 9
    if group:
        nb, sc = C.c_int(0), C.c_int(0)
        _check(lib().te_wgrad_group_plan(kind, B, Co, Ci, H, W, C.byref(nb), C.byref(sc)), 'te_wgrad_group_plan')
        if nb.value > 1:
            slabs = torch.empty(B // nb.value, sc.value, Co, Ci, taps, device=g.device, dtype=g.dtype)
            _check(lib().te_wgrad_group_f32(_ptr(slabs), _ptr(g), _ptr(x), kind, B, Co, Ci, H, W, sc.value, nb.value, _stream()),
                   'te_wgrad_group_f32')
            return slabs
    S = lib().te_wgrad_slab_count(kind, B, Co, Ci, H, W)
    if S <= 0:
        raise RuntimeError(f'te_wgrad_slab_count failed ({S})')
    slabs = torch.empty(B, S, Co, Ci, taps, device=g.device, dtype=g.dtype)
    _check(lib().te_wgrad_f32(_ptr(slabs), _ptr(g), _ptr(x), kind, B, Co, Ci, H, W, S, _stream()), 'te_wgrad_f32')
    return slabs


def wgrad_t2_wide(on=-1):
    """form of the split transposed-kind weight-gradient kernel: 1 = 64 x 128 channels per block where Ci % 128 == 0 (default), 0 = 64 x 64
    (bit-identical slabs); returns the previous value (-1: query only)"""
    return int(lib().te_wgrad_t2_wide(on))


def wgrad_split(on=-1):
    """switch of the split-bf16 weight-gradient kernel (csrc/wgrad6.hip): 0 / 1 sets it, returns the previous value (-1: query only)"""
    return int(lib().te_wgrad_split_bf16(on))


def wgrad_split_ok(kind, Co, Ci, H, W):
    """does the split-bf16 weight-gradient kernel cover this problem?  (taken only while wgrad_split() is on)"""
    return bool(lib().te_wgrad_split_supported(kind, Co, Ci, H, W))


def wgrad_pair_form(kind, Co, Ci, H, W):
    """does the weight-gradient kernel take the pair (Winograd F(3,2)) form for this problem?  (FLOP accounting only)"""
    return bool(lib().te_wgrad_pair_form(kind, Co, Ci, H, W))


def wgrad_reduce(slabs, w, wscale=1.0, isc=None, osc=None, want_w=True, want_isc=False, want_osc=False):
    B, S, Co, Ci, taps = slabs.shape
    dev, dt = slabs.device, slabs.dtype
    # all three outputs are WRITTEN (shares of different blocks meet in the workspace, summed in a fixed order: no atomics)
    gw = torch.empty(Co, Ci, taps, device=dev, dtype=dt) if want_w else None
    gisc = torch.empty(B, Ci, device=dev, dtype=dt) if want_isc else None
    gosc = torch.empty(B, Co, device=dev, dtype=dt) if want_osc else None
    n = lib().te_wgrad_reduce_ws_floats(B, S, Co, Ci, taps, int(want_w), int(want_isc), int(want_osc))
    if n < 0:
        raise RuntimeError(f'te_wgrad_reduce_ws_floats failed ({n})')
    ws = torch.empty(n, device=dev, dtype=dt) if n > 0 else None
    _check(lib().te_wgrad_reduce_f32(_ptr(gw), _ptr(gisc), _ptr(gosc), _ptr(ws), _ptr(slabs), _ptr(w.contiguous()), wscale,
                                     _ptr(isc), _ptr(osc), B, S, Co, Ci, taps, _stream()), 'te_wgrad_reduce_f32')
    return gw, gisc, gosc


# --------------------------------------------------------------------------------------------- M3
def rgb_supported(M, K, HW):
    return bool(lib().te_rgb_supported(M, K, HW))


def rgb_fwd(x, w, isc, bias, wscale=1.0):
    x = x.contiguous()
    B, K, H, W = x.shape
    out = torch.empty(B, 3, H, W, device=x.device, dtype=x.dtype)
    _check(lib().te_rgb_fwd_f32(_ptr(out), _ptr(x), _ptr(w.contiguous()), _ptr(isc), _ptr(bias), wscale, B, K, H * W, _stream()),
           'te_rgb_fwd_f32')
    return out


def rgb_dgrad(g, w, isc, K, wscale=1.0):
    g = g.contiguous()
    B, _, H, W = g.shape
    gx = torch.empty(B, K, H, W, device=g.device, dtype=g.dtype)
    _check(lib().te_rgb_dgrad_f32(_ptr(gx), _ptr(g), _ptr(w.contiguous()), _ptr(isc), wscale, B, K, H * W, _stream()),
           'te_rgb_dgrad_f32')
    return gx


def rgb_wgrad_sum_slabs(g3, x):
    """slabs [B,S,4,K]: rows 0-2 = sum_p g3[b,o,p] x[b,k,p], row 3 = sum_p x[b,k,p]"""
    g3, x = g3.contiguous(), x.contiguous()
    B, K, H, W = x.shape
    S = lib().te_rgb_wgrad_slab_count(B, K, H * W)
    slabs = torch.empty(B, S, 4, K, device=x.device, dtype=x.dtype)
    _check(lib().te_rgb_wgrad_sum_f32(_ptr(slabs), _ptr(g3), _ptr(x), B, K, H * W, S, _stream()), 'te_rgb_wgrad_sum_f32')
    return slabs


def rgb_expand(x3, w3k, bias, act, wscale=1.0):
    """from-RGB stem: x3 [B,3,H,W], w3k [3,K] -> act(wscale * sum_o w3k[o,k] x3[b,o] + bias[k])  [B,K,H,W]"""
    x3 = x3.contiguous()
    B, _, H, W = x3.shape
    K = w3k.shape[1]
    out = torch.empty(B, K, H, W, device=x3.device, dtype=x3.dtype)
    _check(lib().te_rgb_expand_f32(_ptr(out), _ptr(x3), _ptr(w3k.contiguous()), _ptr(bias), act, wscale, B, K, H * W, _stream()),
           'te_rgb_expand_f32')
    return out


def rgb_wgrad_slabs(g, x):
    g, x = g.contiguous(), x.contiguous()
    B, K, H, W = x.shape
    S = lib().te_rgb_wgrad_slab_count(B, K, H * W)
    slabs = torch.empty(B, S, 3, K, 1, device=x.device, dtype=x.dtype)
    _check(lib().te_rgb_wgrad_f32(_ptr(slabs), _ptr(g), _ptr(x), B, K, H * W, S, _stream()), 'te_rgb_wgrad_f32')
    return slabs


# --------------------------------------------------------------------------------------------- G2/A2
def small_gemm(I, J, K, a, sai, sak, b, sbk, sbj, bias=None, residual=None, alpha=1.0, beta=1.0, act=0, want_pre=False,
               rowsum_scale=None):
    """C[I,J] = act(alpha * A B + beta * bias) + residual with strided operands (see te_hip.h); a / b are the base
    tensors (contiguous storage, addressed through the element strides).  Returns (C, pre | None, rowsum | None);
    rowsum_scale != None asks for rowsum[i] = rowsum_scale * sum_k A(i,k)."""
    c = torch.empty(I, J, device=a.device, dtype=a.dtype)
    pre = torch.empty_like(c) if want_pre else None
    rs = torch.empty(I, device=a.device, dtype=a.dtype) if rowsum_scale is not None else None
    # a / b are addressed through explicit strides: only device / dtype are checked
    _check(lib().te_small_gemm_f32(_ptr(c), _ptr(pre), _raw(a), _raw(b), _ptr(bias), _ptr(residual), _ptr(rs),
                                   rowsum_scale if rowsum_scale is not None else 0.0, I, J, K, sai, sak, sbk, sbj, alpha,
                                   beta, act, _stream()), 'te_small_gemm_f32')
    return c, pre, rs


def small_gemm_splitk(I, J, K, S, a, sai, sak, b, sbk, sbj, bias=None, residual=None, alpha=1.0, beta=1.0, act=0, want_pre=False):
    """small_gemm for wide reductions: K in S chunks over the grid, fixed-order second pass (see te_hip.h)."""
    c = torch.empty(I, J, device=a.device, dtype=a.dtype)
    pre = torch.empty_like(c) if want_pre else None
    ws = torch.empty(S, I, J, device=a.device, dtype=a.dtype)
    _check(lib().te_small_gemm_splitk_f32(_ptr(c), _ptr(pre), _ptr(ws), S, _raw(a), _raw(b), _ptr(bias), _ptr(residual), I, J, K,
                                          sai, sak, sbk, sbj, alpha, beta, act, _stream()), 'te_small_gemm_splitk_f32')
    return c, pre


def minibatch_stddev_fwd(x, group, eps, chunks=1):
    """x [B, C, H, W] -> y [B, C + 1, H, W] (input copied, stddev channel appended); `chunks` > 1: the batch is `chunks`
    independent minibatches laid end to end (one launch each on its slice)"""
    x = x.contiguous()
    B, Cn, H, W = x.shape
    y = torch.empty(B, Cn + 1, H, W, device=x.device, dtype=x.dtype)
    Bc = B // chunks
    for c in range(chunks):
        _check(lib().te_minibatch_stddev_fwd_f32(_ptr(y[c * Bc:(c + 1) * Bc]), _ptr(x[c * Bc:(c + 1) * Bc]), Bc, group, Cn, H * W, eps,
                                                 _stream()), 'te_minibatch_stddev_fwd_f32')
    return y


def minibatch_stddev_bwd(gy, x, group, eps, chunks=1):
    gy = gy.contiguous()
    B, Cn, H, W = x.shape
    gx = torch.empty_like(x)
    Bc = B // chunks
    for c in range(chunks):
        sl = slice(c * Bc, (c + 1) * Bc)
        _check(lib().te_minibatch_stddev_bwd_f32(_ptr(gx[sl]), _ptr(gy[sl]), _ptr(x[sl]), Bc, group, Cn, H * W, eps, _stream()),
               'te_minibatch_stddev_bwd_f32')
    return gx


def small_gemm_batched(c, a, b, bias, nz, za, zc, I, J, K, sai, sak, sbk, sbj, sci, scj, zb=0, zbias=0, b_tab=None,
                       bias_tab=None, alpha=1.0, beta=1.0, act=0):
    """nz GEMMs in one launch (see te_hip.h); c is written in place through (zc, sci, scj).  b_tab / bias_tab: lists of
    element offsets relative to b / bias for separately allocated per-z operands."""
    bt = (C.c_int64 * nz)(*b_tab) if b_tab is not None else None
    bit = (C.c_int64 * nz)(*bias_tab) if bias_tab is not None else None
    _check(lib().te_small_gemm_batched_f32(_raw(c), _raw(a), _raw(b), _raw(bias),
                                           nz, za, zc, zb, zbias, bt, bit, I, J, K, sai, sak, sbk, sbj, sci, scj, alpha, beta,
                                           act, _stream()), 'te_small_gemm_batched_f32')
    return c


def layer_norm_supported(R, N):
    return bool(lib().te_layer_norm_supported(R, N))


def layer_norm_fwd(x2, eps):
    """x2 [R, N] contiguous -> (y [R, N], stats [R, 2])"""
    R, N = x2.shape
    y = torch.empty_like(x2)
    stats = torch.empty(R, 2, device=x2.device, dtype=x2.dtype)
    _check(lib().te_layer_norm_fwd_f32(_ptr(y), _ptr(stats), _ptr(x2), R, N, eps, _stream()), 'te_layer_norm_fwd_f32')
    return y, stats


def layer_norm_bwd(g2, y2, stats):
    R, N = y2.shape
    gx = torch.empty_like(y2)
    _check(lib().te_layer_norm_bwd_f32(_ptr(gx), _ptr(g2), _ptr(y2), _ptr(stats), R, N, _stream()), 'te_layer_norm_bwd_f32')
    return gx


def pixel_norm_supported(B, D, Cn):
    return bool(lib().te_pixel_norm_supported(B, D, Cn))


def pixel_norm_fwd(x, eps):
    """x [B, D, C] contiguous -> (y, r [B, C])"""
    B, D, Cn = x.shape
    y = torch.empty_like(x)
    r = torch.empty(B, Cn, device=x.device, dtype=x.dtype)
    _check(lib().te_pixel_norm_fwd_f32(_ptr(y), _ptr(r), _ptr(x), B, D, Cn, eps, _stream()), 'te_pixel_norm_fwd_f32')
    return y, r


def pixel_norm_bwd(g, y, r):
    B, D, Cn = y.shape
    gx = torch.empty_like(y)
    _check(lib().te_pixel_norm_bwd_f32(_ptr(gx), _ptr(g), _ptr(y), _ptr(r), B, D, Cn, _stream()), 'te_pixel_norm_bwd_f32')
    return gx


# --------------------------------------------------------------------------------------------- M1
def demod_fwd(w, s, wscale, eps):
    """w [Co,Ci,T] raw weights, s [B,Ci] -> (d [B,Co], wsq [Co,Ci])"""
    Co, Ci, T = w.shape
    B = s.shape[0]
    d = torch.empty(B, Co, device=w.device, dtype=w.dtype)
    wsq = torch.empty(Co, Ci, device=w.device, dtype=w.dtype)
    _check(lib().te_demod_fwd_f32(_ptr(d), _ptr(wsq), _ptr(w), _ptr(s), wscale, eps, B, Co, Ci, T, _stream()),
           'te_demod_fwd_f32')
    return d, wsq


def demod_from_wsq(wsq, s, eps):
    """d [B,Co] from a cached wsq [Co,Ci] (frozen weights: the 9-tap squares are not recomputed)"""
    Co, Ci = wsq.shape
    B = s.shape[0]
    d = torch.empty(B, Co, device=wsq.device, dtype=wsq.dtype)
    _check(lib().te_demod_fwd_f32(_ptr(d), None, _ptr(wsq), _ptr(s), 1.0, eps, B, Co, Ci, 0, _stream()), 'te_demod_fwd_f32')
    return d


def demod_bwd(gd, d, w, wsq, s, wscale, want_w=True, want_s=True, into=None):
    """into = (gw, gs): accumulate into these existing gradients (either may be None) instead of allocating new ones."""
    Co, Ci, T = w.shape
    B = s.shape[0]
    if into is not None:
        gw, gs = into
    else:
        gw = torch.empty_like(w) if want_w else None
        gs = torch.empty_like(s) if want_s else None
    _check(lib().te_demod_bwd_f32(_ptr(gw), _ptr(gs), _ptr(gd.contiguous()), _ptr(d), _ptr(w), _ptr(wsq), _ptr(s), wscale,
                                  B, Co, Ci, T, 1 if into is not None else 0, _stream()), 'te_demod_bwd_f32')
    return gw, gs


# --------------------------------------------------------------------------------------------- F2
def attn_fwd(q, k, v, scale, groups):
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    N, M, Cn = q.shape
    L = k.shape[1]
    D = Cn // groups
    o = torch.empty_like(q)
    sim = torch.empty(N, groups, M, L, device=q.device, dtype=q.dtype)
    _check(lib().te_attn_fwd_f32(_ptr(o), _ptr(sim), _ptr(q), _ptr(k), _ptr(v), scale, N, groups, M, L, D, _stream()),
           'te_attn_fwd_f32')
    return o, sim


def attn_bwd(go, gsim, q, k, v, sim, scale, groups):
    N, M, Cn = q.shape
    L = k.shape[1]
    gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _check(lib().te_attn_bwd_f32(_ptr(gq), _ptr(gk), _ptr(gv), _ptr(go.contiguous()),
                                 _ptr(gsim.contiguous()) if gsim is not None else None, _ptr(q), _ptr(k), _ptr(v),
                                 _ptr(sim), scale, N, groups, M, L, Cn // groups, _stream()), 'te_attn_bwd_f32')
    return gq, gk, gv


# --------------------------------------------------------------------------------------------- T2
def mt_adam(table, chunks, n_tensors, n_chunks, chunk_elems, lr, beta1, beta2, eps, step):
    """table: int64 device tensor [5, n]; chunks: int32 device tensor [2, n_chunks] (see te_hip.h)"""
    _on_current_device(table)
    _check(lib().te_mt_adam_f32(table.data_ptr(), chunks.data_ptr(), n_tensors, n_chunks, chunk_elems, lr, beta1, beta2, eps,
                                step, _stream()), 'te_mt_adam_f32')


def mt_ema(table, chunks, n_tensors, n_chunks, chunk_elems, decay):
    _on_current_device(table)
    _check(lib().te_mt_ema_f32(table.data_ptr(), chunks.data_ptr(), n_tensors, n_chunks, chunk_elems, float(decay), _stream()),
           'te_mt_ema_f32')


# --------------------------------------------------------------------------------------------- channel scale / dot
def chan_scale(x, s):
    """x [B,C,...] * s[B,C] broadcast over the trailing dims"""
    x, s = x.contiguous(), s.contiguous()
    rows = x.shape[0] * x.shape[1]
    out = torch.empty_like(x)
    _check(lib().te_chan_scale_f32(_ptr(out), _ptr(x), _ptr(s), rows, x.numel() // max(rows, 1), _stream()), 'te_chan_scale_f32')
    return out


def chan_dot(a, b):
    """sum over the trailing dims of a * b -> [B,C]"""
    a, b = a.contiguous(), b.contiguous()
    rows = a.shape[0] * a.shape[1]
    out = torch.empty(a.shape[0], a.shape[1], device=a.device, dtype=a.dtype)
    _check(lib().te_chan_dot_f32(_ptr(out), _ptr(a), _ptr(b), rows, a.numel() // max(rows, 1), _stream()), 'te_chan_dot_f32')
    return out


# --------------------------------------------------------------------------------------------- roctx ranges (SURVEY §5 tracing)
# TE_ROCTX=1: every tensor-level wrapper above runs inside a roctx range "te:<op> <shape of its first tensor>", so a
# `rocprofv3 --kernel-trace --marker-trace` timeline attributes kernels to operators instead of showing template names only
# (torch.cuda.nvtx is backed by roctx on ROCm builds).  Off by default: two extra calls per launch.
def _install_roctx():
    import functools
    import torch.cuda.nvtx as nvtx
    names = ['bias_act', 'bias_act_bwd', 'upfirdn2d_raw', 'blur_actgrad', 'blur_gradact', 'conv_pack', 'conv_pack2', 'conv_pack_multi', 'conv',
             'wgrad_slabs', 'wgrad_reduce', 'rgb_fwd', 'rgb_dgrad', 'rgb_expand', 'rgb_wgrad_slabs', 'small_gemm',
             'small_gemm_splitk', 'small_gemm_batched', 'minibatch_stddev_fwd', 'minibatch_stddev_bwd',
             'layer_norm_fwd', 'layer_norm_bwd', 'pixel_norm_fwd', 'pixel_norm_bwd', 'demod_fwd', 'demod_from_wsq', 'demod_bwd',
             'attn_fwd', 'attn_bwd', 'mt_adam', 'mt_ema', 'chan_scale', 'chan_dot']
    g = globals()

    def wrap(fn, name):
        @functools.wraps(fn)
        def run(*a, **k):
            t = next((x for x in a if torch.is_tensor(x)), None)
            nvtx.range_push(f'te:{name} {tuple(t.shape)}' if t is not None else f'te:{name}')
            try:
                return fn(*a, **k)
            finally:
                nvtx.range_pop()
        return run
    for n in names:
        if n in g:
            g[n] = wrap(g[n], n)


ROCTX = os.environ.get('TE_ROCTX') == '1'
if ROCTX:
    _install_roctx()
