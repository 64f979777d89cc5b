"""CPU oracle for the TransEditor generator/discriminator hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``transeditor_amd/`` may import this
module; it is used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` as the *checker*, never as the product.

This is a functional (state_dict-driven) restatement in plain PyTorch of the
algorithm in the reference ``model_spatial_query.py`` and ``utils/op``.  It is
deliberately written without nn.Module classes: every function takes the
parameter dictionary ``P`` (keys = the reference's state_dict schema, SURVEY
§8b) and follows the reference formulas line by line (file:line cited on each
function; paths relative to the reference repo root).

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY §4).  The oracle is therefore pinned against outputs of the reference
itself, imported in the build container by ``oracle/gen_golden.py`` (three
import shims, no edits to reference files); the resulting vectors are committed
under ``tests/golden/`` and ``tests/test_oracle_golden.py`` re-checks the
oracle against them on every run.

All arithmetic is whatever dtype the inputs carry (fp32 for parity, fp64 for
gradcheck).  Every op is built from differentiable torch primitives, so first
and second derivatives come from autograd.
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = 2 ** 0.5

# channel table, model_spatial_query.py:473-483 (channel_multiplier applied by caller)


def channel_table(channel_multiplier=2):
    cm = channel_multiplier
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm,
            256: 64 * cm, 512: 32 * cm, 1024: 16 * cm}


# --------------------------------------------------------------------------- ops

# Test hook: a callable that receives the output of EVERY leaky-ReLU the oracle evaluates, in execution order
# (tests/pinning.py records the slope signs the fp64 oracle took and pins the HIP path's activation gradients to them, so that
# gradient parity can be asserted without the slope flips of pre-activations that sit within round-off of the kink).
ACT_TAP = None


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=SQRT2):
    """utils/op/fused_act.py:50-56 + fused_bias_act_kernel.cu:26-47 (act=3, grad=0):
    y = lrelu(x + b[channel]) * scale, channel = dim 1."""
    if bias is not None:
        x = x + bias.reshape(1, -1, *([1] * (x.ndim - 2)))
    y = torch.where(x > 0, x, x * negative_slope) * scale
    if ACT_TAP is not None:
        ACT_TAP(y)
    return y


def fir_kernel(taps, gain=1.0):
    """make_kernel, model_spatial_query.py:84-92: outer product, normalised to sum 1."""
    k = torch.as_tensor(taps, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum() * gain


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """utils/op/upfirdn2d.py:143-148 (same pad on x and y) with the arithmetic of
    upfirdn2d_kernel.cu:85-129: zero-insert by `up`, pad/crop, TRUE convolution with
    `kernel` (taps flipped, :77), keep every `down`-th sample.
    out = (in*up + pad0 + pad1 - k)//down + 1  (upfirdn2d.py:101-102)."""
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    z = x.new_zeros(B * C, 1, H * up, W * up)
    z[:, :, ::up, ::up] = x.reshape(B * C, 1, H, W)
    z = F.pad(z, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    z = z[:, :, max(-p0, 0): z.shape[2] - max(-p1, 0), max(-p0, 0): z.shape[3] - max(-p1, 0)]
    w = torch.flip(kernel, [0, 1]).to(x.dtype).reshape(1, 1, kh, kw)
    y = F.conv2d(z, w)[:, :, ::down, ::down]
    return y.reshape(B, C, y.shape[2], y.shape[3])


def pixel_norm(x, dim):
    """model_spatial_query.py:80-81."""
    return x * torch.rsqrt(torch.mean(x * x, dim=dim, keepdim=True) + 1e-8)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    """EqualLinear.forward, model_spatial_query.py:210-221."""
    w = weight * ((1.0 / math.sqrt(weight.shape[1])) * lr_mul)
    if activation:
        return fused_leaky_relu(F.linear(x, w), bias * lr_mul)
    return F.linear(x, w, None if bias is None else bias * lr_mul)


def modulated_conv2d(x, style, weight, mod_w, mod_b, demodulate=True, upsample=False,
                     blur_taps=(1, 3, 3, 1), downsample=False):
    """ModulatedConv2d.forward, model_spatial_query.py:296-337 (the downsample branch, :323-329, is not used by G).
    weight: [1, Cout, Cin, k, k]; per-sample weights, grouped conv with groups=batch."""
    B, Cin, H, W = x.shape
    _, Cout, _, k, _ = weight.shape
    s = equal_linear(style, mod_w, mod_b).reshape(B, 1, Cin, 1, 1)          # :299 (bias_init=1 lives in mod_b)
    w = (1.0 / math.sqrt(Cin * k * k)) * weight * s                          # :300
    if demodulate:
        d = torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8)                      # :303
        w = w * d.reshape(B, Cout, 1, 1, 1)
    if upsample:
        wt = w.transpose(1, 2).reshape(B * Cin, Cout, k, k)                  # :315-317
        y = F.conv_transpose2d(x.reshape(1, B * Cin, H, W), wt, padding=0, stride=2, groups=B)
        y = y.reshape(B, Cout, y.shape[2], y.shape[3])
        p = (len(blur_taps) - 2) - (k - 1)                                   # :264-266
        pad = ((p + 1) // 2 + 1, p // 2 + 1)
        return upfirdn2d(y, fir_kernel(blur_taps, 4.0).to(y), pad=pad)       # Blur, :137-153 (gain = factor**2)
    if downsample:
        p = (len(blur_taps) - 2) + (k - 1)                                   # :270-276
        x = upfirdn2d(x, fir_kernel(blur_taps).to(x), pad=((p + 1) // 2, p // 2))          # :324
        y = F.conv2d(x.reshape(1, B * Cin, x.shape[2], x.shape[3]), w.reshape(B * Cout, Cin, k, k), padding=0, stride=2,
                     groups=B)                                               # :327
        return y.reshape(B, Cout, y.shape[2], y.shape[3])
    y = F.conv2d(x.reshape(1, B * Cin, H, W), w.reshape(B * Cout, Cin, k, k), padding=k // 2, groups=B)
    return y.reshape(B, Cout, H, W)


def styled_conv(P, pre, x, style, upsample=False, noise=None, inject_noise=False):
    """StyledConv.forward, model_spatial_query.py:395-403."""
    y = modulated_conv2d(x, style, P[pre + '.conv.weight'], P[pre + '.conv.modulation.weight'],
                         P[pre + '.conv.modulation.bias'], True, upsample)
    if inject_noise:                                                          # :340-351
        if noise is None:
            noise = torch.randn(y.shape[0], 1, y.shape[2], y.shape[3], dtype=y.dtype)
        y = y + P[pre + '.noise.weight'] * noise
    return fused_leaky_relu(y, P[pre + '.activate.bias'])


def to_rgb(P, pre, x, style, skip=None):
    """ToRGB.forward, model_spatial_query.py:416-425; Upsample pad from :103-108."""
    y = modulated_conv2d(x, style, P[pre + '.conv.weight'], P[pre + '.conv.modulation.weight'],
                         P[pre + '.conv.modulation.bias'], demodulate=False)
    y = y + P[pre + '.bias']
    if skip is not None:
        y = y + upfirdn2d(skip, fir_kernel((1, 3, 3, 1), 4.0).to(y), up=2, pad=(2, 1))
    return y


def attention(P, pre, x, p, lr_mul, groups=4):
    """Attention.forward, model_spatial_query.py:883-901.  Scale is planes**-0.5 (:873),
    the `reshape(N, planes, L)` quirk of :894 is kept."""
    N, L, _ = x.shape
    M = p.shape[1]
    planes = P[pre + '.q_transform.weight'].shape[0]
    gp = planes // groups
    lin = lambda t, n: equal_linear(t, P[f'{pre}.{n}.weight'], P[f'{pre}.{n}.bias'], lr_mul)
    q = lin(p, 'q_transform').reshape(N, M, groups, gp).permute(0, 2, 3, 1)
    k = lin(x, 'k_transform').reshape(N, L, groups, gp).permute(0, 2, 3, 1)
    v = lin(x, 'v_transform').reshape(N, L, groups, gp).permute(0, 2, 3, 1)
    qk = torch.einsum('abcd,abce->abde', q, k) * planes ** -0.5
    sim = F.softmax(qk, dim=3)
    sv = torch.einsum('abcd,abed->abec', sim, v)
    out = lin(sv.reshape(N, planes, L).permute(0, 2, 1), 'proj')
    return out, sim


def attention_block(P, pre, x, p, lr_mul):
    """AttentionBlock.forward, model_spatial_query.py:920-936 (LayerNorm over tokens x channels, no affine)."""
    a, sim = attention(P, pre + '.atten', F.layer_norm(x, x.shape[1:]), p, lr_mul)
    if (pre + '.proj.weight') in P:
        x = equal_linear(x, P[pre + '.proj.weight'], P[pre + '.proj.bias'], lr_mul)
    x = x + a
    h = equal_linear(F.layer_norm(x, x.shape[1:]), P[pre + '.mlp.0.weight'], P[pre + '.mlp.0.bias'], lr_mul)
    h = F.gelu(h)
    x = x + equal_linear(h, P[pre + '.mlp.2.weight'], P[pre + '.mlp.2.bias'], lr_mul)
    return x, sim


def token_mapping(P, pre, x, pixel_norm_dim, lr_mlp, n_map=16):
    """model_spatial_query.py:626-646: PixelNorm, then token i through its OWN single
    EqualLinear(512,512,lr_mul)+fused_lrelu; tokens >= n_map stay zero."""
    x = pixel_norm(x, pixel_norm_dim)
    cols = []
    for i in range(x.shape[2]):
        if i < n_map:
            cols.append(equal_linear(x[:, :, i], P[f'{pre}.{i + 1}.weight'], P[f'{pre}.{i + 1}.bias'],
                                     lr_mlp, activation=True))
        else:
            cols.append(torch.zeros_like(x[:, :, i]))
    return torch.stack(cols, dim=2)


def generator_latent(P, z, p, n_trans=8, lr_mlp=0.01, pixel_norm_dim=1, num_region=1,
                     use_spatial_mapping=True, use_style_mapping=True, trans_interact=True):
    """Mapping + interaction part of Generator.forward, model_spatial_query.py:626-686.
    Returns (latent[B,token,512], spatialcode[B,16,512], stylecode[B,512,16], spatial[B,512,16], sims)."""
    n_map = int(16 / num_region)
    sp = token_mapping(P, 'spatial_mapping_network', p, pixel_norm_dim, lr_mlp, n_map) if use_spatial_mapping else p
    st = token_mapping(P, 'style_mapping_network', z, pixel_norm_dim, lr_mlp, n_map) if use_style_mapping else z
    stylecode, spatialcode = st.permute(0, 2, 1), sp.permute(0, 2, 1)
    sims = []
    if trans_interact:
        eye = P['token_spatial'].to(z.dtype).unsqueeze(0).expand(z.shape[0], -1, -1)
        x, s0 = attention_block(P, 'interact.0', torch.cat([stylecode, eye], 2),
                                torch.cat([spatialcode, eye], 2), lr_mlp)
        sims.append(s0)
        for i in range(1, n_trans):
            x, si = attention_block(P, f'interact.{i}', x, spatialcode, lr_mlp)
            sims.append(si)
    else:
        x = stylecode
    latent = equal_linear(x.permute(0, 2, 1), P['adjust_style.weight'], P['adjust_style.bias']).permute(0, 2, 1)
    return latent, spatialcode, st, sp, sims


def synthesis(P, latent, spatialcode, size, inject_noise=False, noise=None, taps=None):
    """Synthesis part of Generator.forward, model_spatial_query.py:696-716.  `taps`, if a
    dict, receives every intermediate activation (for per-layer parity checks)."""
    B = spatialcode.shape[0]
    log_size = int(math.log2(size))
    n_layers = (log_size - 2) * 2 + 1
    noise = noise if noise is not None else [None] * n_layers
    out = spatialcode.permute(0, 2, 1).reshape(B, 512, 4, 4)                 # :699
    out = styled_conv(P, 'conv1', out, latent[:, 0], noise=noise[0], inject_noise=inject_noise)
    skip = to_rgb(P, 'to_rgb1', out, latent[:, 1])
    if taps is not None:
        taps['conv1'] = out
        taps['to_rgb1'] = skip
    i = 1
    for j in range(log_size - 2):
        out = styled_conv(P, f'convs.{2 * j}', out, latent[:, i], True, noise[1 + 2 * j], inject_noise)
        if taps is not None:
            taps[f'convs.{2 * j}'] = out
        out = styled_conv(P, f'convs.{2 * j + 1}', out, latent[:, i + 1], False, noise[2 + 2 * j], inject_noise)
        skip = to_rgb(P, f'to_rgbs.{j}', out, latent[:, i + 2], skip)
        if taps is not None:
            taps[f'convs.{2 * j + 1}'] = out
            taps[f'to_rgbs.{j}'] = skip
        i += 2
    return skip


def generator_forward(P, z, p, size, n_trans=8, lr_mlp=0.01, pixel_norm_dim=1, input_is_latent=False,
                      taps=None, **kw):
    """Generator.forward, model_spatial_query.py:591-728; returns (image, latent, spatialcode[B,16,512])."""
    if input_is_latent:                                                      # :618-621
        sp = token_mapping(P, 'spatial_mapping_network', p, pixel_norm_dim, lr_mlp)
        latent, spatialcode = z, sp.permute(0, 2, 1)
    else:
        latent, spatialcode, _, _, _ = generator_latent(P, z, p, n_trans, lr_mlp, pixel_norm_dim, **kw)
    return synthesis(P, latent, spatialcode, size, taps=taps), latent, spatialcode


# ------------------------------------------------------------------- discriminator

def conv_layer(P, pre, x, k, downsample=False, bias=True, activate=True):
    """ConvLayer, model_spatial_query.py:731-777: [Blur] -> EqualConv2d -> FusedLeakyReLU|ScaledLeakyReLU.
    Sequential indices: blur=0 (if downsample), conv, act."""
    ci = 1 if downsample else 0
    w = P[f'{pre}.{ci}.weight']
    if downsample:
        pp = 2 + (k - 1)
        x = upfirdn2d(x, fir_kernel((1, 3, 3, 1)).to(x), pad=((pp + 1) // 2, pp // 2))
    scale = 1.0 / math.sqrt(w.shape[1] * k * k)                              # :165
    b = P.get(f'{pre}.{ci}.bias') if (bias and not activate) else None
    y = F.conv2d(x, w * scale, b, stride=2 if downsample else 1, padding=0 if downsample else k // 2)
    if activate:
        if bias:
            y = fused_leaky_relu(y, P[f'{pre}.{ci + 1}.bias'])
        else:
            y = fused_leaky_relu(y, None)                                    # ScaledLeakyReLU, :229-238
    return y


def discriminator_forward(P, img, size):
    """Discriminator.forward, model_spatial_query.py:841-859."""
    log_size = int(math.log2(size))
    out = conv_layer(P, 'convs.0', img, 1)
    for n, _ in enumerate(range(log_size, 2, -1), start=1):                  # ResBlock :791-798
        h = conv_layer(P, f'convs.{n}.conv1', out, 3)
        h = conv_layer(P, f'convs.{n}.conv2', h, 3, downsample=True)
        s = conv_layer(P, f'convs.{n}.skip', out, 1, downsample=True, bias=False, activate=False)
        out = (h + s) / math.sqrt(2)
    B, C, H, W = out.shape
    group = min(B, 4)
    sd = out.reshape(group, -1, 1, C, H, W)
    sd = torch.sqrt(sd.var(0, unbiased=False) + 1e-8).mean([2, 3, 4], keepdim=True).squeeze(2)
    out = torch.cat([out, sd.repeat(group, 1, H, W)], 1)
    out = conv_layer(P, 'final_conv', out, 3)
    out = equal_linear(out.reshape(B, -1), P['final_linear.0.weight'], P['final_linear.0.bias'], activation=True)
    return equal_linear(out, P['final_linear.1.weight'], P['final_linear.1.bias'])


# ------------------------------------------------------------------- losses (train_spatial_query.py:70-105)

def d_logistic_loss(real_pred, fake_pred):
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def g_nonsaturating_loss(fake_pred):
    return F.softplus(-fake_pred).mean()


def d_r1_loss(real_pred, real_img):
    g, = torch.autograd.grad(real_pred.sum(), real_img, create_graph=True)
    return g.pow(2).reshape(g.shape[0], -1).sum(1).mean()


def g_path_regularize(fake_img, latents, mean_path_length, noise, decay=0.01):
    """train_spatial_query.py:92-105 with the randn_like noise passed in (already / sqrt(H*W))."""
    g, = torch.autograd.grad((fake_img * noise).sum(), latents, create_graph=True)
    lengths = torch.sqrt(g.pow(2).sum(2).mean(1))
    mean = mean_path_length + decay * (lengths.mean() - mean_path_length)
    return (lengths - mean).pow(2).mean(), mean.detach(), lengths
