"""TransEditor generator / discriminator on MI355X-native kernels.

Drop-in for the reference's ``model_spatial_query.py``: identical class names, constructor
arguments, ``forward`` keyword surface, return conventions and ``state_dict`` schema (SURVEY
§8b), so ``train_spatial_query.py`` / ``test_spatial_query.py`` and published ``g_ema``
checkpoints sit on top unchanged.  What differs is underneath:

* ``ModulatedConv2d`` never materialises per-sample weights: style modulation and demodulation
  are per-channel scalings fused into one fp32-MFMA implicit-GEMM kernel (``op/modconv.py``);
  the upsampling layers run the stride-2 transposed convolution polyphase (no zero insertion).
* ``StyledConv`` fuses bias + leaky-ReLU into the convolution epilogue (plain layers) or into the
  blur FIR kernel (upsampling layers).
* the 2 x 16 per-token mapping linears are two batched GEMMs + one fused bias/lrelu launch
  instead of 64 launches and 32 slice copies (reference model_spatial_query.py:626-646).
* the attention core runs QK^T and sim.V on fp32 MFMA (``op/attention.py``).

There is no CPU path: the ops raise if libte_hip.so is missing or tensors are not on the GPU.
File:line citations refer to the reference's model_spatial_query.py.
"""
import math
import os

import torch
from torch import nn
from torch.nn import functional as F

from .op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d
from .op.attention import attention_core
from .op.fir_act import blur_bias_act
from .op.layernorm import pixel_norm, sample_layer_norm
from .op.linear import linear_fused, shared_input_linears
from .op.modconv import latent_producer_section, modconv, _STATE as _modconv_state
from .op.resblock import resblock
from .op import modulation, styled_rgb
from .op.stddev import minibatch_stddev
from .op.style import demod
from .op.token_mlp import token_mlp

CHANNELS = lambda cm: {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm,
                       512: 32 * cm, 1024: 16 * cm}     # :473-483


class PixelNorm(nn.Module):
    def __init__(self, pixel_norm_op_dim):
        super().__init__()
        self.pixel_norm_op_dim = pixel_norm_op_dim

    def forward(self, input):                                                        # :80-81
        return pixel_norm(input, self.pixel_norm_op_dim)


def make_kernel(k):                                                                  # :84-92
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum()


def _resample_pad(taps, factor, up):
    p = taps - factor
    return ((p + 1) // 2 + factor - 1, p // 2) if up else ((p + 1) // 2, p // 2)


class Upsample(nn.Module):                                                           # :95-113
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel) * factor ** 2)
        self.pad = _resample_pad(self.kernel.shape[0], factor, True)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):                                                         # :116-134
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel))
        self.pad = _resample_pad(self.kernel.shape[0], factor, False)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):                                                               # :137-153
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * upsample_factor ** 2
        self.register_buffer('kernel', kernel)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualConv2d(nn.Module):                                                        # :156-191
    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding = stride, padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input, act_bias=None, presampled=False, gain=1.0):
        """`act_bias` (extension used by ConvLayer): fuse '+ act_bias' and the scaled leaky-ReLU that follows.
        `presampled` (ConvLayer, 1x1 stride 2 only): the caller already kept every second row / column.
        `gain` (ResBlock): constant output factor folded into the weights (no activation) or the leaky-ReLU gain.
        The four configurations the discriminator uses run on the MI355X convolution kernels (same family as the
        generator, no modulation); anything else raises (no library fallback)."""
        w, ws = self.weight, self.scale
        k, cfg = self.weight.shape[2], (self.weight.shape[2], self.stride, self.padding)
        act = act_bias is not None
        bias = act_bias if act else self.bias
        if gain != 1.0:
            if act:
                act = math.sqrt(2) * gain                 # lrelu(conv + b) * sqrt(2) * gain: the gain of the fused activation
            elif bias is None:
                ws = ws * gain
            else:
                return self.forward(input, act_bias, presampled) * gain
        if input.is_cuda and input.dtype == torch.float32:
            if cfg == (3, 1, 1):
                return modconv(input, w, None, None, bias, act, '3x3', ws)
            if cfg == (1, 1, 0):
                return modconv(input, w, None, None, bias, act, '1x1', ws)
            if cfg == (3, 2, 0) and input.shape[2] % 2 == 1 and input.shape[3] % 2 == 1:
                return modconv(input, w, None, None, bias, act, 'down', ws)
            if cfg == (1, 2, 0):
                x = input if presampled else input[:, :, ::2, ::2].contiguous()
                return modconv(x, w, None, None, bias, act, '1x1', ws)
        # no library-convolution fallback: a configuration without a kernel fails loudly, like every other op of this path
        raise RuntimeError(f'te_hip: EqualConv2d(kernel {k}, stride {self.stride}, padding {self.padding}) on {input.dtype} '
                           f'{input.device} {tuple(input.shape)} has no MI355X kernel (covered: 3x3 s1 p1, 1x1 s1 p0, 3x3 s2 p0 on '
                           f'odd sizes, 1x1 s2 p0; fp32 on the GPU; there is no CPU / library path)')

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]},'
                f' {self.weight.shape[2]}, stride={self.stride}, padding={self.padding})')


class EqualLinear(nn.Module):                                                        # :194-226
    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1.0, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input, act=None, residual=None):
        """One fused launch (op/linear.py): act(scale * x W^T + lr_mul * b) + residual.  `act` / `residual` are
        extensions used by the attention blocks ('gelu', skip connection); activation='fused_lrelu' is the reference's."""
        if self.activation:
            if self.bias is None:       # the reference fails the same way (bias * lr_mul on None)
                raise TypeError("unsupported operand type(s) for *: 'NoneType' and 'float'")
            return linear_fused(input, self.weight, self.bias, self.scale, self.lr_mul, 'lrelu')
        return linear_fused(input, self.weight, self.bias, self.scale, self.lr_mul, act, residual)

    def __repr__(self):
        return f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})'


class ScaledLeakyReLU(nn.Module):                                                    # :229-238
    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, negative_slope=self.negative_slope) * math.sqrt(2)


class ModulatedConv2d(nn.Module):                                                    # :241-337
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if kernel_size not in (1, 3) or (upsample and kernel_size != 3) or (upsample and downsample):
            raise NotImplementedError('MI355X kernels exist for 3x3, 3x3 upsample, 3x3 / 1x1 downsample and 1x1 modulated '
                                      'convolutions')
        self.eps = 1e-8
        self.kernel_size, self.in_channel, self.out_channel = kernel_size, in_channel, out_channel
        self.upsample, self.downsample = upsample, downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        if downsample:                                                               # :270-276
            p = (len(blur_kernel) - 2) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, '
                f'upsample={self.upsample}, downsample={self.downsample})')

    @property
    def kind(self):
        if self.downsample and self.kernel_size == 3:
            return 'down'
        return 'up' if self.upsample else ('3x3' if self.kernel_size == 3 else '1x1')

    def scales(self, style):
        """(w, s, d): shared UNSCALED weight [Co,Ci,k,k] (the kernels apply self.scale), style scale [B,Ci],
        demodulation [B,Co] or None.  d[b,co] = rsqrt(sum_ci s^2 * sum_k (scale w)^2 + eps)  ==  :300-304 without the B
        weight copies, one launch (op/style.py)."""
        s = self.modulation(style)
        w = self.weight.view(self.weight.shape[1:])
        d = demod(w, s, self.scale, self.eps) if self.demodulate else None
        return w, s, d

    def forward(self, input, style, bias=None, act=False, s=None):
        """`bias`/`act` (extensions used by StyledConv / ToRGB) fuse '+ bias' and the scaled leaky-ReLU.  `s`: the style scale
        `self.modulation(style)` when the caller already has it (Generator batches all modulations of a pass, op/modulation.py)."""
        if s is None:
            s = self.modulation(style)
        w = self.weight.view(self.weight.shape[1:])
        eps = self.eps if self.demodulate else None          # demodulation is computed inside the modconv node
        if self.downsample:                                  # :323-329: blur, then the stride-2 convolution without padding
            if self.kernel_size == 1:                        # a 1x1 stride-2 conv reads every second blurred pixel only
                input = upfirdn2d(input, self.blur.kernel, down=2, pad=self.blur.pad)
                return modconv(input, w, s, None, bias, act, '1x1', self.scale, demod_eps=eps)
            input = self.blur(input)                         # (H+1)x(W+1); the strided kernel takes (2h+1)x(2w+1):
            if input.shape[2] % 2 == 0 or input.shape[3] % 2 == 0:       # a last row / column no output tap reaches
                input = input[:, :, :input.shape[2] - 1 + input.shape[2] % 2, :input.shape[3] - 1 + input.shape[3] % 2]
            return modconv(input, w, s, None, bias, act, 'down', self.scale, demod_eps=eps)
        if not self.upsample:
            return modconv(input, w, s, None, bias, act, self.kind, self.scale, demod_eps=eps)
        out = modconv(input, w, s, None, None, False, 'up', self.scale, demod_eps=eps)
        if act:
            return blur_bias_act(out, self.blur.kernel, bias, self.blur.pad)
        out = self.blur(out)
        return out if bias is None else out + bias.view(1, -1, 1, 1)


class NoiseInjection(nn.Module):                                                     # :340-351
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            batch, _, height, width = image.shape
            noise = image.new_empty(batch, 1, height, width).normal_()
        return image + self.weight * noise


class ConstantInput(nn.Module):                                                      # :354-364 (unused by G)
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):                                                         # :367-403
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True, layer_noise_injection=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.layer_noise_injection = layer_noise_injection
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None, s=None):
        if self.layer_noise_injection:
            return self.activate(self.noise(self.conv(input, style, s=s), noise=noise))
        a = self.activate
        if a.bias is not None and a.negative_slope == 0.2 and abs(a.scale - 2 ** 0.5) < 1e-12:
            return self.conv(input, style, bias=a.bias, act=True, s=s)       # bias + lrelu fused into the conv / blur kernel
        return a(self.conv(input, style, s=s))


class ToRGB(nn.Module):                                                              # :406-425
    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None, s=None):
        out = self.conv(input, style, bias=self.bias.view(3), s=s)
        if skip is not None:
            out = out + self.upsample(skip)
        return out


class Generator(nn.Module):                                                          # :428-728
    def __init__(self, size, style_dim, param_dim, token_dim, channel_multiplier=2, blur_kernel=[1, 3, 3, 1],
                 lr_mlp=0.01, layer_noise_injection=False, use_spatial_mapping=True, num_region=1, n_trans=4,
                 pixel_norm_op_dim=2, no_trans=False):
        super().__init__()
        self.size, self.lr_mlp, self.n_trans, self.no_trans = size, lr_mlp, n_trans, no_trans
        self.style_dim, self.param_dim, self.token_dim = style_dim, param_dim, token_dim
        self.layer_noise_injection = layer_noise_injection
        self.style = None
        self.use_spatial_mapping, self.num_region = use_spatial_mapping, num_region
        self.num_spatial_mapping = int(16 / num_region)
        self.num_style_mapping = self.num_spatial_mapping
        self.pixel_norm_op_dim = pixel_norm_op_dim

        if self.use_spatial_mapping:
            self.spatial_mapping_network = self.spatial_mapping()
        self.style_mapping_network = self.style_mapping()
        self.channels = CHANNELS(channel_multiplier)
        self.adjust_style = EqualLinear(in_dim=16, out_dim=self.token_dim)

        ch4 = self.channels[4]
        self.conv1 = StyledConv(ch4, ch4, 3, style_dim, blur_kernel=blur_kernel,
                                layer_noise_injection=layer_noise_injection)
        self.to_rgb1 = ToRGB(ch4, style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs, self.upsamples, self.to_rgbs = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.noises = nn.Module()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f'noise_{layer_idx}', torch.randn(1, 1, 2 ** res, 2 ** res))
        in_channel = ch4
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True, blur_kernel=blur_kernel,
                                         layer_noise_injection=layer_noise_injection))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel,
                                         layer_noise_injection=layer_noise_injection))
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2
        self.register_buffer('token', torch.eye(self.token_dim))
        self.register_buffer('token_spatial', torch.eye(16))
        self.trans_interact = not self.no_trans
        if self.trans_interact:
            self.interact = self.interaction_network()

    def _mapping(self, n):
        layers = [PixelNorm(self.pixel_norm_op_dim)]
        layers += [EqualLinear(self.style_dim, self.style_dim, lr_mul=self.lr_mlp, activation='fused_lrelu')
                   for _ in range(n)]
        return nn.Sequential(*layers)

    def style_mapping(self):                                                         # :547-555
        return self._mapping(self.num_style_mapping)

    def spatial_mapping(self):                                                       # :558-566
        return self._mapping(self.num_spatial_mapping)

    def interaction_network(self):                                                   # :569-577
        blocks = [AttentionBlock(self.style_dim + 16, self.param_dim + 16, self.style_dim, lr_mul=self.lr_mlp)]
        blocks += [AttentionBlock(self.style_dim, self.param_dim, self.style_dim, lr_mul=self.lr_mlp)
                   for _ in range(1, self.n_trans)]
        return nn.Sequential(*blocks)

    def make_noise(self):                                                            # :579-588
        device = self.token.device
        noises = [torch.randn(1, 1, 4, 4, device=device)]
        for i in range(3, self.log_size + 1):
            noises += [torch.randn(1, 1, 2 ** i, 2 ** i, device=device) for _ in range(2)]
        return noises

    def _conv_rgb(self, conv, to_rgb, x, style_c, style_r, skip, noise, s_c=None, s_r=None):
        """`out = conv(x, style_c); skip = to_rgb(out, style_r, skip)` (:702-714).  In a training forward the plain StyledConv
        and the ToRGB reading it run as one autograd node (op/styled_rgb.py): the ToRGB data gradient is folded into the
        convolution's activation-gradient pass instead of a pass + a gradient-accumulation add of its own."""
        m, a = conv.conv, conv.activate
        w = m.weight.view(m.weight.shape[1:])
        wr = to_rgb.conv.weight.view(to_rgb.conv.weight.shape[1:])
        if (torch.is_grad_enabled() and not conv.layer_noise_injection and not _modconv_state['second_order']
                and not _modconv_state.frozen_on and m.kind == '3x3' and m.demodulate and not to_rgb.conv.demodulate
                and to_rgb.conv.kind == '1x1' and a.bias is not None and a.negative_slope == 0.2
                and abs(a.scale - 2 ** 0.5) < 1e-12 and styled_rgb.supported(x, w, wr)):
            out, rgb = styled_rgb.styled_conv_rgb(x, w, m.modulation(style_c) if s_c is None else s_c, a.bias, wr,
                                                  to_rgb.conv.modulation(style_r) if s_r is None else s_r,
                                                  to_rgb.bias.view(3), m.scale, m.eps, to_rgb.conv.scale)
            return out, (rgb if skip is None else rgb + to_rgb.upsample(skip))
        out = conv(x, style_c, noise=noise, s=s_c)
        return out, to_rgb(out, style_r, skip, s=s_r)

    def _map_tokens(self, net, codes, n_map):
        """:626-646 — PixelNorm, then token i through its own EqualLinear + fused lrelu: all tokens of a network in one
        batched launch (op/token_mlp.py).  codes [B, D, C] -> [B, D, C] (tokens >= n_map stay 0)."""
        B, D, Cn = codes.shape
        x = net[0](codes)
        lins = [net[i + 1] for i in range(n_map)]
        y = token_mlp(x, [l.weight for l in lins], [l.bias for l in lins], lins[0].scale, lins[0].lr_mul)   # [B, T, D]
        out = y.permute(0, 2, 1)
        if n_map < Cn:
            out = torch.cat([out, out.new_zeros(B, D, Cn - n_map)], dim=2)
        return out

    def forward(self, style, op_param, return_latents=False, input_is_latent=False, noise=None, randomize_noise=True,
                return_style=False, return_p_latent=False, return_only_style=False, return_only_style_latent=False,
                return_only_mapped_p=False, return_only_mapped_z=False, use_spatial_mapping=True,
                use_style_mapping=True, trans_interact=True, return_mapped_codes=False):
        if self.no_trans:                                                            # :615-621
            trans_interact = False
        if input_is_latent:
            use_spatial_mapping, use_style_mapping, trans_interact = True, False, False

        # everything that PRODUCES the latent; under second_order(wrt='latent') it keeps its fused first-order nodes
        with latent_producer_section():
            spatialcode = (self._map_tokens(self.spatial_mapping_network, op_param, self.num_spatial_mapping)
                           if use_spatial_mapping else op_param)
            stylecode = (self._map_tokens(self.style_mapping_network, style, self.num_style_mapping)
                         if use_style_mapping else style)
            if return_mapped_codes:
                return stylecode, spatialcode
            if return_only_mapped_p:
                return spatialcode
            if return_only_mapped_z:
                return stylecode

            if noise is None:                                                        # :658-664
                noise = ([None] * self.num_layers if randomize_noise
                         else [getattr(self.noises, f'noise_{i}') for i in range(self.num_layers)])

            stylecode, spatialcode = stylecode.permute(0, 2, 1), spatialcode.permute(0, 2, 1)   # [B,16,512]
            x = None
            if trans_interact:                                                           # :670-679
                eye = self.token_spatial.unsqueeze(0).expand(stylecode.shape[0], -1, -1)
                z0, p0 = torch.cat([stylecode, eye], 2), torch.cat([spatialcode, eye], 2)
                x = self.interact[0](z0, p0)
                if self.n_trans > 1:          # P is never updated between blocks (:675-678): every later block's query projection
                    qs = shared_input_linears(spatialcode, [self.interact[i].atten.q_transform for i in range(1, self.n_trans)])
                    for i in range(1, self.n_trans):
                        x = self.interact[i](x, spatialcode, q=qs[i - 1])
            if self.no_trans:                                                            # :682-688
                latent = self.adjust_style(stylecode.permute(0, 2, 1)).permute(0, 2, 1)
            elif not input_is_latent:
                if x is None:
                    # same failure as the reference (:686 reads `x`, which :675 never assigned)
                    raise UnboundLocalError("local variable 'x' referenced before assignment (trans_interact=False on a "
                                            "generator built with no_trans=False, as in the reference)")
                latent = self.adjust_style(x.permute(0, 2, 1)).permute(0, 2, 1)
            else:
                latent = style
        if return_only_style_latent or return_only_style:
            return latent

        batch = spatialcode.shape[0]
        out = spatialcode.permute(0, 2, 1).reshape(batch, 512, 4, 4)                 # :699  P code IS the 4x4 map
        # per-layer styles latent[:, i]: one contiguous copy + one unbind, so the backward is a single stack instead of
        # a zero-fill + add of the whole latent per layer
        lat = latent.contiguous().unbind(1)
        # every style modulation of the pass up front, as a few batched launches (op/modulation.py); sm[j] = the scale of the
        # j-th modulated convolution in execution order, None when the batched form does not apply
        mods, index = [self.conv1.conv.modulation, self.to_rgb1.conv.modulation], [0, 1]
        i = 1
        for conv_up, conv, to_rgb in zip(self.convs[::2], self.convs[1::2], self.to_rgbs):
            mods += [conv_up.conv.modulation, conv.conv.modulation, to_rgb.conv.modulation]
            index += [i, i + 1, i + 2]
            i += 2
        if not _modconv_state['second_order'] and modulation.supported(latent, mods):
            sm = modulation.batched_modulation(latent.contiguous(), mods, index)
        else:
            sm = [None] * len(mods)
        out, skip = self._conv_rgb(self.conv1, self.to_rgb1, out, lat[0], lat[1], None, noise[0], sm[0], sm[1])
        i, j = 1, 2
        for conv_up, conv, n1, n2, to_rgb in zip(self.convs[::2], self.convs[1::2], noise[1::2], noise[2::2],
                                                 self.to_rgbs):
            out = conv_up(out, lat[i], noise=n1, s=sm[j])
            out, skip = self._conv_rgb(conv, to_rgb, out, lat[i + 1], lat[i + 2], skip, n2, sm[j + 1], sm[j + 2])
            i += 2
            j += 3
        image = skip

        if return_style:
            return image, latent
        if return_p_latent:
            return image, spatialcode
        if return_latents:
            return image, latent, None
        return image, None, None


class ConvLayer(nn.Sequential):                                                      # :731-777
    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=[1, 3, 3, 1], bias=True,
                 activate=True):
        layers = []
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
            stride, self.padding = 2, 0
        else:
            stride, self.padding = 1, kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride,
                                  bias=bias and not activate))
        if activate:
            layers.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*layers)

    def forward(self, input, gain=1.0):
        """Same sequence as nn.Sequential, with EqualConv2d -> FusedLeakyReLU fused into the conv epilogue.  `gain`
        (ResBlock) is a constant factor on the layer output, folded into its last convolution when that is the last op."""
        mods = list(self)
        i = 0
        pres = False
        last_conv = max((j for j, m in enumerate(mods) if isinstance(m, EqualConv2d)), default=-1)
        fold = gain != 1.0 and last_conv >= 0 and (last_conv == len(mods) - 1 or (
            last_conv == len(mods) - 2 and isinstance(mods[-1], FusedLeakyReLU) and mods[-1].bias is not None
            and mods[last_conv].bias is None and mods[-1].negative_slope == 0.2 and abs(mods[-1].scale - 2 ** 0.5) < 1e-12))
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if (isinstance(m, Blur) and isinstance(nxt, EqualConv2d) and nxt.weight.shape[2] == 1 and nxt.stride == 2
                    and nxt.padding == 0 and input.is_cuda and input.dtype == torch.float32):
                # skip branch of ResBlock: blur, then a 1x1 conv that reads every second pixel -> let the FIR pass
                # produce only those pixels (same taps and pads, down = 2): a quarter of the writes, no strided copy
                input = upfirdn2d(input, m.kernel, down=2, pad=m.pad)
                pres = True
                i += 1
                continue
            g_here = gain if (fold and i == last_conv) else 1.0
            if pres:
                input = m(input, presampled=True, gain=g_here)
                pres = False
                i += 1
                continue
            if (isinstance(m, EqualConv2d) and isinstance(nxt, FusedLeakyReLU) and nxt.bias is not None
                    and m.bias is None and nxt.negative_slope == 0.2 and abs(nxt.scale - 2 ** 0.5) < 1e-12):
                input = m(input, act_bias=nxt.bias, gain=g_here)
                i += 2
            elif isinstance(m, EqualConv2d):
                input = m(input, gain=g_here)
                i += 1
            else:
                input = m(input)
                i += 1
        return input if (fold or gain == 1.0) else input * gain


class ResBlock(nn.Module):                                                           # :780-798
    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3, blur_kernel=blur_kernel)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True, blur_kernel=blur_kernel)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, blur_kernel=blur_kernel, bias=False,
                              activate=False)

    def _standard(self):
        """the reference's structure (conv3x3+lrelu; blur, conv3x3 s2 + lrelu; blur, conv1x1 s2), which the fused node covers"""
        def act_ok(a):
            return (isinstance(a, FusedLeakyReLU) and a.bias is not None and a.negative_slope == 0.2
                    and abs(a.scale - 2 ** 0.5) < 1e-12)
        c1, c2, sk = list(self.conv1), list(self.conv2), list(self.skip)
        return (len(c1) == 2 and isinstance(c1[0], EqualConv2d) and c1[0].bias is None and act_ok(c1[1])
                and (c1[0].weight.shape[2], c1[0].stride, c1[0].padding) == (3, 1, 1)
                and len(c2) == 3 and isinstance(c2[0], Blur) and isinstance(c2[1], EqualConv2d) and c2[1].bias is None
                and act_ok(c2[2]) and (c2[1].weight.shape[2], c2[1].stride, c2[1].padding) == (3, 2, 0)
                and len(sk) == 2 and isinstance(sk[0], Blur) and isinstance(sk[1], EqualConv2d) and sk[1].bias is None
                and (sk[1].weight.shape[2], sk[1].stride, sk[1].padding) == (1, 2, 0)
                and tuple(c2[0].kernel.shape) == (4, 4) and tuple(sk[0].kernel.shape) == (4, 4))

    def fusable(self, input):
        return (input.is_cuda and input.dtype == torch.float32 and not _modconv_state['second_order'] and input.shape[2] >= 8
                and input.shape[3] >= 8 and input.shape[2] % 2 == 0 and input.shape[3] % 2 == 0 and self._standard())

    def forward(self, input, stem=None):
        """`stem` (extension used by Discriminator): the from-RGB ConvLayer(3, C, 1) in front of this block, run inside the
        block's autograd node; `input` is the image then."""
        # (conv2 + skip) / sqrt(2) (:796) with the constant folded into the two branches' last convolutions
        g = 1 / math.sqrt(2)
        if self.fusable(input):
            # the whole block as one autograd node (op/resblock.py): sums in convolution epilogues, conv1's activation
            # gradient in the adjoint blur
            c1, c2, sk = self.conv1, self.conv2, self.skip
            st = None if stem is None else (stem[0].weight, stem[1].bias, stem[0].scale)
            return resblock(input, c1[0].weight, c1[1].bias, c2[1].weight, c2[2].bias, sk[1].weight, c2[0].kernel, sk[0].kernel,
                            c1[0].scale, c2[1].scale, sk[1].scale, c2[0].pad, sk[0].pad, g, stem=st)
        if stem is not None:
            input = stem(input)
        return self.conv2(self.conv1(input), gain=g) + self.skip(input, gain=g)


class Discriminator(nn.Module):                                                      # :801-859
    def __init__(self, size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        channels = CHANNELS(channel_multiplier)
        convs = [ConvLayer(3, channels[size], 1)]
        in_channel = channels[size]
        for i in range(int(math.log(size, 2)), 2, -1):
            out_channel = channels[2 ** (i - 1)]
            convs.append(ResBlock(in_channel, out_channel, blur_kernel))
            in_channel = out_channel
        self.convs = nn.Sequential(*convs)
        self.stddev_group, self.stddev_feat = 4, 1
        self.final_conv = ConvLayer(in_channel + 1, channels[4], 3)
        self.final_linear = nn.Sequential(EqualLinear(channels[4] * 4 * 4, channels[4], activation='fused_lrelu'),
                                          EqualLinear(channels[4], 1))

    def _stem_fusable(self, input):
        """from-RGB layer + first ResBlock as one node (op/resblock.py): the standard ConvLayer(3, C, 1) followed by a standard block"""
        if len(self.convs) < 2 or not isinstance(self.convs[1], ResBlock) or not isinstance(self.convs[0], ConvLayer):
            return False
        st = list(self.convs[0])
        return (len(st) == 2 and isinstance(st[0], EqualConv2d) and st[0].bias is None
                and (st[0].weight.shape[1], st[0].weight.shape[2], st[0].stride, st[0].padding) == (3, 1, 1, 0)
                and isinstance(st[1], FusedLeakyReLU) and st[1].bias is not None and st[1].negative_slope == 0.2
                and abs(st[1].scale - 2 ** 0.5) < 1e-12 and input.shape[1] == 3 and st[0].weight.shape[0] <= 512
                and (input.shape[2] * input.shape[3]) % 4 == 0 and self.convs[1].fusable(input))

    def forward(self, input, chunks=1):
        """`chunks` (extension; 1 = the reference's forward): `input` holds that many independent minibatches of equal size
        laid end to end, and the minibatch-stddev statistic (:844-852) - the only place where the samples of a batch meet -
        is formed inside each of them:  D(cat([fake, real]), chunks=2) == cat([D(fake), D(real)]).  The discriminator step
        (train_spatial_query.py:190-192) runs its two passes as one this way: half the launches, and the 4x4 - 16x16 layers
        see 32 samples instead of 16."""
        if self._stem_fusable(input):
            out = self.convs[1](input, stem=self.convs[0])
            for m in list(self.convs)[2:]:
                out = m(out)
        else:
            out = self.convs(input)
        batch = out.shape[0]
        # :844-852 minibatch stddev + concat: one launch (op/stddev.py)
        out = minibatch_stddev(out, self.stddev_group, self.stddev_feat, second_order=_modconv_state['second_order'], chunks=chunks)
        out = self.final_conv(out)
        return self.final_linear(out.view(batch, -1))


class Attention(nn.Module):                                                          # :862-901
    def __init__(self, in_dim, param_dim, out_dim, lr_mul=1.0, groups=4, compress=4):
        assert out_dim % (groups * compress) == 0
        super().__init__()
        self.in_dim, self.param_dim, self.out_dim = in_dim, param_dim, out_dim
        self.compress, self.groups = compress, groups
        self.planes = out_dim // compress
        self.group_planes = self.planes // groups
        self.scale = self.planes ** -0.5          # NB: planes, not head dim (:873)
        self.q_transform = EqualLinear(param_dim, self.planes, lr_mul=lr_mul)
        self.k_transform = EqualLinear(in_dim, self.planes, lr_mul=lr_mul)
        self.v_transform = EqualLinear(in_dim, self.planes, lr_mul=lr_mul)
        self.proj = EqualLinear(self.planes, out_dim, lr_mul=lr_mul)

    def forward(self, attention, op_param, return_similarity=False, residual=None, q=None):
        """`q` (extension used by Generator): `self.q_transform(op_param)` when the caller already has it (the blocks after the
        first all project the same P code: one batched launch, op/linear.py::shared_input_linears)."""
        if q is None:
            q = self.q_transform(op_param)                 # [N, M, planes]; head g = channels g*gp..
        k, v = shared_input_linears(attention, [self.k_transform, self.v_transform])      # same input: one launch
        # softmax(scale q k^T) v per head; the reference's reshape(N, planes, L).permute(0,2,1) (:894, with
        # L == M) lands exactly on this token-major [N, M, planes] layout
        stacked, similarity = attention_core(q, k, v, self.scale, self.groups)
        output = self.proj(stacked, residual=residual)     # the block's skip connection rides in the epilogue
        return (output, similarity) if return_similarity else output


class AttentionBlock(nn.Module):                                                     # :904-936
    def __init__(self, in_dim, param_dim, out_dim, lr_mul=1.0, groups=4):
        super().__init__()
        self.in_dim, self.out_dim, self.param_dim = in_dim, out_dim, param_dim
        self.atten = Attention(in_dim, param_dim, out_dim, lr_mul=lr_mul, groups=groups)
        self.mlp = nn.Sequential(EqualLinear(out_dim, out_dim, lr_mul=lr_mul), nn.GELU(),
                                 EqualLinear(out_dim, out_dim, lr_mul=lr_mul))
        if out_dim != in_dim:
            self.proj = EqualLinear(in_dim, out_dim, lr_mul=lr_mul)

    def forward(self, x, op_param, return_similarity=False, q=None):
        skip = self.proj(x) if self.out_dim != self.in_dim else x
        a = self.atten(sample_layer_norm(x), op_param, return_similarity=return_similarity, residual=skip, q=q)
        x, similarity = a if return_similarity else (a, None)           # x = skip + attention (:926-930)
        h = self.mlp[0](sample_layer_norm(x), act='gelu')      # Linear + GELU, then Linear + skip (:932-934)
        x = self.mlp[2](h, residual=x)
        return (x, similarity) if return_similarity else x
