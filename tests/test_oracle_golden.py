"""The oracle (oracle/te_oracle.py) against the golden vectors generated FROM THE REFERENCE
(oracle/gen_golden.py).  CPU only; this is what pins the oracle."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import te_oracle as O
from oracle.gen_golden import UPFIRDN_CASES
from transeditor_amd import synth

TOL = 2e-5


@pytest.mark.parametrize('case', UPFIRDN_CASES, ids=[c[0] for c in UPFIRDN_CASES])
def test_upfirdn2d(golden, case):
    g = golden('upfirdn2d')
    name, _, _, _, up, down, pad = case
    x = g[f'{name}.x'].clone().requires_grad_(True)
    y = O.upfirdn2d(x, g[f'{name}.k'], up, down, pad)
    assert y.shape == g[f'{name}.y'].shape                      # integer output-size formula: exact
    assert rel_err(y, g[f'{name}.y']) < TOL
    gx, = torch.autograd.grad((y * g[f'{name}.wy']).sum(), x)
    assert rel_err(gx, g[f'{name}.gx']) < TOL


def _upfirdn2d_loops(x, k, up, down, pad):
    """Literal per-output index arithmetic of the reference CUDA kernel (upfirdn2d_kernel.cu:85-129,
    floor_div :18-26), pure Python loops, small inputs only."""
    B, C, H, W = x.shape
    kh, kw = k.shape
    oh = (H * up + pad[0] + pad[1] - kh) // down + 1
    ow = (W * up + pad[0] + pad[1] - kw) // down + 1
    out = torch.zeros(B, C, oh, ow, dtype=x.dtype)
    kf = torch.flip(k, [0, 1])
    for oy in range(oh):
        mid_y = oy * down + up - 1 - pad[0]
        iy0 = math.floor(mid_y / up)
        jy0 = (iy0 + 1) * up - mid_y - 1
        for ox in range(ow):
            mid_x = ox * down + up - 1 - pad[0]
            ix0 = math.floor(mid_x / up)
            jx0 = (ix0 + 1) * up - mid_x - 1
            acc = torch.zeros(B, C, dtype=x.dtype)
            for a in range((kh - jy0 + up - 1) // up):
                for b in range((kw - jx0 + up - 1) // up):
                    iy, ix = iy0 + a, ix0 + b
                    if 0 <= iy < H and 0 <= ix < W:
                        acc += x[:, :, iy, ix] * kf[jy0 + a * up, jx0 + b * up]
            out[:, :, oy, ox] = acc
    return out


@pytest.mark.parametrize('case', UPFIRDN_CASES, ids=[c[0] for c in UPFIRDN_CASES])
def test_upfirdn2d_index_arithmetic(golden, case):
    g = golden('upfirdn2d')
    name, _, _, _, up, down, pad = case
    y = _upfirdn2d_loops(g[f'{name}.x'].double(), g[f'{name}.k'].double(), up, down, pad)
    assert y.shape == g[f'{name}.y'].shape
    assert rel_err(y, g[f'{name}.y']) < TOL


@pytest.mark.parametrize('name', ['2d', '4d', '3d'])
def test_fused_leaky_relu(golden, name):
    g = golden('fused_leaky_relu')
    x = g[f'{name}.x'].clone().requires_grad_(True)
    b = g[f'{name}.b'].clone().requires_grad_(True)
    wy = g[f'{name}.wy'].clone().requires_grad_(True)
    y = O.fused_leaky_relu(x, b)
    assert rel_err(y, g[f'{name}.y']) < TOL
    gx, gb = torch.autograd.grad((y * wy).sum(), (x, b), create_graph=True)
    assert rel_err(gx, g[f'{name}.gx']) < TOL and rel_err(gb, g[f'{name}.gb']) < TOL
    ggy, = torch.autograd.grad((gx * g[f'{name}.u']).sum() + (gb * g[f'{name}.ub']).sum(), wy)
    assert rel_err(ggy, g[f'{name}.ggy']) < TOL


@pytest.mark.parametrize('name', ['plain3', 'up3', 'rgb1', 'plain3_wide', 'up3_wide'])
def test_modulated_conv2d(golden, name):
    g = golden('modulated_conv2d')
    demod, upsmp = bool(g[f'{name}.cfg'][0]), bool(g[f'{name}.cfg'][1])
    x = g[f'{name}.x'].clone().requires_grad_(True)
    s = g[f'{name}.s'].clone().requires_grad_(True)
    w, mw, mb = (g[f'{name}.{k}'].clone().requires_grad_(True) for k in ('weight', 'mod_w', 'mod_b'))
    y = O.modulated_conv2d(x, s, w, mw, mb, demod, upsmp)
    assert rel_err(y, g[f'{name}.y']) < TOL
    gr = torch.autograd.grad((y * g[f'{name}.wy']).sum(), (x, s, w, mw, mb), create_graph=True)
    for got, key in zip(gr, ('gx', 'gs', 'gw', 'gmw', 'gmb')):
        assert rel_err(got, g[f'{name}.{key}']) < TOL, key
    pl = gr[1].pow(2).sum()
    assert abs(float(pl) - float(g[f'{name}.pl'])) / float(g[f'{name}.pl']) < TOL
    g2 = torch.autograd.grad(pl, (x, w, mw, mb), allow_unused=True)
    for got, key in zip(g2, ('pl_gx', 'pl_gw', 'pl_gmw', 'pl_gmb')):
        got = torch.zeros_like(g[f'{name}.{key}']) if got is None else got
        assert rel_err(got, g[f'{name}.{key}']) < TOL or float(g[f'{name}.{key}'].abs().max()) == 0, key


@pytest.mark.parametrize('name', ['down3', 'down3_wide', 'down1', 'down3_nodemod'])
def test_modulated_conv2d_downsample(golden, name):
    g = golden('modulated_conv2d_down')
    x = g[f'{name}.x'].clone().requires_grad_(True)
    s = g[f'{name}.s'].clone().requires_grad_(True)
    w, mw, mb = (g[f'{name}.{k}'].clone().requires_grad_(True) for k in ('weight', 'mod_w', 'mod_b'))
    y = O.modulated_conv2d(x, s, w, mw, mb, bool(g[f'{name}.cfg'][0]), downsample=True)
    assert rel_err(y, g[f'{name}.y']) < TOL
    gr = torch.autograd.grad((y * g[f'{name}.wy']).sum(), (x, s, w, mw, mb))
    for got, key in zip(gr, ('gx', 'gs', 'gw', 'gmw', 'gmb')):
        assert rel_err(got, g[f'{name}.{key}']) < TOL, key


def attention_block_params(name, cin, cp, device='cpu'):
    """Weights of the attention-block fixture (same rule as oracle/gen_golden.py)."""
    shapes = {'atten.q_transform.weight': (128, cp), 'atten.q_transform.bias': (128,),
              'atten.k_transform.weight': (128, cin), 'atten.k_transform.bias': (128,),
              'atten.v_transform.weight': (128, cin), 'atten.v_transform.bias': (128,),
              'atten.proj.weight': (512, 128), 'atten.proj.bias': (512,),
              'mlp.0.weight': (512, 512), 'mlp.0.bias': (512,), 'mlp.2.weight': (512, 512), 'mlp.2.bias': (512,)}
    if cin != 512:
        shapes.update({'proj.weight': (512, cin), 'proj.bias': (512,)})
    return {k: (synth.normal(v, f'ab.{name}.{k}') * (10.0 if 'bias' in k else 100.0)).to(device)
            for k, v in shapes.items()}


@pytest.mark.parametrize('name,cin', [('b0_528', 528), ('b_512', 512)])
def test_attention_block(golden, name, cin):
    g = golden('attention_block')
    P = {'b.' + k: v for k, v in attention_block_params(name, cin, cin).items()}
    x = g[f'{name}.x'].clone().requires_grad_(True)
    p = g[f'{name}.p'].clone().requires_grad_(True)
    y, sim = O.attention_block(P, 'b', x, p, 0.01)
    assert rel_err(y, g[f'{name}.y']) < TOL and rel_err(sim, g[f'{name}.sim']) < TOL
    gx, gp = torch.autograd.grad((y * g[f'{name}.wy']).sum(), (x, p))
    assert rel_err(gx, g[f'{name}.gx']) < TOL and rel_err(gp, g[f'{name}.gp']) < TOL


def generator_state(size, seed, device='cpu'):
    """Parameters of the Generator fixture of `size`: the product module is only used as a source of the
    state_dict schema (constructible on CPU); values come from the deterministic PRNG."""
    from transeditor_amd.model_spatial_query import Generator
    token = 2 * (int(math.log2(size)) - 1)
    g = Generator(size, 512, 512, token, n_trans=8, pixel_norm_op_dim=1)
    sd = g.state_dict()
    synth.fill_state_dict(sd, seed)
    return g, {k: v.to(device) for k, v in sd.items()}


def test_generator64_config1(golden):
    """BASELINE config 1: Generator forward, 64x64, batch 4, num_trans 8 (+ grads wrt latents)."""
    g = golden('generator64_b4')
    _, P = generator_state(64, 0)
    z, p = synth.latents(4, 1000)
    z.requires_grad_(True)
    p.requires_grad_(True)
    taps = {}
    img, latent, spatial = O.generator_forward(P, z, p, 64, taps=taps)
    assert rel_err(img, g['image']) < TOL and rel_err(latent, g['latent']) < TOL
    assert rel_err(spatial.permute(0, 2, 1), g['spatialcode']) < TOL
    for k, t in taps.items():
        assert rel_err(torch.stack([t.mean(), t.abs().max()]), g[f'layer.{k}.stats']) < 1e-4, k
    wimg = synth.normal(tuple(img.shape), 'wimg.64')
    gz, gp = torch.autograd.grad((img * wimg).sum() / img.numel(), (z, p))
    assert rel_err(gz, g['gz']) < 1e-4 and rel_err(gp, g['gp']) < 1e-4


def test_generator_flags(golden):
    g = golden('generator64_flags')
    _, P = generator_state(64, 0)
    zz, pp = synth.latents(2, 1001)
    latent, spatial, st, sp, _ = O.generator_latent(P, zz, pp)
    assert rel_err(sp, g['mapped_p']) < TOL and rel_err(st, g['mapped_z']) < TOL
    assert rel_err(latent, g['style_latent']) < TOL
    img = O.synthesis(P, latent, spatial, 64)
    assert rel_err(img, g['img_default']) < TOL
    lat2, spc2, _, _, _ = O.generator_latent(P, g['mapped_z'], g['mapped_p'], use_spatial_mapping=False,
                                             use_style_mapping=False)
    assert rel_err(O.synthesis(P, lat2, spc2, 64), g['img_nomap']) < TOL
    img3, _, _ = O.generator_forward(P, g['style_latent'], pp, 64, input_is_latent=True)
    assert rel_err(img3, g['img_from_latent']) < TOL


@pytest.mark.parametrize('size', [8, 32])
def test_generator_small_and_path_length(golden, size):
    g = golden(f'generator{size}_b2')
    _, P = generator_state(size, size)
    z, p = synth.latents(2, 2000 + size)
    if size == 32:
        P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in P.items()}
    img, latent, _ = O.generator_forward(P, z, p, size)
    assert rel_err(img, g['image']) < TOL and rel_err(latent, g['latent']) < TOL
    if size == 32:
        noise = synth.normal(tuple(img.shape), 'pl.noise') / math.sqrt(size * size)
        pen, _, lengths = O.g_path_regularize(img, latent, 0.0, noise)
        assert rel_err(lengths, g['path_lengths']) < TOL
        names = [str(n) for n in g['pl_grad_names']]
        gs = torch.autograd.grad(pen, [P[n] for n in names], allow_unused=True)
        for n, got, want in zip(names, gs, g['pl_grad_norms']):
            if want > 1e-8:
                assert abs(float(got.double().norm()) - want) / want < 1e-4, n


def test_discriminator(golden):
    from transeditor_amd.model_spatial_query import Discriminator
    g = golden('discriminator64_b4')
    sd = Discriminator(64).state_dict()
    synth.fill_state_dict(sd, 5)
    img = synth.normal((4, 3, 64, 64), 'd.img').clamp(-1, 1).requires_grad_(True)
    fake = synth.normal((4, 3, 64, 64), 'd.fake').clamp(-1, 1)
    pred = O.discriminator_forward(sd, img, 64)
    fpred = O.discriminator_forward(sd, fake, 64)
    assert rel_err(pred, g['pred']) < TOL and rel_err(fpred, g['fake_pred']) < TOL
    assert abs(float(O.d_r1_loss(pred, img)) - float(g['r1'])) / float(g['r1']) < 1e-4
    assert abs(float(O.d_logistic_loss(pred, fpred)) - float(g['d_loss'])) < 1e-4
    assert abs(float(O.g_nonsaturating_loss(fpred)) - float(g['g_loss'])) < 1e-4
