"""Deterministic synthetic parameters and latents (no checkpoints, no datasets, no network).

A counter-based integer PRNG (splitmix64 over `hash(key) + seed + index`) followed by
Box-Muller in float64.  The same numbers are regenerated bit-identically in the build
container (where the golden fixtures are produced from the imported reference) and on the
GPU box (where the HIP path is checked against them), so the 160 MB of generator weights
never have to be committed.

Initialisation statistics follow the reference constructors (model_spatial_query.py:162-164,
200-203, 282-286, 414): weights ~ N(0,1) (divided by lr_mul where EqualLinear does),
modulation bias ~ 1; unlike the reference, biases get a small random component so that
parity tests exercise every bias path.
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(text):
    h = 0xCBF29CE484222325
    for ch in text.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x):
    with np.errstate(over='ignore'):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def normal(shape, key, seed=0):
    """Standard normal fp32 tensor, a pure function of (shape, key, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((_fnv1a(key) + 0x632BE59BD9B4E019 * (seed + 1)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(2) + base
        a = _splitmix64(idx)
        b = _splitmix64(idx + np.uint64(1))
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740993.0)
    u2 = (b >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(z.astype(np.float32).reshape(shape))


def _lr_mul_for(key, lr_mlp):
    if key.startswith(('spatial_mapping_network.', 'style_mapping_network.', 'interact.')):
        return lr_mlp
    return 1.0


def fill_state_dict(state, seed=0, lr_mlp=0.01):
    """Overwrite every *parameter-like* entry of `state` (a name->tensor mapping taken from a
    freshly constructed model) in place; constant buffers (eye tokens, FIR kernels) are kept,
    `noises.*` buffers are regenerated.  Returns `state`."""
    for key, t in state.items():
        if key in ('token', 'token_spatial') or key.endswith(('.kernel',)):
            continue
        if not torch.is_floating_point(t):
            continue
        lm = _lr_mul_for(key, lr_mlp)
        r = normal(tuple(t.shape), key, seed).to(t.dtype)
        if key.endswith('modulation.bias'):
            v = 1.0 + 0.1 * r
        elif key.endswith('bias') or key.endswith('noise.weight'):
            v = 0.1 * r / lm
        else:
            v = r / lm
        with torch.no_grad():
            t.copy_(v)
    return state


def latents(batch, seed, dim=512, tokens=16):
    """(z, p) ~ N(0,1), shape [B, 512, 16] each (utils/sample.py:8-10,18-19 in the reference)."""
    return normal((batch, dim, tokens), 'latent.z', seed), normal((batch, dim, tokens), 'latent.p', seed)
