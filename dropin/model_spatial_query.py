"""Drop-in for the reference's `model_spatial_query.py` (whole module): `PYTHONPATH=<this repo>/dropin:<this repo>:<reference>`
makes `from model_spatial_query import Generator, Discriminator` (train_spatial_query.py:27, test_spatial_query.py:14) resolve
to the MI355X path without editing the reference tree.  Pure re-export."""
from transeditor_amd.model_spatial_query import *                     # noqa: F401,F403
from transeditor_amd.model_spatial_query import (                     # noqa: F401  (names a `*` import would hide or that scripts reach for)
    Attention, AttentionBlock, Blur, ConstantInput, ConvLayer, Discriminator, Downsample, EqualConv2d, EqualLinear,
    FusedLeakyReLU, Generator, ModulatedConv2d, NoiseInjection, PixelNorm, ResBlock, ScaledLeakyReLU, StyledConv, ToRGB,
    Upsample, fused_leaky_relu, make_kernel, upfirdn2d)
