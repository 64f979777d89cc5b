"""Latent samplers; same names / argument meaning as the reference's utils/sample.py:3-21.
`args` needs `.latent` (512) and `.para_num` (16)."""
import torch


def prepare_param(n_sample, args, device, method="batch_same", truncation=1.0):
    if method == "batch_same":
        return torch.randn(args.para_num, args.latent, device=device).repeat(n_sample, 1, 1) * truncation
    if method == "batch_diff":
        return torch.randn(n_sample, args.para_num, args.latent, device=device) * truncation
    if method == "spatial":                      # [B, 512, 16]
        return torch.randn(n_sample, args.latent, args.para_num, device=device) * truncation
    if method == "spatial_same":
        return torch.randn(args.latent, args.para_num, device=device).repeat(n_sample, 1, 1) * truncation
    return None


def prepare_noise_new(n_sample, args, device, method="multi", truncation=1.0, mode='train'):
    if method == 'query':                        # [B, 512, 16]
        return torch.randn(n_sample, args.latent, args.para_num, device=device) * truncation
    if method == 'query_same':
        return torch.randn(args.latent, args.para_num, device=device).repeat(n_sample, 1, 1) * truncation
    return None
