"""Plain-torch restatements of the dense-stack formulas, written from the reference's lines and used ONLY as the independent
side of op-level comparisons (first order in fp64, recorded backward in fp32/fp64).  Nothing here imports the product: a test
that compares `transeditor_amd.op.*` with one of these compares it with stock framework ops (einsum / F.linear / F.softmax),
not with the product's own recorded-backward expressions."""
import math

import torch
import torch.nn.functional as F


def attention_core(q, k, v, scale, groups):
    """model_spatial_query.py:888-894: q [N,M,C], k / v [N,L,C] -> (o [N,M,C], sim [N,G,M,L])"""
    N, M, C = q.shape
    L, D = k.shape[1], C // groups
    qh, kh, vh = q.reshape(N, M, groups, D), k.reshape(N, L, groups, D), v.reshape(N, L, groups, D)
    sim = F.softmax(torch.einsum('nmgd,nlgd->ngml', qh, kh) * scale, dim=-1)
    return torch.einsum('ngml,nlgd->nmgd', sim, vh).reshape(N, M, C), sim


def equal_linear(x, weight, bias, alpha, beta, act=None, residual=None):
    """model_spatial_query.py:213-221 (+ the GELU / skip the attention block applies around it, :932-934)"""
    y = F.linear(x, weight * alpha, None if bias is None else bias * beta)
    if act == 'gelu':
        y = F.gelu(y)
    elif act == 'lrelu':
        y = F.leaky_relu(y, 0.2) * math.sqrt(2)
    return y if residual is None else y + residual


def token_mlp(x, weights, biases, scale, lr_mul):
    """model_spatial_query.py:626-646: token t of x [B, D, C] through its own EqualLinear + fused lrelu -> [B, T, D]"""
    return torch.stack([F.leaky_relu(F.linear(x[:, :, t], w * scale, b * lr_mul), 0.2) * math.sqrt(2)
                        for t, (w, b) in enumerate(zip(weights, biases))], 1)


def demod(w, s, wscale, eps):
    """model_spatial_query.py:299-304 on the B materialised weight copies: d[b, co] = rsqrt(sum (wscale w s)^2 + eps)"""
    wb = wscale * w.unsqueeze(0) * s.view(s.shape[0], 1, -1, 1, 1)
    return torch.rsqrt(wb.pow(2).sum([2, 3, 4]) + eps)


def minibatch_stddev(out, group, feat=1):
    """model_spatial_query.py:844-852"""
    batch, channel, height, width = out.shape
    sd = out.view(group, -1, feat, channel // feat, height, width)
    sd = torch.sqrt(sd.var(0, unbiased=False) + 1e-8).mean([2, 3, 4], keepdims=True).squeeze(2)
    return torch.cat([out, sd.repeat(group, 1, height, width)], 1)


def pixel_norm(x, dim):
    """model_spatial_query.py:80-81"""
    return x * torch.rsqrt(torch.mean(x ** 2, dim=dim, keepdim=True) + 1e-8)
