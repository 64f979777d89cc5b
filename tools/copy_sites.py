"""Diagnostic: python call sites of real `.contiguous()` / `.reshape()` copies during one generator forward + backward (which
strided tensors get densified, and where).   python tools/copy_sites.py"""
import collections
import sys
import traceback

import torch

sys.path.insert(0, '.')
from transeditor_amd.model_spatial_query import Generator

dev = torch.device('cuda')
torch.manual_seed(0)
G = Generator(256, 512, 512, 14, n_trans=8, pixel_norm_op_dim=1).to(dev)
z, p = torch.randn(16, 512, 16, device=dev), torch.randn(16, 512, 16, device=dev)
sites = collections.Counter()
orig_c, orig_r = torch.Tensor.contiguous, torch.Tensor.reshape


def site():
    fr = [f for f in traceback.extract_stack()[:-2] if 'transeditor_amd' in f.filename]
    return ' <- '.join(f'{f.filename.split("transeditor_amd/")[-1]}:{f.lineno}' for f in fr[-3:][::-1])


def contiguous(self, *a, **k):
    if self.is_cuda and not self.is_contiguous():
        sites[('contiguous', tuple(self.shape), site())] += 1
    return orig_c(self, *a, **k)


def reshape(self, *shape):
    out = orig_r(self, *shape)
    if self.is_cuda and out.data_ptr() != self.data_ptr() and self.numel() > 0:
        sites[('reshape', tuple(self.shape), site())] += 1
    return out


(G(z, p)[0]).sum().backward()
torch.Tensor.contiguous, torch.Tensor.reshape = contiguous, reshape
img = G(z, p)[0]
img.sum().backward()
torch.Tensor.contiguous, torch.Tensor.reshape = orig_c, orig_r
for (kind, shape, where), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f'{n:4d} {kind:10s} {str(shape):22s} {where}')
