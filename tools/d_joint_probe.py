"""Diagnostic: parameter gradients of the discriminator step through the joint pass (chunks=2), through two separate passes, and
through the CPU oracle in fp64 (two passes) - which route deviates, and on which layers.   python tools/d_joint_probe.py [size B]"""
import sys

import torch

sys.path.insert(0, '.')
from oracle import te_oracle as O
from transeditor_amd import synth
from transeditor_amd.model_spatial_query import Discriminator
from transeditor_amd.train_step import d_logistic_loss

size, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 8)
DEV = 'cuda'
torch.manual_seed(5)
D = Discriminator(size).to(DEV)
synth.fill_state_dict(D.state_dict(), 77)
fake, real = torch.randn(B, 3, size, size, device=DEV), torch.randn(B, 3, size, size, device=DEV).clamp(-1, 1)
names = [n for n, _ in D.named_parameters()]
params = list(D.parameters())
fp, rp = D(torch.cat([fake, real]), chunks=2).chunk(2)
ga = torch.autograd.grad(d_logistic_loss(rp, fp), params)
gb = torch.autograd.grad(d_logistic_loss(D(real), D(fake)), params)
P = {n: p.detach().double().cpu().requires_grad_(True) for n, p in D.named_parameters()}
lo = O.d_logistic_loss(O.discriminator_forward(P, real.double().cpu(), size), O.discriminator_forward(P, fake.double().cpu(), size))
gr = torch.autograd.grad(lo, [P[n] for n in names])
rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))
print(f'D {size} px, B = {B}:  joint vs fp64 oracle | two passes vs fp64 oracle | joint vs two passes')
for n, a, b, r in zip(names, ga, gb, gr):
    print(f'{n:28s} {rel(a, r):10.2e} {rel(b, r):10.2e} {rel(a, b.double().cpu()):10.2e}   |g| {float(r.norm()):.3e}')
