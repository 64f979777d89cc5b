"""End-to-end latent-gradient error hunt (see gpu_grad_probe.py): which component makes dz inaccurate?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import te_oracle as O                              # noqa: E402
from transeditor_amd import synth                              # noqa: E402
from test_oracle_golden import generator_state                # noqa: E402

DEV = 'cuda'


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def run(seed_model, seed_lat, size=32):
    G, sd = generator_state(size, seed_model)
    G.load_state_dict(sd)
    G = G.to(DEV)
    z, p = synth.latents(2, seed_lat)
    w = synth.normal((2, 3, size, size), 'smoke.w')
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}

    def oracle(P, dt, dev):
        zc = z.detach().clone().to(dt).to(dev).requires_grad_(True)
        pc = p.detach().clone().to(dt).to(dev).requires_grad_(True)
        Pd = {k: v.to(dev) for k, v in P.items()}
        lat, spc, _, _, _ = O.generator_latent(Pd, zc, pc)
        lat.retain_grad()
        img = O.synthesis(Pd, lat, spc, size)
        (img * w.to(dt).to(dev)).sum().backward()
        return zc.grad, pc.grad, lat.grad, lat

    z64, p64, l64, lat64 = oracle(P64, torch.float64, 'cpu')
    z32, p32, l32, _ = oracle(dict(sd), torch.float32, 'cpu')
    zg, pg, lg, _ = oracle(dict(sd), torch.float32, DEV)
    print(f'[model {seed_model} lat {seed_lat}] oracle cpu32: dz {rel(z32, z64):.1e} dp {rel(p32, p64):.1e} dlat {rel(l32, l64):.1e}')
    print(f'                      oracle on GPU (torch ops, MIOpen): dz {rel(zg, z64):.1e} dp {rel(pg, p64):.1e} dlat {rel(lg, l64):.1e}')

    def product(create_graph):
        zd, pd = z.detach().clone().to(DEV).requires_grad_(True), p.detach().clone().to(DEV).requires_grad_(True)
        img, latent, _ = G(zd, pd, return_latents=True)
        gl, = torch.autograd.grad((img * w.to(DEV)).sum(), latent, retain_graph=True, create_graph=create_graph)
        gz, gp = torch.autograd.grad((img * w.to(DEV)).sum(), (zd, pd), create_graph=create_graph)
        return gz.detach(), gp.detach(), gl.detach()
    for cg in (False, True):
        gz, gp, gl = product(cg)
        print(f'                      product (create_graph={cg!s:5}): dz {rel(gz, z64):.1e} dp {rel(gp, p64):.1e} dlat {rel(gl, l64):.1e}')
    # amplification check: push the product's latent gradient through the fp64 attention-stack Jacobian
    zc, pc = z.detach().double().requires_grad_(True), p.detach().double().requires_grad_(True)
    lat, spc, _, _, _ = O.generator_latent(P64, zc, pc)
    gzm, = torch.autograd.grad((lat * gl.double().cpu()).sum(), zc)
    print(f'                      fp64 Jacobian^T applied to the PRODUCT dlat: dz {rel(gzm, z64):.1e}   (amplification of the dlat error)')
    gzm32, = torch.autograd.grad((O.generator_latent(P64, zc, pc)[0] * l32.double()).sum(), zc)
    print(f'                      fp64 Jacobian^T applied to the CPU32  dlat: dz {rel(gzm32, z64):.1e}')


if __name__ == '__main__':
    run(11, 77)
    run(21, 555)
