"""Where does the latent-gradient error of the HIP path come from?  Compare against an fp64 CPU oracle:
(1) synthesis network only (latent given), (2) mapping + attention stack only, (3) single fused modconv."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import te_oracle as O                              # noqa: E402
from transeditor_amd import synth                              # noqa: E402
from test_oracle_golden import generator_state                # noqa: E402

DEV = 'cuda'


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    size = 32
    G, sd = generator_state(size, 21)
    G.load_state_dict(sd)
    G = G.to(DEV)
    z, p = synth.latents(2, 555)
    w = synth.normal((2, 3, size, size), 'cond.w')
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    P32 = dict(sd)

    # ---- (1) synthesis only: d img / d latent, d img / d spatialcode
    with torch.no_grad():
        lat, spc, _, _, _ = O.generator_latent(P64, z.double(), p.double())
    for tag, P, dt in (('cpu32', P32, torch.float32), ('cpu64', P64, torch.float64)):
        l, s_ = lat.to(dt).requires_grad_(True), spc.to(dt).requires_grad_(True)
        img = O.synthesis(P, l, s_, size)
        g = torch.autograd.grad((img * w.to(dt)).sum(), (l, s_))
        if tag == 'cpu64':
            ref = g
        else:
            g32 = g
    acts = {}
    def run_product():
        zd, pd = z.to(DEV).requires_grad_(True), p.to(DEV).requires_grad_(True)
        img, latent, _ = G(zd, pd, return_latents=True)
        latent.retain_grad()
        (img * w.to(DEV)).sum().backward()
        return zd.grad, pd.grad, latent.grad
    gz, gp, glat = run_product()
    print(f'(1) synthesis d/dlatent : cpu32 vs f64 {rel(g32[0], ref[0]):.2e} | hip vs f64 {rel(glat, ref[0]):.2e}')
    per_tok = [(rel(glat[:, i], ref[0][:, i])) for i in range(glat.shape[1])]
    print('    per latent token (layer order conv1, rgb1, up, conv, rgb ...):', ' '.join(f'{e:.1e}' for e in per_tok))

    # ---- (2) mapping + attention only, driven by the SAME upstream gradient (fp64 latent grad)
    gl64 = ref[0]
    def lat_grads(P, dt):
        zc, pc = z.to(dt).requires_grad_(True), p.to(dt).requires_grad_(True)
        l, s_, _, _, _ = O.generator_latent(P, zc, pc)
        return torch.autograd.grad((l * gl64.to(dt)).sum(), (zc, pc))
    r64, r32 = lat_grads(P64, torch.float64), lat_grads(P32, torch.float32)
    zd, pd = z.to(DEV).requires_grad_(True), p.to(DEV).requires_grad_(True)
    l = G(zd, pd, return_only_style_latent=True)
    gh = torch.autograd.grad((l * gl64.float().to(DEV)).sum(), (zd, pd))
    print(f'(2) attention stack dz  : cpu32 vs f64 {rel(r32[0], r64[0]):.2e} | hip vs f64 {rel(gh[0], r64[0]):.2e}')
    print(f'    attention stack dp  : cpu32 vs f64 {rel(r32[1], r64[1]):.2e} | hip vs f64 {rel(gh[1], r64[1]):.2e}')
    print(f'    latent forward      : cpu32 vs f64 {rel(O.generator_latent(P32, z, p)[0], lat):.2e} | hip vs f64 {rel(l, lat):.2e}')

    # ---- (3) one fused modconv at a realistic shape: gradients vs fp64
    from transeditor_amd.op.modconv import modconv
    for kind, (B, K, M, H) in (('3x3', (2, 512, 512, 16)), ('up', (2, 512, 512, 8)), ('1x1', (2, 512, 3, 16))):
        ks = 1 if kind == '1x1' else 3
        x = synth.normal((B, K, H, H), 'gp.x')
        wt = synth.normal((M, K, ks, ks), 'gp.w') / math.sqrt(K * ks * ks)
        isc = 1 + 0.5 * synth.normal((B, K), 'gp.i')
        osc = (1 + 0.3 * synth.normal((B, M), 'gp.o')).abs() + 0.1
        def ref_fn(dt):
            t = [v.to(dt).requires_grad_(True) for v in (x, wt, isc, osc)]
            xm = t[0] * t[2][:, :, None, None]
            if kind == '3x3':
                y = F.conv2d(xm, t[1], padding=1)
            elif kind == '1x1':
                y = F.conv2d(xm, t[1])
            else:
                y = F.conv_transpose2d(xm, t[1].transpose(0, 1), stride=2)
            y = y * t[3][:, :, None, None]
            gy = synth.normal(tuple(y.shape), 'gp.g').to(dt)
            return torch.autograd.grad((y * gy).sum(), t), gy
        (r64_, gy), (r32_, _) = ref_fn(torch.float64), ref_fn(torch.float32)
        d = [v.to(DEV).requires_grad_(True) for v in (x, wt, isc, osc)]
        y = modconv(d[0], d[1], d[2], d[3], None, False, kind)
        got = torch.autograd.grad((y * gy.float().to(DEV)).sum(), d)
        print(f'(3) modconv {kind:3s} {K}->{M}@{H}: ' + ' | '.join(
            f'{n} cpu32 {rel(a32, a64):.1e} hip {rel(ah, a64):.1e}' for n, a64, a32, ah in zip(('gx', 'gw', 'gisc', 'gosc'), r64_, r32_, got)))


if __name__ == '__main__':
    main()
