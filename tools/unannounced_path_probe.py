"""Path-length step WITHOUT the second_order() hint (what the reference's unchanged loop runs) against the announced route:
which fused node's recorded backward is wrong?   python tools/unannounced_path_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import synth                                        # noqa: E402
from transeditor_amd.model_spatial_query import Generator               # noqa: E402
from transeditor_amd.op import modconv, modulation, styled_rgb          # noqa: E402
from transeditor_amd.train_step import g_path_regularize                # noqa: E402

SIZE = int(os.environ.get('SIZE', '32'))


def grads(G, hint):
    z, p = synth.latents(2, 7003)
    z, p = z.cuda(), p.cuda()
    noise = synth.normal((2, 3, SIZE, SIZE), 'probe.pl').cuda()
    import contextlib
    with (modconv.second_order(wrt='latent') if hint == 'latent' else (modconv.second_order() if hint else contextlib.nullcontext())):
        img, lat, _ = G(z, p, return_latents=True)
    pen, _, pl = g_path_regularize(img, lat, 0, noise)
    G.zero_grad()
    (2.0 * 4 * pen + 0 * img[0, 0, 0, 0]).backward()
    return float(pen), {n: (None if q.grad is None else q.grad.detach().clone()) for n, q in G.named_parameters()}


def compare(a, b, tag):
    worst = []
    top = max(float(v.double().norm()) for v in b.values() if v is not None)
    for n in a:
        if b[n] is not None and float(b[n].double().norm()) <= 1e-6 * top:      # analytically zero (k_transform.bias): round-off both sides
            continue
        if a[n] is None or b[n] is None:
            if (a[n] is None) != (b[n] is None):
                worst.append((float('inf'), n))
            continue
        d = float((a[n].double() - b[n].double()).norm())
        nn = float(b[n].double().norm())
        worst.append((d / nn if nn > 0 else d, n))
    worst.sort(reverse=True)
    print(f'{tag}: worst {[(f"{v:.2e}", n) for v, n in worst[:5]]}', flush=True)


def main():
    token = 2 * (SIZE.bit_length() - 2)
    G = Generator(SIZE, 512, 512, token, n_trans=8, pixel_norm_op_dim=1)
    synth.fill_state_dict(G.state_dict(), 40)
    G = G.cuda()
    pen_ref, ref = grads(G, True)
    pen_lat, lat = grads(G, 'latent')
    compare(lat, ref, f'second_order(wrt=latent) vs second_order(): penalty {pen_lat:.6g} vs {pen_ref:.6g}')
    pen_un, un = grads(G, False)
    compare(un, ref, f'UNANNOUNCED vs second_order(): penalty {pen_un:.6g} vs {pen_ref:.6g}')
    # switch fused nodes off one at a time (unannounced route)
    saved = (styled_rgb.supported, modulation.supported)
    styled_rgb.supported = lambda *a, **k: False
    pen, g = grads(G, False)
    compare(g, ref, f'unannounced, no conv+ToRGB node: penalty {pen:.6g}')
    modulation.supported = lambda *a, **k: False
    pen, g = grads(G, False)
    compare(g, ref, f'unannounced, no conv+ToRGB node, no batched modulation: penalty {pen:.6g}')
    styled_rgb.supported = saved[0]
    pen, g = grads(G, False)
    compare(g, ref, f'unannounced, no batched modulation: penalty {pen:.6g}')
    modulation.supported = saved[1]
    from transeditor_amd.op import linear, token_mlp as tm
    for name in ('batched_supported', 'supported', 'shared_supported'):
        for mod in (linear, tm, modulation):
            if hasattr(mod, name):
                print('has', mod.__name__, name)


if __name__ == '__main__':
    main()
