import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transeditor_amd.model_spatial_query import Generator
from transeditor_amd.inference import GeneratorSampler
from transeditor_amd import synth
dev='cuda'
G = Generator(256, 512, 512, 14, n_trans=8, pixel_norm_op_dim=1)
sd = G.state_dict(); synth.fill_state_dict(sd, 3); G.load_state_dict(sd); G = G.to(dev)
for B in (1, 8):
    z, p = (t.to(dev) for t in synth.latents(B, 5))
    outs = {}
    for fused in (False, True):
        S = GeneratorSampler(G, use_graph=True, fused_attention=fused)
        for _ in range(3): img = S(z, p)[0]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): img = S(z, p)[0]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        outs[fused] = img
        print(f'B={B} fused={fused}: {dt*1e3:.3f} ms  {B/dt:.1f} img/s', flush=True)
    e = float((outs[True] - outs[False]).abs().max() / outs[False].abs().max())
    print(f'B={B} image rel diff fused vs layer-by-layer: {e:.2e}')
