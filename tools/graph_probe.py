"""Is the generator step launch-bound?  Eager issue time vs wall time, then the same step replayed as one hipGraph."""
import math
import sys
import time

import torch

sys.path.insert(0, '.')
from transeditor_amd.model_spatial_query import Generator

size, B = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device('cuda')
torch.manual_seed(1234)
G = Generator(size, 512, 512, 2 * (int(math.log2(size)) - 1), n_trans=8, pixel_norm_op_dim=1).to(dev)
params = list(G.parameters())
zs = [torch.randn(B, 512, 16, device=dev) for _ in range(13)]
ps = [torch.randn(B, 512, 16, device=dev) for _ in range(13)]
wimg = torch.randn(B, 3, size, size, device=dev)


def step(z, p):
    for q in params:
        q.grad = None
    img = G(z, p)[0]
    (img * wimg).sum().backward()


for i in range(3):
    step(zs[i], ps[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    step(zs[3 + i], ps[3 + i])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'eager: issue {1e3 * (t1 - t0) / 10:.2f} ms/step, wall {1e3 * (t2 - t0) / 10:.2f} ms/step', flush=True)
ref = {n: q.grad.clone() for n, q in G.named_parameters() if q.grad is not None}

# ---- whole-step graph
sz, sp = zs[0].clone(), ps[0].clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(2):
        step(sz, sp)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
for q in params:
    q.grad = None
with torch.cuda.graph(g):
    img = G(sz, sp)[0]
    (img * wimg).sum().backward()
torch.cuda.synchronize()
sz.copy_(zs[12]); sp.copy_(ps[12])
g.replay()
torch.cuda.synchronize()
worst = 0.0
for n, q in G.named_parameters():
    if q.grad is not None and n in ref:
        worst = max(worst, float((q.grad - ref[n]).abs().max() / (ref[n].abs().max() + 1e-30)))
print(f'graph replay vs eager grads (same latents): worst rel err {worst:.2e}', flush=True)
t0 = time.perf_counter()
for i in range(10):
    sz.copy_(zs[3 + i]); sp.copy_(ps[3 + i])
    g.replay()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f'graph: wall {1e3 * (t1 - t0) / 10:.2f} ms/step -> {B * 10 / (t1 - t0):.1f} img/s', flush=True)
