"""Census of the aten ops / kernels one generator step issues (torch profiler), grouped by op + input shapes."""
import math
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, '.')
from transeditor_amd.model_spatial_query import Generator

dev = torch.device('cuda')
torch.manual_seed(1234)
G = Generator(256, 512, 512, 14, n_trans=8, pixel_norm_op_dim=1).to(dev)
params = list(G.parameters())
z, p = torch.randn(16, 512, 16, device=dev), torch.randn(16, 512, 16, device=dev)
wimg = torch.randn(16, 3, 256, 256, device=dev)


def step():
    for q in params:
        q.grad = None
    (G(z, p)[0] * wimg).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True, group_by_stack_n=4)
rows = [e for e in ev if e.device_time_total > 0 or e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.count)
import re
print('---- aten / autograd ops that launch work (count, self device us, op, shapes)')
for e in rows:
    if e.key.startswith(('void ', '(anonymous', 'Cijk', 'Memcpy', 'Memset', '__amd')) or 'evaluate_function' in e.key:
        continue
    if e.self_device_time_total <= 0:
        continue
    print(f'{e.count:5d} {e.self_device_time_total:9.0f}us {e.key[:40]:40s} {str(e.input_shapes)[:110]}')
print('---- kernels')
for e in rows:
    if e.key.startswith(('void ', '(anonymous', 'Cijk', 'Memcpy', 'Memset', '__amd')):
        k = re.sub(r'at::native::|\(anonymous namespace\)::|void ', '', e.key)
        print(f'{e.count:5d} {e.self_device_time_total:9.0f}us {k[:150]}')
