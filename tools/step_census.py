"""Census of the framework (aten) ops that launch GPU work inside the timed training iteration: 16 iterations (one R1 step,
four path-length steps) under the torch profiler, grouped by op + input shapes + the innermost python frames.  Answers
"which of the ~1 900 dispatches per iteration are not ours, and who issues them".   python tools/step_census.py [iters]"""
import re
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, '.')
from transeditor_amd.train_step import TrainStep, default_args

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda')
torch.manual_seed(1234)
ts = TrainStep(default_args(size=256, batch=16), dev)
reals = [torch.randn(16, 3, 256, 256, device=dev).clamp(-1, 1) for _ in range(4)]
for i in range(3):
    ts.iteration(i, reals[i % 4])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for i in range(N):
        ts.iteration(i, reals[i % 4])
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True, group_by_stack_n=6)
KERNEL = ('void ', '(anonymous', 'Cijk', 'Memcpy', 'Memset', '__amd')
rows = [e for e in ev if e.self_device_time_total > 0 and not e.key.startswith(KERNEL) and 'evaluate_function' not in e.key]
rows.sort(key=lambda e: -e.self_device_time_total if len(sys.argv) > 2 and sys.argv[2] == "time" else -e.count)
print(f'---- framework ops that launch work, {N} iterations (count per iteration, self device us per iteration, op, shapes, frames)')
tot_n = tot_t = 0
for e in rows:
    tot_n += e.count
    tot_t += e.self_device_time_total
    frames = [re.sub(r'.*/(transeditor_amd|torch)/', r'\1/', s) for s in e.stack if 'transeditor_amd' in s or 'bench' in s][:3]
    print(f'{e.count / N:7.1f} {e.self_device_time_total / N:8.0f}us {e.key[:34]:34s} {str(e.input_shapes)[:90]:90s} {" <- ".join(frames)[:200]}')
print(f'TOTAL {tot_n / N:.0f} framework launches, {tot_t / N / 1e3:.2f} ms per iteration')
print('---- kernels (count per iteration, device us per iteration)')
ks = [e for e in ev if e.key.startswith(KERNEL)]
agg = {}
for e in ks:
    k = re.sub(r'at::native::|\(anonymous namespace\)::|void ', '', e.key)[:120]
    a = agg.setdefault(k, [0, 0.0])
    a[0] += e.count
    a[1] += e.self_device_time_total
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f'{c / N:7.1f} {t / N:8.0f}us {k}')
print(f'TOTAL {sum(a[0] for a in agg.values()) / N:.0f} kernels, {sum(a[1] for a in agg.values()) / N / 1e3:.2f} ms per iteration')
