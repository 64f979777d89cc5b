"""Time the four sub-steps of the FFHQ-256 train iteration separately (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transeditor_amd.train_step import TrainStep, default_args, accumulate
dev = 'cuda'
args = default_args(size=256, batch=16)
ts = TrainStep(args, dev)
real = torch.randn(16, 3, 256, 256, device=dev).clamp(-1, 1)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
d, r1, g, p = t(lambda: ts.d_step(real)), t(lambda: ts.r1_step(real)), t(ts.g_step), t(ts.path_step)
ema = t(lambda: accumulate(ts.g_ema, ts.generator, ts.accum))
print(f'd_step {d:.1f} ms | r1_step {r1:.1f} ms | g_step {g:.1f} ms | path_step {p:.1f} ms | ema {ema:.2f} ms')
print(f'amortised iteration: {d + r1 / 16 + g + p / 4 + ema:.1f} ms -> {16e3 / (d + r1 / 16 + g + p / 4 + ema):.1f} img/s')
