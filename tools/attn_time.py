"""time the fused attention stack (fwd, fwd+bwd) against the single-op composition at the generator's shapes (N = 16, 8 blocks)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transeditor_amd import synth, _lib                        # noqa: E402
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'exp', f'libte_{sys.argv[1]}.so')
from transeditor_amd.model_spatial_query import AttentionBlock  # noqa: E402
from transeditor_amd.op import attn_stack as A                  # noqa: E402

DEV = 'cuda'
N, nb = 16, 8
blocks = [AttentionBlock(528, 528, 512, lr_mul=0.01)] + [AttentionBlock(512, 512, 512, lr_mul=0.01) for _ in range(nb - 1)]
blocks = [b.to(DEV) for b in blocks]
prm = [A.block_params(b) for b in blocks]
x0 = torch.randn(N, 16, 528, device=DEV, requires_grad=True)
p0 = torch.randn(N, 16, 528, device=DEV, requires_grad=True)
p = torch.randn(N, 16, 512, device=DEV, requires_grad=True)
gy = torch.randn(N, 16, 512, device=DEV)
ins = [x0, p0, p] + [t for b in prm for t in b]


def timeit(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def fwd(f):
    with torch.no_grad():
        return f(x0, p0, p, prm, 0.01, 128 ** -0.5)


def fb(f):
    y = f(x0, p0, p, prm, 0.01, 128 ** -0.5)
    torch.autograd.grad(y, ins, gy)


print(f'fused  : fwd {timeit(lambda: fwd(A.attention_stack)):8.1f} us   fwd+bwd {timeit(lambda: fb(A.attention_stack)):8.1f} us')
if len(sys.argv) <= 1:
    print(f'single : fwd {timeit(lambda: fwd(A._composite)):8.1f} us   fwd+bwd {timeit(lambda: fb(A._composite)):8.1f} us')
