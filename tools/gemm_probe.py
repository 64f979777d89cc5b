import torch
dev='cuda'
def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
x=torch.randn(256,512,device=dev)
for N in (128,256,384,512):
    W=torch.randn(N,512,device=dev); b=torch.randn(N,device=dev)
    print(f'x[256,512]@W[{N},512]^T: linear {timeit(lambda: torch.nn.functional.linear(x,W,b)):.1f} us | transposed (W@x^T)^T {timeit(lambda: (W@x.t()).t()):.1f} | mm {timeit(lambda: x@W.t()):.1f}')
gy=torch.randn(256,512,device=dev); W=torch.randn(512,128,device=dev)
print(f'dx gy[256,512]@W[512,128]: {timeit(lambda: gy@W):.1f} | transposed {timeit(lambda: (W.t()@gy.t()).t()):.1f}')
x3=torch.randn(16,16,512,device=dev); W=torch.randn(128,512,device=dev)
print(f'3-D linear [16,16,512]->128: {timeit(lambda: torch.nn.functional.linear(x3,W)):.1f}')
Wp=torch.randn(512,128,device=dev); h=torch.randn(256,128,device=dev)
print(f'proj fwd h[256,128]@Wp^T[128,512]: {timeit(lambda: h@Wp.t()):.1f}')
