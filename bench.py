"""Benchmark of the TransEditor generator hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch: FFHQ-256 generator forward + backward
(BASELINE.json configs[1]: batch 16 per GPU, num_trans=8, fp32), latents already resident in HBM.  With
N > 1 every rank runs its own batch (weak scaling, data parallel) and the step includes the gradient
all-reduce over RCCL/xGMI (the only exchange step of the path).  Rank 0 prints ONE JSON line.

Extra objects in the JSON:
  roofline      dominant kernel class (fp32-MFMA implicit-GEMM convolutions): ALGORITHMIC FLOPs of its launches
                in the timed region / their summed duration, measured live with HIP events on the launch stream.
  cpu_baseline  the CPU oracle (oracle/te_oracle.py, a port of the reference's algorithm) timed on this box's
                host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3          # MI355X fp32 (vector == fp32-MFMA), MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--workload', choices=['generator', 'train'], default='generator',
                    help='generator = BASELINE configs[1] (default, the driver contract); train = configs[2]/[3] full G+D step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of individual te_conv / te_wgrad launches (they run on torch's current stream, so
    torch.cuda.Event brackets exactly the kernel)."""

    def __init__(self):
        self.records = []      # (class, flops, start, end)
        self.enabled = False

    def install(self):
        from transeditor_amd import _lib
        timer = self
        orig_conv, orig_wgrad = _lib.conv, _lib.wgrad_slabs
        names = {_lib.CONV_3X3: 'conv3x3', _lib.CONV_T2: 'convT2', _lib.CONV_S2: 'convS2', _lib.CONV_1X1: 'conv1x1'}

        def conv(x, wp, kind, M, H, W, *a, **k):
            if not timer.enabled:
                return orig_conv(x, wp, kind, M, H, W, *a, **k)
            B, K = x.shape[0], x.shape[1]
            taps = 1 if kind == _lib.CONV_1X1 else 9
            flops = 2.0 * taps * K * M * H * W * B       # algorithmic: T2/S2 counted on the low-res grid
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig_conv(x, wp, kind, M, H, W, *a, **k)
            e.record()
            timer.records.append((names[kind], flops, s, e))
            return out

        def wgrad(g, x, kind, H, W):
            if not timer.enabled:
                return orig_wgrad(g, x, kind, H, W)
            taps = 1 if kind == _lib.CONV_1X1 else 9
            flops = 2.0 * taps * g.shape[1] * x.shape[1] * H * W * g.shape[0]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig_wgrad(g, x, kind, H, W)
            e.record()
            timer.records.append(('wgrad_' + names[kind], flops, s, e))
            return out

        _lib.conv, _lib.wgrad_slabs = conv, wgrad

    def summary(self):
        agg = {}
        for name, flops, s, e in self.records:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += s.elapsed_time(e) * 1e-3
        return {k: {'launches': v[0], 'avg_ms': 1e3 * v[2] / v[0], 'tflops': v[1] / v[2] / 1e12, 'total_ms': 1e3 * v[2]}
                for k, v in agg.items()}


def cpu_baseline(size):
    """Oracle generator fwd+bwd on the host cores; bounded sample (batch 1, 1 warm-up + 10 timed iterations, ~12 s)."""
    from oracle import te_oracle as O
    from transeditor_amd import synth
    from transeditor_amd.model_spatial_query import Generator
    # 256 hardware threads oversubscribe the small CPU convolutions badly (measured 187 s/iter); use a socket's
    # worth of threads and say so in `cores`
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    token = 2 * (int(math.log2(size)) - 1)
    sd = Generator(size, 512, 512, token, n_trans=8, pixel_norm_op_dim=1).state_dict()
    P = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'noises' not in k and 'kernel' not in k
             and not k.startswith('token') else v) for k, v in sd.items()}
    leaves = [v for v in P.values() if v.requires_grad]
    B, iters = 1, 10
    times = []
    for it in range(iters + 1):
        z, p = synth.latents(B, 900 + it)
        t0 = time.perf_counter()
        img, _, _ = O.generator_forward(P, z, p, size)
        torch.autograd.grad(img.sum(), leaves, allow_unused=True)
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / iters
    model = 'unknown'
    try:
        model = next(l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name'))
    except Exception:
        pass
    return {'value': B / dt, 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'CPU oracle (PyTorch fp32 restatement of the reference), generator fwd+bwd {size}x{size}, '
                      f'batch {B}, 1 warm-up + {iters} timed iterations, {dt:.2f} s/iter',
            'cpu_model': model, 'host_threads': os.cpu_count()}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU path for the product)'
    # TE_BENCH_SHARE_GPU=1 (rehearsal of the multi-process path on a 1-GPU box): every rank uses cuda:0 and the
    # collectives go through gloo instead of RCCL, which refuses two ranks on one device.  Never set by the driver.
    share = os.environ.get('TE_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('gloo' if share else 'nccl', init_method='env://')      # "nccl" == RCCL on ROCm
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from transeditor_amd.model_spatial_query import Generator
    from transeditor_amd.utils import distributed as D

    size, B = args.size, args.batch
    token = 2 * (int(math.log2(size)) - 1)
    torch.manual_seed(1234)                                        # same init on every rank (reference-style randn init)
    G = Generator(size, 512, 512, token, n_trans=8, pixel_norm_op_dim=1).to(dev)
    D.broadcast_module(G)
    sync = D.GradSync(G)
    params = [p for p in G.parameters()]
    torch.manual_seed(1000 + rank)
    n_in = args.steps + args.warmup
    zs = [torch.randn(B, 512, 16, device=dev) for _ in range(n_in)]   # resident in HBM before the timed region
    ps = [torch.randn(B, 512, 16, device=dev) for _ in range(n_in)]
    wimg = torch.randn(B, 3, size, size, device=dev)

    timer = KernelTimer()
    if not args.no_kernel_timing:
        timer.install()

    if args.workload == 'train':
        # BASELINE configs[2] (N = 1) / configs[3] (N > 1): full train_spatial_query step with lazy R1 (1/16) and
        # path-length (1/4) regularisers on synthetic "real" images; --steps should be a multiple of 16
        from transeditor_amd.train_step import TrainStep, default_args
        targs = default_args(size=size, batch=B)
        ts = TrainStep(targs, dev, generator=G)
        reals = [torch.randn(B, 3, size, size, device=dev).clamp(-1, 1) for _ in range(4)]
        it = [0]

        def train_iter(_):
            ts.iteration(it[0], reals[it[0] % 4])
            it[0] += 1
        for i in range(args.warmup):
            train_iter(i)
        it[0] = 0
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            train_iter(i)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
        if rank == 0:
            print(json.dumps({
                'metric': '256x256 images/sec/GPU, G+D fwd+bwd, batch 16; 1/2/4/8-GPU scaling',
                'value': world * B * args.steps / elapsed, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
                'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                'config': {'workload': f'FFHQ-{size} full G+D train step (BASELINE configs[2]/[3]): D step, R1 every 16, G step, '
                                       f'path-length reg every 4 on batch {B // 2}, Adam, EMA; batch {B}/GPU',
                           'global_batch': world * B, 'parallelism': f'dp{world}'}}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    def step(i):
        for p in params:
            p.grad = None
        img = G(zs[i], ps[i])[0]
        (img * wimg).sum().backward()
        sync.all_reduce()                                          # no-op at world == 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    timer.enabled = not args.no_kernel_timing
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        out = {
            'metric': '256x256 images/sec/GPU, G+D fwd+bwd, batch 16; 1/2/4/8-GPU scaling',
            'value': world * B * args.steps / elapsed, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'FFHQ-{size} generator fwd+bwd (BASELINE configs[1]), batch {B}/GPU, num_trans=8, '
                                   f'random-init weights, random latents', 'global_batch': world * B,
                       'parallelism': f'dp{world}', 'per_gpu_images_per_sec': B * args.steps / elapsed},
        }
        if not args.no_kernel_timing:
            ks = timer.summary()
            conv_keys = [k for k in ks if not k.endswith('1x1')]
            flops = sum(ks[k]['tflops'] * ks[k]['total_ms'] for k in conv_keys)      # TFLOP*ms
            tms = sum(ks[k]['total_ms'] for k in conv_keys)
            ach = flops / tms if tms else 0.0
            traffic, traffic_note = None, None
            try:    # HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/pmc_round.sh + pmc_summary.py)
                pm = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')))
                k3 = next(v for k, v in pm.items() if k.startswith('conv_mfma_kernel<0'))
                traffic = (k3['hbm_read_mb'] + k3['hbm_write_mb']) * 1e6
                traffic_note = (f"PMC pass of the 128->128 @256x256 batch-16 launch: FETCH_SIZE x2 (gfx950 correction) = "
                                f"{k3['hbm_read_mb']:.0f} MB read + WRITE_SIZE {k3['hbm_write_mb']:.0f} MB written vs "
                                f"{k3['algorithmic_mb']:.0f} MB algorithmic; MFMA utilisation {k3['mfma_util_pct']:.1f} % "
                                f"at {k3['mhz']:.0f} MHz (profiles/r01_pmc_summary.txt)")
            except Exception:
                pass
            out['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': ach / PEAK_FP32_TFLOPS, 'traffic': traffic, 'traffic_note': traffic_note,
                               'kernel': 'conv_mfma_kernel / wgrad_mfma_kernel (fp32 v_mfma_f32_32x32x2, all 3x3 kinds)',
                               'kernel_time_share': tms / (ms * args.steps), 'per_kernel': ks}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(size)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
