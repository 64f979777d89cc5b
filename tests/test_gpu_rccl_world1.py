"""RCCL under the data-parallel glue, for real, on the ONE GPU a test box has.

`init_process_group('nccl', world_size=1)` (backend "nccl" IS RCCL on ROCm) and `GradSync(force=True)` run everything the
8-GPU job runs - post-accumulate hooks firing on the autograd thread while HIP kernels are in flight, the multi-tensor
bucket packing, asynchronous all-reduces on RCCL's own stream, `.grad` re-pointed into the buckets, `FusedAdam` reading
the gradients from there - with a group of one rank, where the result is known exactly: a sum over one rank divided by
one.  Compared against the same sub-steps without any exchange (train_spatial_query.py:173-250, 422-428, 494-509).

Runs in a child process with a timeout so that a communicator that fails to come up cannot take the test session down.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
SIZE, BATCH = 32, 4


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Sampler:
    def __init__(self):
        self.n = 0

    def latents(self, n):
        from transeditor_amd import synth
        self.n += 1
        z, p = synth.latents(n, 9000 + self.n)
        return z.cuda(), p.cuda()

    def randn_like(self, t):
        from transeditor_amd import synth
        return synth.normal(tuple(t.shape), 'rccl.pl').to(t)


def _run(force, real):
    from transeditor_amd import synth
    from transeditor_amd.model_spatial_query import Discriminator, Generator
    from transeditor_amd.train_step import TrainStep, default_args
    args = default_args(size=SIZE, batch=BATCH)
    G = Generator(SIZE, 512, 512, args.token, n_trans=8, pixel_norm_op_dim=1)
    Dn = Discriminator(SIZE)
    synth.fill_state_dict(G.state_dict(), 40)
    synth.fill_state_dict(Dn.state_dict(), 41)
    ts = TrainStep(args, 'cuda', G.cuda(), Dn.cuda(), _Sampler(), force_sync=force)
    info = {'hook_launches': 0, 'calls': 0, 'max_exchange_diff': 0.0}
    for sync in (ts.g_sync, ts.d_sync):
        orig = sync.all_reduce

        def wrapped(tag='default', _s=sync, _o=orig):
            info['calls'] += 1
            info['hook_launches'] += sum(w is not None for w in _s._work)      # buckets that left from the backward hooks
            # the LOCAL gradients (still in .grad: the hooks only copied them into the buckets) before the exchange ...
            local = [None if p.grad is None else p.grad.detach().clone() for p in _s.params]
            out = _o(tag)
            if force:
                # ... and what the exchange hands to the optimiser: a sum over ONE rank divided by one is exact
                for p, l in zip(_s.params, local):
                    if l is not None:
                        info['max_exchange_diff'] = max(info['max_exchange_diff'], float((p.grad - l).abs().max()))
            return out
        sync.all_reduce = wrapped
    grads = {}
    for it in range(2):                     # iteration 0 learns the unused set, iteration 1 overlaps the exchange with backward
        ts.d_step(real)
        grads[f'd{it}'] = [None if p.grad is None else p.grad.detach().clone() for p in ts.discriminator.parameters()]
        ts.r1_step(real)
        grads[f'r1{it}'] = [None if p.grad is None else p.grad.detach().clone() for p in ts.discriminator.parameters()]
        ts.g_step()
        grads[f'g{it}'] = [None if p.grad is None else p.grad.detach().clone() for p in ts.generator.parameters()]
        ts.path_step()
        grads[f'path{it}'] = [None if p.grad is None else p.grad.detach().clone() for p in ts.generator.parameters()]
    in_bucket = None
    if force:
        def inside(sync):
            ok = True
            for p in sync.params:
                if p.grad is not None:
                    bi, off = sync._slot[p]
                    ok = ok and p.grad.data_ptr() == sync._flat[bi].data_ptr() + 4 * off
            return ok
        in_bucket = inside(ts.g_sync) and inside(ts.d_sync)
    torch.cuda.synchronize()
    # numpy arrays travel through the queue by value (torch tensors would go through shared memory owned by the child)
    params = {'g': [p.detach().cpu().numpy() for p in ts.generator.parameters()],
              'd': [p.detach().cpu().numpy() for p in ts.discriminator.parameters()]}
    grads = {k: [None if g is None else g.cpu().numpy() for g in v] for k, v in grads.items()}
    return grads, params, info, in_bucket


def _worker(port, q):
    import torch.distributed as dist
    from transeditor_amd import synth
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    torch.cuda.set_device(0)
    real = synth.normal((BATCH, 3, SIZE, SIZE), 'rccl.real').clamp(-1, 1).cuda()
    ref = _run(False, real)                                   # no process group: no exchange at all
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        t = torch.arange(1024, dtype=torch.float32, device='cuda')
        dist.all_reduce(t)                                    # one plain RCCL collective first
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), torch.arange(1024, dtype=torch.float32))
        backend = dist.get_backend()
        got = _run(True, real)
        q.put((backend, ref, got))
    finally:
        dist.destroy_process_group()


def test_gradsync_over_rccl_world1_matches_no_exchange():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    pr = ctx.Process(target=_worker, args=(_free_port(), q))
    pr.start()
    try:
        backend, (g0, p0, i0, _), (g1, p1, i1, in_bucket) = q.get(timeout=420)
    finally:
        pr.join(timeout=60)
        if pr.is_alive():
            pr.kill()
    assert pr.exitcode == 0
    assert backend == 'nccl'
    assert i0['hook_launches'] == 0                            # no group: GradSync is inert
    assert i1['calls'] == 8 and i1['hook_launches'] > 0, i1   # buckets left from the hooks while the backward was running
    assert in_bucket, '.grad must alias the all-reduced bucket (FusedAdam reads it there)'
    assert i1['max_exchange_diff'] == 0.0, i1           # every exchanged gradient bit-identical to the local one (all 8 calls)
    for key in ('d0', 'r10', 'g0', 'path0'):            # (later steps of two separately evolving runs are not comparable: atomics + Adam)
        top = max(float(np.abs(a).max()) for a in g0[key] if a is not None)
        for i, (a, b) in enumerate(zip(g0[key], g1[key])):
            assert (a is None) == (b is None), (key, i)
            if a is None:
                continue
            # sum over one rank / 1 is exact; what differs run to run is the atomic accumulation order of a few reducers
            # of the backward itself (and, from iteration 1 on, Adam's reaction to it: first steps move by lr * sign(g))
            tol = 1e-5 if key == 'd0' else 5e-2
            assert float(np.abs(a - b).max()) <= tol * max(float(np.abs(a).max()), 1e-3 * top), (key, i)
    for net in ('g', 'd'):
        for a, b in zip(p0[net], p1[net]):
            assert np.isfinite(b).all()
            close = (np.abs(a - b) <= 1e-3 + 1e-2 * np.abs(a)).mean()          # two runs, 8 Adam steps each: same trajectory
            assert float(close) > 0.9, net
