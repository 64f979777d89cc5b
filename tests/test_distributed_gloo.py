"""world_size-2 data-parallel glue on CPU (gloo): the gradient exchange step and the scalar collectives
of the train step (reference utils/distributed.py; train_spatial_query.py:249,296,494-509)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from transeditor_amd.utils import distributed as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        assert D.get_rank() == rank and D.get_world_size() == world
        D.synchronize()
        torch.manual_seed(rank)
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
        unused = torch.nn.Parameter(torch.zeros(3))           # like the generator's 13 unused noise.weight params
        net.register_parameter('unused', unused)
        D.broadcast_module(net)
        w0 = [p.detach().clone() for p in net.parameters()]
        x = torch.randn(5, 8)
        net(x).pow(2).sum().backward()
        local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        sync = D.GradSync(net, bucket_bytes=256)               # tiny buckets: exercise several of them
        assert len(sync.buckets) > 1
        sync.all_reduce()
        assert unused.grad is None, 'a parameter without gradient must keep .grad None (DDP find_unused_parameters semantics)'
        first_sync = [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in net.parameters()]
        # second step: the hooks installed by GradSync fire DURING backward (buckets launch as they fill up)
        for prm in net.parameters():
            prm.grad = None
        x2 = torch.randn(5, 8)
        net(x2).pow(2).sum().backward()
        assert any(w is not None for w in sync._work), 'hook-driven launch did not happen'
        hook_local = None    # local grads were overwritten in place by the flat copy only after all_reduce()
        sync.all_reduce()
        assert unused.grad is None
        hooked = [p.grad.clone() for p in net.parameters() if p is not unused] + [torch.zeros_like(unused)]
        ref_net_grads = torch.autograd.grad(net(x2).pow(2).sum(), [prm for prm in net.parameters() if prm is not unused])
        # frozen module: nothing to exchange, nothing breaks
        for prm in net.parameters():
            prm.requires_grad = False
        sync.all_reduce()
        for prm in net.parameters():
            prm.requires_grad = True
        red = D.reduce_sum(torch.tensor(float(rank + 1)))
        losses = D.reduce_loss_dict({'g': torch.tensor(1.0 + rank), 'd': torch.tensor(10.0 * (rank + 1))})
        q.put((rank, [w.numpy() for w in w0], [None if g is None else g.numpy() for g in local],
               [g.numpy() for g in first_sync], float(red), {k: float(v) for k, v in losses.items()},
               [g.numpy() for g in hooked], [g.numpy() for g in ref_net_grads]))
    finally:
        dist.destroy_process_group()


def test_gradient_sync_and_scalar_collectives_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w_a, loc_a, syn_a, red_a, loss_a, hk_a, rf_a), (_, w_b, loc_b, syn_b, red_b, loss_b, hk_b, rf_b) = res
    # hook-driven step: both ranks hold the mean of the two local gradients (last entry = the unused parameter: zeros)
    for i, (ra, rb) in enumerate(zip(rf_a, rf_b)):
        mean = (ra + rb) / 2
        assert abs(hk_a[i] - mean).max() < 1e-6 and abs(hk_b[i] - mean).max() < 1e-6
    assert (hk_a[-1] == 0).all() and (hk_b[-1] == 0).all()
    for a, b in zip(w_a, w_b):                                 # broadcast from rank 0
        assert (a == b).all()
    for la, lb, sa, sb in zip(loc_a, loc_b, syn_a, syn_b):     # mean of the local gradients, on both ranks
        if la is None:
            assert (sa == 0).all() and (sb == 0).all()
            continue
        mean = (la + lb) / 2
        assert abs(sa - mean).max() < 1e-6 and abs(sb - mean).max() < 1e-6
    assert red_a == red_b == 3.0
    assert loss_a == {'d': 15.0, 'g': 1.5}                     # rank 0 holds the mean, sorted keys


class _DeferredWork:
    """Stand-in for an asynchronous RCCL work object: the reduction only happens when wait() is called, on whatever the
    buffer holds AT THAT TIME.  Reading the buffer before wait(), or writing it between launch and wait(), gives wrong
    results, which is exactly what must not happen on a backend whose collectives really are asynchronous."""
    log = []

    def __init__(self, real_all_reduce, tensor, op):
        self.real, self.tensor, self.op, self.done = real_all_reduce, tensor, op, False
        self.snapshot = tensor.clone()
        _DeferredWork.log.append(self)

    def wait(self):
        assert not self.done
        assert torch.equal(self.tensor, self.snapshot), 'bucket modified between launch and wait()'
        self.real(self.tensor, op=self.op)
        self.done = True


def _worker_deferred(rank, world, port, q):
    import torch.distributed as dist
    from transeditor_amd.utils import distributed as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        real = dist.all_reduce

        def fake_all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=False, **kw):
            if async_op:
                return _DeferredWork(real, tensor, op)
            return real(tensor, op=op)
        D.dist.all_reduce = fake_all_reduce
        torch.manual_seed(100 + rank)
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                                  torch.nn.Linear(16, 4))
        D.dist.all_reduce = real
        D.broadcast_module(net)
        D.dist.all_reduce = fake_all_reduce
        sync = D.GradSync(net, bucket_bytes=512)
        assert len(sync.buckets) >= 3
        outs = []
        for step in range(3):                                   # several steps: buffers are reused, .grad aliases the buckets
            for prm in net.parameters():
                prm.grad = None
            x = torch.randn(6, 8)
            loss = net(x).pow(2).sum()
            ref = torch.autograd.grad(loss, list(net.parameters()), retain_graph=True)
            n_before = len(_DeferredWork.log)
            loss.backward()
            launched = _DeferredWork.log[n_before:]
            assert launched and not any(w.done for w in launched), 'buckets must launch from the hooks and stay pending'
            sync.all_reduce()
            assert all(w.done for w in _DeferredWork.log)
            outs.append(([g.numpy() for g in ref], [prm.grad.clone().numpy() for prm in net.parameters()]))
        # second pass WITHOUT resetting .grad (zero_grad(set_to_none=False) style): autograd accumulates into the bucket views
        for prm in net.parameters():
            prm.grad.zero_()
        x = torch.randn(6, 8)
        loss = net(x).pow(2).sum()
        ref = torch.autograd.grad(loss, list(net.parameters()), retain_graph=True)
        loss.backward()
        sync.all_reduce()
        outs.append(([g.numpy() for g in ref], [prm.grad.clone().numpy() for prm in net.parameters()]))
        # disabled: a no-op
        sync.enabled = False
        for prm in net.parameters():
            prm.grad = None
        net(x).pow(2).sum().backward()
        sync.all_reduce()
        local_only = [prm.grad.clone().numpy() for prm in net.parameters()]
        q.put((rank, outs, [g.numpy() for g in ref], local_only, sync.bytes_per_call()))
    finally:
        dist.destroy_process_group()


def test_gradsync_with_deferred_async_work_world2():
    """GradSync's ordering must not depend on gloo completing collectives eagerly (RCCL does not)."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_deferred, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, outs_a, ref_a, loc_a, nb_a), (_, outs_b, ref_b, loc_b, nb_b) = res
    assert nb_a == nb_b == 4 * (8 * 16 + 16 + 16 * 16 + 16 + 16 * 4 + 4)          # (every tensor here is a multiple of 4 floats)
    for (ra, ga), (rb, gb) in zip(outs_a, outs_b):
        for x, y, ma, mb in zip(ra, rb, ga, gb):
            mean = (x + y) / 2
            assert abs(ma - mean).max() < 1e-6 and abs(mb - mean).max() < 1e-6
            assert (ma == mb).all()                              # bit-identical on both ranks
    for x, l in zip(ref_a, loc_a):                               # disabled sync leaves the local gradient untouched
        assert (x == l).all()


class _Branchy(torch.nn.Module):
    """`extra` sits UPSTREAM of the linear layer (its gradient is the last one backward produces) and is only used when asked"""

    def __init__(self):
        super().__init__()
        self.extra = torch.nn.Parameter(torch.ones(8))           # registered first -> same bucket as the layer, reported last
        self.lin = torch.nn.Linear(8, 4)

    def forward(self, x, use_extra):
        return self.lin(x * self.extra if use_extra else x)


def _worker_masks(rank, world, port, q):
    import torch.distributed as dist
    from transeditor_amd.utils import distributed as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(7)
        net = _Branchy()
        D.broadcast_module(net)
        sync = D.GradSync(net)                                   # one bucket
        assert len(sync.buckets) == 1
        out = {}

        def run(tag, use_extra, seed):
            for prm in net.parameters():
                prm.grad = None
            torch.manual_seed(seed + rank)
            x = torch.randn(5, 8)
            loss = net(x, use_extra).pow(2).sum()
            ref = torch.autograd.grad(loss, [prm for prm in net.parameters()], retain_graph=True, allow_unused=True)
            loss.backward()
            sync.all_reduce(tag)
            return [None if g is None else g.numpy() for g in ref], [None if prm.grad is None else prm.grad.clone().numpy()
                                                                      for prm in net.parameters()]
        # kind 'a': `extra` unused on every rank -> keeps .grad None, learnt as unused (the hooks stop waiting for it)
        for i in range(3):
            out[f'a{i}'] = run('a', False, 10 * i)
        assert net.extra.grad is None and net.extra in sync._unused
        # kind 'b': used on every rank, but the bucket has already left when its gradient arrives -> straggler path
        out['b0'] = run('b', True, 100)
        assert net.extra not in sync._unused
        out['b1'] = run('b', True, 110)
        out['b2'] = run('b', True, 120)                          # past the warm-up: learnt mask, no exchange
        # kind 'c': used on rank 0 ONLY -> every rank must still get the averaged gradient (DDP's used-bitmap semantics)
        out['c0'] = run('c', rank == 0, 200)
        out['c1'] = run('c', rank == 0, 210)
        out['c2'] = run('c', rank == 0, 220)
        # a gradient nobody announced during the warm-up of its kind is refused instead of silently diverging
        err = None
        try:
            run('a', True, 300)
        except RuntimeError as e:
            err = str(e)
        q.put((rank, out, err))
    finally:
        dist.destroy_process_group()


def test_gradsync_global_used_mask_and_stragglers_world2():
    """`.grad` must not be decided from rank-local information.  A parameter used on one rank only gets the
    averaged gradient everywhere; a parameter thought unused whose gradient arrives after its bucket was launched is reduced
    as a straggler on every rank (same collectives everywhere); an unannounced gradient after the warm-up raises."""
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_masks, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, oa, ea), (_, ob, eb) = res
    assert ea and eb and 'warm-up' in ea
    for key in oa:
        (ra, ga), (rb, gb) = oa[key], ob[key]
        for i, (x, y, ma, mb) in enumerate(zip(ra, rb, ga, gb)):
            if x is None and y is None:
                assert ma is None and mb is None, key             # unused everywhere: .grad stays None
                continue
            mean = ((0 if x is None else x) + (0 if y is None else y)) / 2
            assert ma is not None and mb is not None, (key, i)
            assert np.abs(ma - mean).max() < 1e-6 and np.abs(mb - mean).max() < 1e-6, (key, i)
            assert (ma == mb).all(), (key, i)


class _TwoOrders(torch.nn.Module):
    """two parameters of one bucket whose gradients arrive in an order the caller chooses (autograd runs the younger branch first)"""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.ones(4))
        self.b = torch.nn.Parameter(torch.ones(4))

    def forward(self, x, use_b, b_first):
        if not use_b:
            return (x * self.a).sum()
        if b_first:                       # b's branch is the younger one: its gradient is reported first
            ta = (x * self.a).sum()
            return ta + (2 * x * self.b).sum()
        tb = (2 * x * self.b).sum()
        return tb + (x * self.a).sum()


def _worker_mixed_straggler(rank, world, port, q):
    import torch.distributed as dist
    from transeditor_amd.utils import distributed as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        net = _TwoOrders()
        sync = D.GradSync(net)
        assert len(sync.buckets) == 1

        def run(tag, use_b, b_first, seed):
            for prm in net.parameters():
                prm.grad = None
            torch.manual_seed(seed + rank)
            x = torch.randn(4)
            loss = net(x, use_b, b_first)
            ref = torch.autograd.grad(loss, [net.a, net.b], retain_graph=True, allow_unused=True)
            loss.backward()
            late = [p is net.b for p in sync._late]
            sync.all_reduce(tag)
            return ([None if g is None else g.numpy() for g in ref],
                    [None if prm.grad is None else prm.grad.clone().numpy() for prm in net.parameters()], late)
        out = {}
        for i in range(3):                # kind 'x': b unused everywhere -> learnt as unused, the hooks stop waiting for it
            out[f'x{i}'] = run('x', False, False, 10 * i)
        assert net.b in sync._unused
        # kind 'y': b used on both ranks; on rank 0 its gradient is ON TIME (reported before a's, packed into the bucket), on
        # rank 1 it is LATE (a's gradient launches the bucket first): rank 0's share must not be lost
        out['y0'] = run('y', True, rank == 0, 100)
        assert out['y0'][2] == ([] if rank == 0 else [True]), out['y0'][2]
        out['y1'] = run('y', True, rank == 0, 110)
        q.put((rank, {k: v[:2] for k, v in out.items()}))
    finally:
        dist.destroy_process_group()


def test_gradsync_straggler_on_one_rank_keeps_the_on_time_share_world2():
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_mixed_straggler, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, oa), (_, ob) = res
    for key in oa:
        (ra, ga), (rb, gb) = oa[key], ob[key]
        for i, (x, y, ma, mb) in enumerate(zip(ra, rb, ga, gb)):
            if x is None and y is None:
                assert ma is None and mb is None, key
                continue
            mean = (x + y) / 2
            assert np.abs(ma - mean).max() < 1e-6 and np.abs(mb - mean).max() < 1e-6, (key, i, ma, mb, mean)
            assert (ma == mb).all(), (key, i)
