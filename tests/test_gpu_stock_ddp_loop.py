"""The reference's OWN loop shape on our modules: what `PYTHONPATH=dropin python train_spatial_query.py`
executes under BASELINE configs[3] - our `Generator` / `Discriminator` wrapped in stock `torch.nn.parallel.DistributedDataParallel(
find_unused_parameters=True, broadcast_buffers=False)` (train_spatial_query.py:494-509) on an `nccl` (= RCCL) process group, stock
`torch.optim.Adam(lr * c, betas=(0 ** c, 0.99 ** c))` (:458-473), and the four sub-steps written exactly as :173-250 writes them:
no `second_order()` hint, no `GradSync`, no `FusedAdam`, no joint discriminator pass - DDP's reducer hooks, its unused-parameter
walk and its bucket views sit on our autograd nodes, and the double backward of R1 / path length runs through whatever our fused
nodes record when nobody told them a second differentiation was coming.

Compared with `TrainStep` (the route every other test takes) on the same weights and draws:
  * D step and G step (first order): bit-identical gradients (same kernels in the same order; `d_joint` off for the comparison);
  * R1 and path-length steps (second order): `TrainStep` announces the double backward (`second_order()`), the reference's loop does
    not, so the two build different graphs of the same function - every parameter gradient within 2e-4 (relative L2);
  * `.grad is None` exactly for the parameters DDP leaves unused (the noise-injection weights: 7 at 32 px, 13 at 256);
  * then two iterations with the real learning rate: stock Adam on DDP-reduced gradients keeps our parameters on FusedAdam's trajectory.
FOUND by this test: on torch >= 1.9 the reference's path-length step fails under its own DDP wrapper - the wrapper returns
`latents` and `fake_img` through an identity node (`_DDPSink`), so one is no longer an ancestor of the other - whatever model sits inside
(the reference pins torch 1.7).  The test first shows that stock behaviour, then applies `utils.distributed.legacy_ddp_outputs()` (what
the drop-in package does on import) and runs the loop.
The group has one rank (the box has one GPU): the all-reduce is RCCL's, the arithmetic is sum / 1.  Runs in a child process
with a timeout."""
import os
import socket

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
    """the draws of one iteration, in the order the loop asks for them (D, G, path), identical for both routes"""

    def __init__(self):
        self.n = 0

    def latents(self, n):
        from transeditor_amd import synth
        self.n += 1
        z, p = synth.latents(n, 7000 + self.n)
        return z.cuda(), p.cuda()

    def randn_like(self, t):
        from transeditor_amd import synth
        return synth.normal(tuple(t.shape), 'ddp.pl').to(t)


def _models():
    from transeditor_amd import synth
    from transeditor_amd.model_spatial_query import Discriminator, Generator
    from transeditor_amd.train_step import default_args
    args = default_args(size=SIZE, batch=BATCH)
    G = Generator(SIZE, 512, 512, args.token, n_trans=8, pixel_norm_op_dim=1)
    Dn = Discriminator(SIZE)
    synth.fill_state_dict(G.state_dict(), 40)
    synth.fill_state_dict(Dn.state_dict(), 41)
    return args, G.cuda(), Dn.cuda()


def _grads(mod):
    return [None if p.grad is None else p.grad.detach().clone() for p in mod.parameters()]


def _reference_loop(lr, real_img, iters):
    """train_spatial_query.py:166-250 on DDP-wrapped modules, statement for statement (names as there)"""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from transeditor_amd.train_step import (d_logistic_loss, d_r1_loss, g_nonsaturating_loss, g_path_regularize, requires_grad)
    args, g_module, d_module = _models()
    args.lr = lr
    g_reg_ratio = args.g_reg_every / (args.g_reg_every + 1)
    d_reg_ratio = args.d_reg_every / (args.d_reg_every + 1)
    g_optim = torch.optim.Adam(g_module.parameters(), lr=args.lr * g_reg_ratio, betas=(0 ** g_reg_ratio, 0.99 ** g_reg_ratio))
    d_optim = torch.optim.Adam(d_module.parameters(), lr=args.lr * d_reg_ratio, betas=(0 ** d_reg_ratio, 0.99 ** d_reg_ratio))
    generator = DistributedDataParallel(g_module, device_ids=[0], output_device=0, broadcast_buffers=False, find_unused_parameters=True)
    discriminator = DistributedDataParallel(d_module, device_ids=[0], output_device=0, broadcast_buffers=False, find_unused_parameters=True)
    assert dist.get_backend() == 'nccl'
    sampler = _Sampler()
    mean_path_length = 0
    out = {}
    for i in range(iters):
        real_img = real_img.detach()
        requires_grad(generator, False)
        requires_grad(discriminator, True)
        noise, operated_param = sampler.latents(args.batch)
        fake_img, _, _ = generator(noise, operated_param)
        fake_pred = discriminator(fake_img)
        real_pred = discriminator(real_img)
        d_loss = d_logistic_loss(real_pred, fake_pred)
        discriminator.zero_grad()
        d_loss.backward()
        out[f'd{i}'] = _grads(d_module)
        d_optim.step()

        if i % args.d_reg_every == 0:
            real_img.requires_grad = True
            real_pred = discriminator(real_img)
            r1_loss = d_r1_loss(real_pred, real_img)
            discriminator.zero_grad()
            (args.r1 / 2 * r1_loss * args.d_reg_every + 0 * real_pred[0]).backward()
            out[f'r1{i}'] = _grads(d_module)
            d_optim.step()

        requires_grad(generator, True)
        requires_grad(discriminator, False)
        noise, operated_param = sampler.latents(args.batch)
        fake_img, _, _ = generator(noise, operated_param)
        fake_pred = discriminator(fake_img)
        g_loss = g_nonsaturating_loss(fake_pred)
        generator.zero_grad()
        g_loss.backward()
        out[f'g{i}'] = _grads(g_module)
        g_optim.step()

        if i % args.g_reg_every == 0:
            path_batch_size = max(1, args.batch // args.path_batch_shrink)
            noise, operated_param = sampler.latents(path_batch_size)
            fake_img, latents, similarity = generator(noise, operated_param, return_latents=True)
            path_loss, mean_path_length, path_lengths = g_path_regularize(fake_img, latents, mean_path_length,
                                                                           sampler.randn_like(fake_img))
            generator.zero_grad()
            weighted_path_loss = args.path_regularize * args.g_reg_every * path_loss
            if args.path_batch_shrink:
                weighted_path_loss += 0 * fake_img[0, 0, 0, 0]
            weighted_path_loss.backward()
            out[f'path{i}'] = _grads(g_module)
            g_optim.step()
    torch.cuda.synchronize()
    params = {'g': [p.detach().clone() for p in g_module.parameters()], 'd': [p.detach().clone() for p in d_module.parameters()]}
    return out, params


def _train_step_route(lr, real_img, iters):
    from transeditor_amd.train_step import TrainStep
    args, G, Dn = _models()
    args.lr = lr
    args.d_joint = False                       # the reference runs the two discriminator passes separately (:190-191)
    ts = TrainStep(args, 'cuda', G, Dn, _Sampler())
    out = {}
    for i in range(iters):
        ts.d_step(real_img)
        out[f'd{i}'] = _grads(ts.discriminator)
        if i % args.d_reg_every == 0:
            ts.r1_step(real_img)
            out[f'r1{i}'] = _grads(ts.discriminator)
        ts.g_step()
        out[f'g{i}'] = _grads(ts.generator)
        if i % args.g_reg_every == 0:
            ts.path_step()
            out[f'path{i}'] = _grads(ts.generator)
    torch.cuda.synchronize()
    params = {'g': [p.detach().clone() for p in ts.generator.parameters()], 'd': [p.detach().clone() for p in ts.discriminator.parameters()]}
    return out, params


def _compare(a, b):
    """-> per key: (max relative-L2 difference over the parameters, all bit-identical?, None-pattern equal?, number of None grads)"""
    res = {}
    for key in a:
        worst, same, pat, nnone = 0.0, True, True, 0
        top = max(float(y.double().norm()) for y in b[key] if y is not None)
        for x, y in zip(a[key], b[key]):
            pat = pat and ((x is None) == (y is None))
            if x is None or y is None:
                nnone += x is None
                continue
            same = same and torch.equal(x, y)
            if float(y.double().norm()) <= 1e-6 * top:       # analytically zero (a key bias shifts every logit of a softmax row alike:
                continue                                      # interact.*.atten.k_transform.bias) - round-off on both sides
            d = float((x.double() - y.double()).norm())
            n = float(y.double().norm())
            worst = max(worst, d / n if n > 0 else d)
        res[key] = (worst, same, pat, int(nnone))
    return res


def _worker(port, q):
    import torch.distributed as dist
    from transeditor_amd import synth
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    torch.cuda.set_device(0)
    real = synth.normal((BATCH, 3, SIZE, SIZE), 'ddp.real').clamp(-1, 1).cuda()
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        # (0) torch's own behaviour first: the outputs of a find_unused_parameters wrapper pass an identity node, so the returned
        # latents are not what the returned image was computed from - the reference's path-length step cannot run on torch >= 1.9
        from torch.nn.parallel import DistributedDataParallel
        from transeditor_amd.train_step import g_path_regularize
        from transeditor_amd.utils.distributed import legacy_ddp_outputs
        _, g_module, _ = _models()
        wrapped = DistributedDataParallel(g_module, device_ids=[0], output_device=0, broadcast_buffers=False, find_unused_parameters=True)
        z, p = _Sampler().latents(2)
        img, lat, _ = wrapped(z, p, return_latents=True)
        try:
            g_path_regularize(img, lat, 0, torch.randn_like(img))
            stock_error = None
        except RuntimeError as e:
            stock_error = str(e)
        del wrapped, g_module, img, lat
        assert legacy_ddp_outputs()                      # what dropin/utils/distributed.py does on import
        # (1) learning rate 0: every sub-step of both routes sees the SAME weights, gradients comparable one by one
        ref0, _ = _reference_loop(0.0, real, 1)
        ts0, _ = _train_step_route(0.0, real, 1)
        cmp0 = _compare(ref0, ts0)
        # (2) the real learning rate, two iterations: stock Adam on DDP-reduced gradients against FusedAdam on ours
        _, pref = _reference_loop(0.002, real, 2)
        _, pts = _train_step_route(0.002, real, 2)
        pdiff = {}
        for net in ('g', 'd'):
            close, finite = [], True
            for x, y in zip(pref[net], pts[net]):
                finite = finite and bool(torch.isfinite(x).all())
                close.append(float(((x - y).abs() <= 1e-3 + 1e-2 * y.abs()).float().mean()))
            pdiff[net] = (min(close), finite)
        q.put((cmp0, pdiff, stock_error))
    finally:
        dist.destroy_process_group()


def test_reference_loop_under_stock_ddp_and_adam_matches_train_step():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    pr = ctx.Process(target=_worker, args=(_free_port(), q))
    pr.start()
    try:
        cmp0, pdiff, stock_error = q.get(timeout=600)
    finally:
        pr.join(timeout=60)
        if pr.is_alive():
            pr.kill()
    assert pr.exitcode == 0
    # torch >= 1.9: the unpatched wrapper breaks the reference's path-length step (whatever the model); legacy_ddp_outputs() repairs it
    assert stock_error is not None and 'not have been used in the graph' in stock_error, stock_error
    print('stock DDP loop vs TrainStep (max rel-L2 over parameters, bit-identical, same None pattern, unused):',
          {k: (f'{v[0]:.2e}', v[1], v[2], v[3]) for k, v in cmp0.items()})
    for key in ('d0', 'g0'):                     # first order: the same kernels in the same order
        assert cmp0[key][2], key
        assert cmp0[key][1], (key, cmp0[key])    # bit-identical
    for key in ('r10', 'path0'):                 # second order: two graphs of the same function
        assert cmp0[key][2], key
        assert cmp0[key][0] < 2e-4, (key, cmp0[key])
    assert cmp0['g0'][3] == 7 and cmp0['path0'][3] == 7      # DDP leaves the unused noise-injection weights (conv1 + 6 of convs at 32 px) without a gradient
    for net in ('g', 'd'):                       # (two separately evolving runs: first Adam steps move by lr * sign(g))
        assert pdiff[net][1], net
        assert pdiff[net][0] > 0.9, (net, pdiff)
