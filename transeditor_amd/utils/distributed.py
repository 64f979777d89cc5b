"""Data-parallel glue: rank helpers with the reference's names (utils/distributed.py:7-124) and a
bucketed gradient all-reduce for one-process-per-GPU training over RCCL/xGMI.

On PyTorch-ROCm the "nccl" backend IS RCCL.  The only exchange step of the TransEditor train step
is the gradient all-reduce (171.7 MB for G, 115.5 MB for D at 256 px) plus three
scalar-sized collectives (C5/C6).  `GradSync` replaces the two DistributedDataParallel wrappers
(`find_unused_parameters=True`, train_spatial_query.py:494-509): gradients are packed into a few
large flat fp32 buckets (sized for the per-link xGMI bandwidth, default 32 MiB => 6 buckets for G)
and all-reduced asynchronously on RCCL's stream; parameters whose .grad is None (the 13 unused
`noise.weight`s) contribute zeros and keep .grad None, which is what DDP's unused-parameter handling amounts to.
The same code runs on CPU tensors with the gloo backend (tests).
"""
import torch
from torch import distributed as dist


def _ready():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if _ready() else 0


def get_world_size():
    return dist.get_world_size() if _ready() else 1


def synchronize():
    if _ready() and dist.get_world_size() > 1:
        dist.barrier()


def reduce_sum(tensor):
    if not _ready():
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def gather_grad(params):
    world_size = get_world_size()
    if world_size == 1:
        return
    for param in params:
        if param.grad is not None:
            dist.all_reduce(param.grad.data, op=dist.ReduceOp.SUM)
            param.grad.data.div_(world_size)


def all_gather(data):
    """Every rank's picklable `data` as a list in rank order (utils/distributed.py:67-99: byte tensors padded to the longest
    payload, one size exchange + one all_gather).  The payload travels on the device the backend communicates from."""
    import pickle
    world_size = get_world_size()
    if world_size == 1:
        return [data]
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    payload = torch.frombuffer(bytearray(pickle.dumps(data)), dtype=torch.uint8).to(dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world_size)]
    dist.all_gather(sizes, torch.tensor([payload.numel()], dtype=torch.int64, device=dev))
    sizes = [int(s.item()) for s in sizes]
    buf = torch.zeros(max(sizes), dtype=torch.uint8, device=dev)
    buf[:payload.numel()] = payload
    parts = [torch.empty_like(buf) for _ in range(world_size)]
    dist.all_gather(parts, buf)
    return [pickle.loads(t[:n].cpu().numpy().tobytes()) for n, t in zip(sizes, parts)]


def reduce_loss_dict(loss_dict):
    """Mean of each loss over ranks, valid on rank 0 (reduce to dst=0), keys in sorted order."""
    world_size = get_world_size()
    if world_size < 2:
        return loss_dict
    with torch.no_grad():
        keys = sorted(loss_dict.keys())
        losses = torch.stack([loss_dict[k] for k in keys], 0)
        dist.reduce(losses, dst=0)
        if dist.get_rank() == 0:
            losses /= world_size
        return {k: v for k, v in zip(keys, losses)}


def broadcast_module(module, src=0):
    """Initial parameter + buffer broadcast from rank `src` (what the DDP constructor does, C2)."""
    if get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


class GradSync:
    """Bucketed mean all-reduce of `module`'s gradients, overlapped with the backward pass.

        sync = GradSync(generator)          # once
        loss.backward(); sync.all_reduce()  # every step, before optimizer.step()

    Parameters are bucketed in REVERSE registration order (the order backward produces their gradients).  A
    post-accumulate hook counts the gradients of a bucket; when the last one has arrived the bucket is packed into its
    flat buffer with one multi-tensor copy and its all-reduce is issued asynchronously, so RCCL traffic over xGMI overlaps
    the rest of the backward.  `all_reduce()` issues whatever is left, waits, averages and points every `.grad` at its
    slice of the bucket (no copy back; the buffers are reused by the next backward, after `.grad` has been reset).

    Unused parameters (the generator's `noise.weight`s): DistributedDataParallel with find_unused_parameters=True
    (train_spatial_query.py:494-509) reduces a used-bitmap, hands the averaged gradient to EVERY rank when ANY rank used the
    parameter and leaves `.grad = None` when none did.  Same here: during the first `MASK_CALLS` calls of every call kind
    (`tag`: 'd', 'r1', 'g', 'path', ...) a used-mask is all-reduced (MAX) and decides, identically on every rank, which
    parameters get `.grad`, which are no longer waited for by the hooks, and which stragglers (a gradient that arrived after
    its bucket had been launched) are reduced one by one - so every rank always issues the same collectives.  After the
    warm-up the learnt mask of the kind is used without communication; a rank whose local pattern then departs from it
    raises instead of silently diverging (`reset_masks()` re-learns).

    Nothing here relies on the collective being synchronous: between `_launch` and `work.wait()` the flat buffer is
    neither read nor written by this class (tests/test_distributed_gloo.py drives it with deferred work objects).
    `force=True` (or TE_GRADSYNC_FORCE=1) runs the whole machinery - hooks, buckets, collectives - on a group of ONE rank
    too: tests/test_gpu_rccl_world1.py uses it to put RCCL itself under this code on a single GPU.
    """
    MASK_CALLS = 2

    def __init__(self, module, bucket_bytes=32 << 20, force=None):
        import os
        self.params = [p for p in module.parameters()]
        self.enabled = True                               # False: hooks and all_reduce() do nothing (bench: no-exchange leg)
        self.force = (os.environ.get('TE_GRADSYNC_FORCE') == '1') if force is None else bool(force)
        self.buckets, cur, cur_bytes = [], [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)
        self._slot = {}                                   # param -> (bucket index, offset)
        self._index = {p: i for i, p in enumerate(self.params)}
        self._size = []
        for bi, bucket in enumerate(self.buckets):
            off = 0
            for p in bucket:
                self._slot[p] = (bi, off)
                off += (p.numel() + 3) & ~3               # 16-byte aligned slots: the optimiser kernel reads them with 16-byte loads
            self._size.append(off)
        self._seen = [set() for _ in self.buckets]
        self._work = [None] * len(self.buckets)
        # parameters no rank gave a gradient in any learnt call kind (the generator's `noise.weight`s): no longer waited
        # for, so their buckets can still launch from the hooks
        self._unused = set()
        self._masks = {}                                  # tag -> (calls seen, learnt global used-mask as a tuple of bools)
        self._late = []                                   # gradients that arrived after their bucket was launched
        self._next = 0                                    # next bucket to launch (collectives go out in bucket order)
        self._handles = []
        self.calls = 0
        if self.active():
            self._install_hooks()

    # ---- bookkeeping
    def active(self):
        return get_world_size() > 1 or (self.force and _ready())

    def bytes_per_call(self):
        """bytes this rank hands to the all-reduce in one all_reduce() call (all buckets)"""
        return sum(n * self.params[0].element_size() for n in self._size)

    def remove_hooks(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    def reset_masks(self):
        """forget the learnt used-masks (the next MASK_CALLS calls of every kind exchange them again)"""
        self._masks.clear()
        self._unused.clear()

    # ---- internals
    def _buffer(self, bi):
        bucket = self.buckets[bi]
        n = self._size[bi]
        flat = self._flat[bi]
        if flat is None or flat.numel() != n or flat.device != bucket[0].device:
            flat = self._flat[bi] = torch.zeros(n, device=bucket[0].device, dtype=bucket[0].dtype)     # (alignment gaps stay zero)
        return flat

    def _install_hooks(self):
        if self._handles:
            return
        for p in self.params:
            self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p):
        if not self.enabled:
            return
        bi, _ = self._slot[p]
        if self._work[bi] is not None:                    # bucket already in flight (parameter thought unused): fix up later
            self._late.append(p)
            return
        self._seen[bi].add(p)
        self._launch_ready()

    def _launch_ready(self, flush=False):
        """Launch buckets strictly in index order (every rank issues the same sequence of collectives, whatever the order or
        the rank-local absence of gradients): the next bucket goes as soon as every parameter it still waits for has
        reported; `flush` (from all_reduce) sends whatever is left."""
        while self._next < len(self.buckets):
            bi = self._next
            bucket = self.buckets[bi]
            if any(p.requires_grad for p in bucket):      # (a frozen bucket is skipped on every rank alike)
                if not flush and any(q.requires_grad and q not in self._unused and q not in self._seen[bi] for q in bucket):
                    return
                self._launch(bi)
            self._next += 1

    def _views(self, bi):
        flat = self._buffer(bi)
        return [flat[self._slot[p][1]:self._slot[p][1] + p.numel()].view_as(p) for p in self.buckets[bi]]

    def _launch(self, bi):
        """Pack the bucket with ONE multi-tensor copy (not one launch per parameter) and start its all-reduce."""
        flat = self._buffer(bi)
        views = self._views(bi)
        src, dst = [], []
        for p, v in zip(self.buckets[bi], views):
            if p.grad is None:                            # no gradient (unused or frozen): contributes zeros
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():       # (a .grad still aliasing its slice is already in place)
                src.append(p.grad)
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def _global_mask(self, tag, local):
        """(used-mask over ALL ranks for this call, exchanged now?): exchanged during the warm-up calls of the kind, the
        learnt one afterwards."""
        seen, learnt = self._masks.get(tag, (0, None))
        if seen < self.MASK_CALLS:
            m = torch.tensor([1.0 if u else 0.0 for u in local], device=self.params[0].device)
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
            glob = tuple(v > 0.5 for v in m.tolist())     # (host read-back: warm-up calls only)
            # a parameter used in any warm-up call of the kind counts as used from then on
            self._masks[tag] = (seen + 1, glob if learnt is None else tuple(a or b for a, b in zip(glob, learnt)))
            return glob, True
        bad = [i for i, (u, g) in enumerate(zip(local, learnt)) if u and not g]
        if bad:
            raise RuntimeError(f'GradSync: parameter #{bad[0]} received a gradient in a "{tag}" call although no rank used it '
                               f'during the warm-up calls; the replicas would diverge (call reset_masks() when the graph changes)')
        return learnt, False

    def all_reduce(self, tag='default'):
        if not self.active() or not self.enabled:
            return
        world = get_world_size()
        self.calls += 1
        active = [bi for bi, b in enumerate(self.buckets) if any(p.requires_grad for p in b)]
        if not active:
            return
        # fixed order of collectives on every rank: the buckets (in index order; some already left from the hooks), then the
        # used-mask, then - only while a kind is being learnt - the straggler flags and the stragglers themselves
        self._launch_ready(flush=True)
        local = [p.requires_grad and (p.grad is not None) for p in self.params]
        glob, exchanged = self._global_mask(tag, local)
        thought_unused = self._unused
        # unused on every rank in every kind learnt so far -> the hooks stop waiting for it
        every = [m for _, m in self._masks.values()]
        self._unused = {p for i, p in enumerate(self.params) if not any(m[i] for m in every)}
        late_local = set(self._late)
        for bi in active:
            self._work[bi].wait()
            self._flat[bi].div_(world)
            for p, v in zip(self.buckets[bi], self._views(bi)):
                if p.requires_grad and glob[self._index[p]] and p not in late_local:
                    p.grad = v                            # the averaged gradient lives in the bucket: no copy back
        # Stragglers: a parameter thought unused whose gradient arrived after its bucket had been launched.  Only possible
        # while a kind is being learnt (afterwards _global_mask raises).  The candidates - thought unused, used by some rank
        # now - are the same list on every rank, so the flag exchange and the one-by-one reductions are issued everywhere.
        cands = [p for p in self.params if p in thought_unused and p.requires_grad and glob[self._index[p]]] if exchanged else []
        if cands:
            flags = torch.tensor([1.0 if p in late_local else 0.0 for p in cands], device=self.params[0].device)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
            for p, f in zip(cands, flags.tolist()):
                if f > 0.5:
                    # ranks on which the gradient was on time have their share in the bucket already (a late rank packed zeros
                    # there): reduce the late shares only and add the bucket's averaged slice, identical on every rank
                    bi, off = self._slot[p]
                    share = self._flat[bi][off:off + p.numel()].view_as(p)
                    g = p.grad if p in late_local else torch.zeros_like(p)
                    dist.all_reduce(g, op=dist.ReduceOp.SUM)
                    p.grad = g.div_(world).add_(share)
        elif late_local:
            raise RuntimeError('GradSync: a gradient arrived after its bucket was launched outside a warm-up call')
        self._late.clear()
        self._next = 0
        for bi in range(len(self.buckets)):
            self._seen[bi].clear()
            self._work[bi] = None


def legacy_ddp_outputs(enable=True):
    """Compatibility switch for running the reference's UNCHANGED loop (train_spatial_query.py:494-509 wraps both networks in
    `DistributedDataParallel(find_unused_parameters=True)`) on a current PyTorch.

    Since torch 1.9 such a wrapper returns its outputs through an identity autograd node (`_DDPSink`): the `latents` handed back
    next to `fake_img` are then NEW tensors which `fake_img` does not descend from, and the reference's path-length regulariser -
    `autograd.grad((fake_img * noise).sum(), latents, create_graph=True)` on the outputs of the WRAPPED generator (:226-232,
    g_path_regularize :92-105) - raises "One of the differentiated Tensors appears to not have been used in the graph", whatever
    model sits inside the wrapper (found by tests/test_gpu_stock_ddp_loop.py; the reference pins torch 1.7, environment.yaml:130,
    which has no sink).  `legacy_ddp_outputs()` restores the 1.7 behaviour - outputs returned as the module produced them - by
    replacing that node with a pass-through; `dropin/utils/distributed.py` calls it on import (TE_DROPIN_KEEP_DDP_SINK=1 keeps
    torch's behaviour).  The product's own loop (TrainStep + GradSync) does not use DistributedDataParallel and is unaffected.
    Returns True when the switch took effect."""
    import inspect
    import torch.nn.parallel.distributed as ddp
    if not hasattr(ddp, '_DDPSink'):
        return False
    if not hasattr(ddp, '_te_original_sink'):
        ddp._te_original_sink = ddp._DDPSink
    orig = ddp._te_original_sink
    if not enable:
        ddp._DDPSink = orig
        return True
    # Only the call convention this was written against: `_DDPSink.apply(weakref.ref(ddp_module), *outputs)` (torch >= 2.0).
    # torch 1.9 - 1.13 call `apply(reducer, state_dict, *outputs)`: a pass-through would hand `state_dict` back as the first
    # output, so refuse there and leave torch's behaviour in place.
    try:
        names = list(inspect.signature(orig.forward).parameters)
    except (TypeError, ValueError):
        names = []
    if names != ['ctx', 'ddp_weakref', 'inputs']:
        return False

    class _ScopedPassThrough:
        """pass-through ONLY for the wrappers the reference builds - `find_unused_parameters=True`, no static graph (train_spatial_query.py:
        494-509).  Every other wrapper in the process keeps torch's sink: a `static_graph=True` wrapper enqueues its delayed all-reduce
        from `_DDPSink.backward` on the first iteration and would silently skip the gradient exchange without it.
        What the pass-through gives up for the wrappers it does apply to: an output that does not reach the loss no longer sends an
        undefined gradient to the reducer, so parameters reachable ONLY from such an output are not marked ready (torch 1.7 behaved
        the same).  In the reference loop every output of G / D descends from the one graph that produces the image / the logit."""
        @staticmethod
        def apply(ddp_weakref, *inputs):
            module = ddp_weakref() if callable(ddp_weakref) else None
            if (module is None or getattr(module, 'static_graph', False) or not getattr(module, 'find_unused_parameters', False)
                    or getattr(module, '_te_keep_ddp_sink', False)):
                return orig.apply(ddp_weakref, *inputs)
            return inputs

    ddp._DDPSink = _ScopedPassThrough
    return True


class legacy_ddp_outputs_scope:
    """`with legacy_ddp_outputs_scope(): ...` - the same switch around a block (the reference loop), torch's behaviour restored on
    exit; `.active` says whether it took effect."""

    def __enter__(self):
        import torch.nn.parallel.distributed as ddp
        self._before = getattr(ddp, '_DDPSink', None)
        self.active = legacy_ddp_outputs(True)
        return self

    def __exit__(self, *exc):
        import torch.nn.parallel.distributed as ddp
        if self._before is not None:
            ddp._DDPSink = self._before
        return False
