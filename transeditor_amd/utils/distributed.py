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
    Parameters that received no gradient on this rank (the unused `noise.weight`s; by symmetry unused on every rank)
    contribute zeros to the bucket and keep `.grad = None`, which is what DistributedDataParallel with
    find_unused_parameters=True leaves (train_spatial_query.py:494-509), so optimiser state is created for exactly the
    same parameters as in the reference.

    Nothing here relies on the collective being synchronous: between `_launch` and `work.wait()` the flat buffer is
    neither read nor written by this class (tests/test_distributed_gloo.py drives it with deferred work objects).
    """

    def __init__(self, module, bucket_bytes=32 << 20):
        self.params = [p for p in module.parameters()]
        self.enabled = True                               # False: hooks and all_reduce() do nothing (bench: no-exchange leg)
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
        self._size = []
        for bi, bucket in enumerate(self.buckets):
            off = 0
            for p in bucket:
                self._slot[p] = (bi, off)
                off += (p.numel() + 3) & ~3               # 16-byte aligned slots: the optimiser kernel reads them with 16-byte loads
            self._size.append(off)
        self._seen = [set() for _ in self.buckets]
        self._work = [None] * len(self.buckets)
        # parameters observed to receive no gradient (the generator's `noise.weight`s): learnt at the first
        # all_reduce() and no longer waited for, so their buckets can still launch from the hooks
        self._unused = set()
        self._late = []                                   # gradients that arrived after their bucket was launched
        self._handles = []
        self.calls = 0
        if get_world_size() > 1:
            self._install_hooks()

    # ---- bookkeeping
    def bytes_per_call(self):
        """bytes this rank hands to the all-reduce in one all_reduce() call (all buckets)"""
        return sum(n * self.params[0].element_size() for n in self._size)

    def remove_hooks(self):
        for h in self._handles:
            h.remove()
        self._handles = []

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
        expected = sum(1 for q in self.buckets[bi] if q.requires_grad and q not in self._unused)
        if len(self._seen[bi] - self._unused) >= expected:
            self._launch(bi)

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

    def all_reduce(self):
        world = get_world_size()
        if world == 1 or not self.enabled:
            return
        self.calls += 1
        active = [bi for bi, b in enumerate(self.buckets) if any(p.requires_grad for p in b)]
        had_grad = {}
        for bi in active:
            for p in self.buckets[bi]:
                had_grad[p] = p.grad is not None
                if p.requires_grad and p.grad is None:
                    self._unused.add(p)
                elif p in self._unused and p in self._seen[bi]:
                    self._unused.discard(p)               # it does get gradients after all
            if self._work[bi] is None:
                self._launch(bi)
        late = set(self._late)
        for bi in active:
            self._work[bi].wait()
            self._flat[bi].div_(world)
            for p, v in zip(self.buckets[bi], self._views(bi)):
                if p.requires_grad and p not in late and had_grad[p]:
                    p.grad = v                            # the averaged gradient lives in the bucket: no copy back
        for p in self._late:                              # rare: reduce stragglers one by one and stop treating them as unused
            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
            p.grad.div_(world)
            self._unused.discard(p)
        self._late.clear()
        for bi in range(len(self.buckets)):
            self._seen[bi].clear()
            self._work[bi] = None
