"""Multi-tensor Adam and EMA on the MI355X kernels (te_mt_adam_f32 / te_mt_ema_f32): one launch per optimiser step.

Reference: train_spatial_query.py:458-473 (``optim.Adam(generator.parameters(), lr=lr * ratio, betas=(0 ** ratio,
0.99 ** ratio))`` for G and D) and ``accumulate`` (:56-61).  ``FusedAdam`` is a ``torch.optim.Optimizer`` with exactly
``torch.optim.Adam``'s state layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), so ``state_dict()`` /
``load_state_dict()`` interchange with the reference's ``g_optim`` / ``d_optim`` checkpoints (:311-317).

How a step works: the (static) parameter / state pointers and the (per-step) gradient pointers go into one int64 table
that is uploaded through a double-buffered pinned host buffer with an asynchronous copy (no host-device synchronisation:
the host keeps running ahead of the GPU), then ONE kernel walks every tensor in fixed-size chunks.  Gradients may live
anywhere (fresh autograd tensors, or slices of the GradSync buckets after the all-reduce).
"""
import torch

from . import _lib

CHUNK = 8192          # elements per block (256 threads x 8 x 16 bytes)


class _Table:
    """Device-resident int64 pointer table [rows, n] + the static int32 chunk map [2, n_chunks] for a tensor list."""

    def __init__(self, numels, rows, device):
        self.n, self.rows, self.device = len(numels), rows, device
        t_idx, c_idx = [], []
        for t, ne in enumerate(numels):
            nc = max(1, -(-ne // CHUNK))
            t_idx += [t] * nc
            c_idx += list(range(nc))
        self.n_chunks = len(t_idx)
        self.chunks = torch.tensor([t_idx, c_idx], dtype=torch.int32).to(device)
        self.dev = torch.zeros(rows, self.n, dtype=torch.int64, device=device)
        self._host = [torch.zeros(rows, self.n, dtype=torch.int64).pin_memory() for _ in range(2)]
        self._evt = [None, None]
        self._turn = 0
        self._last = None

    def upload(self, rows):
        """rows: list of `self.rows` python-int lists.  Skips the copy when nothing changed."""
        if rows == self._last:
            return
        i = self._turn
        self._turn ^= 1
        if self._evt[i] is not None:
            self._evt[i].synchronize()               # the copy that last read this pinned buffer (two uploads ago) is done
        self._host[i].copy_(torch.tensor(rows, dtype=torch.int64))
        self.dev.copy_(self._host[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._evt[i] = ev
        self._last = rows


def _check_param(p):
    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
        raise RuntimeError(f'te_hip: FusedAdam / EMA need contiguous fp32 parameters on the GPU, got {p.dtype} {p.device} '
                           '(no CPU path exists)')


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no weight decay, no amsgrad, not maximize) as one kernel launch per parameter group."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, maximize=False, **other):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError(f'invalid Adam hyper-parameters lr={lr} betas={betas} eps={eps}')
        if weight_decay or amsgrad or maximize:
            raise NotImplementedError('FusedAdam covers the reference\'s configuration: no weight decay / amsgrad / maximize')
        # param_groups carry every key torch.optim.Adam's do, so a state_dict written by either loads into the other
        defaults = dict(torch.optim.Adam([torch.zeros(1)]).defaults)
        defaults.update(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False)
        super().__init__(params, defaults)
        self._tables = {}

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = torch.tensor(0.0, dtype=torch.float32)          # torch.optim.Adam's layout (host scalar tensor)
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        else:
            self._host_step(st)
        return st

    @staticmethod
    def _host_step(st):
        """`step` as a host fp32 scalar tensor.  The reference's torch 1.7 Adam (environment.yaml:130) stores a python int, and
        `torch.load(map_location=device)` moves a tensor `step` to the GPU, where `.item()` would synchronise once per parameter
        and step: both are normalised here (once, at load time or at the first step after it)."""
        s = st.get('step')
        if s is not None and not (torch.is_tensor(s) and s.device.type == 'cpu' and s.dtype == torch.float32 and s.ndim == 0):
            st['step'] = torch.tensor(float(s), dtype=torch.float32)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            self._host_step(st)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = group['params']
            if not params:
                continue
            active = [p for p in params if p.grad is not None]
            if not active:
                continue
            for p in active:
                _check_param(p)
                g = p.grad
                if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                    raise RuntimeError('te_hip: FusedAdam needs dense fp32 gradients on the parameter\'s device')
                if not g.is_contiguous():
                    p.grad = g = g.contiguous()
                self._init_state(p)
            key = (gi, tuple(p.numel() for p in params))
            tab = self._tables.get(key)
            if tab is None or tab.device != params[0].device:
                tab = self._tables[key] = _Table([p.numel() for p in params], 5, params[0].device)
            beta1, beta2 = group['betas']
            # parameters of a group normally share one step count; if not (a parameter that skipped steps), one launch per count
            steps = {}
            for p in active:
                st = self.state[p]
                st['step'] += 1
                steps.setdefault(int(st['step'].item()), set()).add(p)
            for step, members in steps.items():
                rows = [[], [], [], [], []]
                for p in params:
                    on = p in members
                    st = self.state.get(p) if on else None
                    rows[0].append(p.data_ptr())
                    rows[1].append(p.grad.data_ptr() if on else 0)
                    rows[2].append(st['exp_avg'].data_ptr() if on else 0)
                    rows[3].append(st['exp_avg_sq'].data_ptr() if on else 0)
                    rows[4].append(p.numel())
                tab.upload(rows)
                _lib.mt_adam(tab.dev, tab.chunks, tab.n, tab.n_chunks, CHUNK, float(group['lr']), float(beta1), float(beta2),
                             float(group['eps']), step)
            # the kernel wrote the parameters through raw pointers: tell autograd / the frozen-weight caches (op/modconv.py)
            torch.autograd.graph.increment_version(active)
        return loss


class MultiTensorEMA:
    """dst <- decay * dst + (1 - decay) * src over all parameters, one launch (reference accumulate(), :56-61)."""

    def __init__(self, dst_module, src_module):
        d, s = dict(dst_module.named_parameters()), dict(src_module.named_parameters())
        self.pairs = [(d[k], s[k]) for k in d.keys()]                    # KeyError on a structure mismatch, like the reference
        self._table = None

    @torch.no_grad()
    def update(self, decay):
        if not self.pairs:
            return
        for a, b in self.pairs:
            _check_param(a)
            _check_param(b)
            if a.shape != b.shape:
                raise RuntimeError(f'EMA: parameter shapes differ {tuple(a.shape)} vs {tuple(b.shape)}')
        dev = self.pairs[0][0].device
        if self._table is None or self._table.device != dev:
            self._table = _Table([a.numel() for a, _ in self.pairs], 3, dev)
        t = self._table
        t.upload([[a.data_ptr() for a, _ in self.pairs], [b.data_ptr() for _, b in self.pairs],
                  [a.numel() for a, _ in self.pairs]])
        _lib.mt_ema(t.dev, t.chunks, t.n, t.n_chunks, CHUNK, decay)
        torch.autograd.graph.increment_version([a for a, _ in self.pairs])      # written through raw pointers
