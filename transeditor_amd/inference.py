"""Inference-only pipeline for the sampling loops that consume the generator.

Reference consumers: test_spatial_query.py:20-31 (``sample_generation``: loop_num x g_ema(sample_z, sample_param)),
our_interfaceGAN/edit_all_noinversion_ffhq.py:103-130 (150 k samples in batches), metrics/fid_query.py:23-41.  They all
call a FROZEN generator under no_grad with a fixed batch shape, thousands of times.  Per call the training path
re-packs every weight, recomputes the demodulation squares and issues ~500 small launches from Python; at sampling batch
sizes the GPU then waits for the host.  ``GeneratorSampler``:

* keeps the packed weights and demodulation squares of every layer (op.modconv.frozen_weights), validated against each
  parameter's version counter;
* captures the whole forward of a given (batch, keyword) signature into ONE hipGraph (torch.cuda.CUDAGraph: HIP stream
  capture of the te_* launches on torch's capture stream) and replays it: one host call per batch instead of ~500.

Same call surface as ``Generator.forward``; results are bit-identical to the eager frozen forward.

Staleness contract.  Cached weights and captured graphs are validated against every parameter's (version counter, storage
address) on each call.  A write that goes around autograd's bookkeeping is invisible to that check - the reference's own
``accumulate()`` updates ``g_ema`` through ``.data`` (train_spatial_query.py:56-61) - so a sampler over a generator that is
updated that way MUST call ``refresh()`` after the update (this repository's ``MultiTensorEMA`` / ``FusedAdam`` bump the
counters themselves).  Tensor-valued keyword arguments (a ``truncation_latent``, explicit ``noise``) are never baked into a
graph: such calls run eagerly (with the weight cache), so neither a fresh tensor per call nor an in-place edit of one can
grow memory or replay stale values; the number of captured signatures is bounded (``max_graphs``, oldest dropped first).
"""
import torch

from .op.modconv import frozen_weights


class GeneratorSampler:
    def __init__(self, generator, use_graph=True, copy_outputs=True, max_graphs=8):
        """copy_outputs=False returns views of the graph's static output buffers (valid until the next call)."""
        self.g = generator.eval()
        self.use_graph = use_graph
        self.copy_outputs = copy_outputs
        self.max_graphs = max_graphs
        self._cache = {}
        self._graphs = {}
        self._stamp = None

    def refresh(self):
        """forget cached weights and captured graphs (after load_state_dict / an EMA update that used `.data`)"""
        self._cache.clear()
        self._graphs.clear()

    def _weights_stamp(self):
        return tuple((p._version, p.data_ptr()) for p in self.g.parameters())

    @torch.no_grad()
    def eager(self, style, op_param, **kw):
        with frozen_weights(self._cache):
            return self.g(style, op_param, **kw)

    @staticmethod
    def _flatten(out):
        if torch.is_tensor(out):
            return [out], lambda ts: ts[0]
        idx = [i for i, t in enumerate(out) if torch.is_tensor(t)]
        return [out[i] for i in idx], lambda ts: tuple(ts[idx.index(i)] if i in idx else None for i in range(len(out)))

    @torch.no_grad()
    def __call__(self, style, op_param, **kw):
        if kw.get('noise') is not None or (self.g.layer_noise_injection and kw.get('randomize_noise', True)):
            return self.eager(style, op_param, **kw)           # per-call noise tensors: not a replayable graph
        if not self.use_graph or any(torch.is_tensor(v) or isinstance(v, (list, tuple, dict)) for v in kw.values()):
            return self.eager(style, op_param, **kw)           # tensor-valued keywords are never captured (see the module docstring)
        stamp = self._weights_stamp()
        if stamp != self._stamp:                                # weights moved or were updated through autograd-visible ops
            self.refresh()
            self._stamp = stamp
        key = (tuple(style.shape), tuple(op_param.shape), style.device, tuple(sorted(kw.items())))
        ent = self._graphs.get(key)
        if ent is None:
            zs, ps = style.clone(), op_param.clone()
            for _ in range(2):                                  # warm-up on a side stream: fills the weight cache, sets kernel attributes
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self.eager(zs, ps, **kw)
                torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.eager(zs, ps, **kw)
            flat, rebuild = self._flatten(out)
            while len(self._graphs) >= self.max_graphs:        # bounded: the oldest signature goes first
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = (graph, zs, ps, flat, rebuild)
        graph, zs, ps, flat, rebuild = ent
        zs.copy_(style)
        ps.copy_(op_param)
        graph.replay()
        return rebuild([t.clone() for t in flat] if self.copy_outputs else flat)
