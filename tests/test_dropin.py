"""The "drop in unchanged" claim, checked mechanically: with `dropin/` in front of the reference tree on
PYTHONPATH every `from model_spatial_query | utils.* import X` of the reference's training / test scripts resolves to this
package, and every call the scripts make on a Generator / Discriminator (constructor and forward, positional count and keyword
names) binds to the signatures here.  The scripts are only PARSED (ast) - nothing of the reference is executed or stored;
without /root/reference (GPU box) the reference-dependent part skips and the shim's own import surface is still checked."""
import ast
import inspect
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
SCRIPTS = ['train_spatial_query.py', 'test_spatial_query.py', 'model_spatial_query.py']
HOT = ('model_spatial_query', 'utils.op', 'utils.sample', 'utils.distributed', 'utils.dataset')

_RESOLVE = r'''
import importlib, json, sys
wanted = json.loads(sys.argv[1])
out = {}
for mod, names in wanted.items():
    m = importlib.import_module(mod)
    out[mod] = {'file': getattr(m, '__file__', None), 'missing': [n for n in names if not hasattr(m, n)]}
import utils
out['__utils_path__'] = list(utils.__path__)
print(json.dumps(out))
'''


def _resolve(wanted, with_reference):
    path = [os.path.join(ROOT, 'dropin'), ROOT] + ([REF] if with_reference else [])
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(path))
    r = subprocess.run([sys.executable, '-c', _RESOLVE, json.dumps(wanted)], env=env, capture_output=True, text=True, timeout=300,
                       cwd='/tmp')
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_shim_modules_export_the_hot_path_surface():
    """what the reference's scripts import from the hot path, resolvable without the reference present"""
    wanted = {'model_spatial_query': ['Generator', 'Discriminator', 'ModulatedConv2d', 'EqualLinear', 'StyledConv', 'ToRGB',
                                      'AttentionBlock', 'FusedLeakyReLU', 'fused_leaky_relu', 'upfirdn2d'],
              'utils.op': ['FusedLeakyReLU', 'fused_leaky_relu', 'upfirdn2d'],
              'utils.sample': ['prepare_param', 'prepare_noise_new'],
              'utils.distributed': ['get_rank', 'synchronize', 'reduce_loss_dict', 'reduce_sum', 'get_world_size', 'gather_grad',
                                    'all_gather'],
              'utils.dataset': ['MultiResolutionDataset']}
    got = _resolve(wanted, with_reference=False)
    for mod in wanted:
        assert got[mod]['missing'] == [], (mod, got[mod])
        assert got[mod]['file'].startswith(os.path.join(ROOT, 'dropin')), got[mod]


def _imports(tree):
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.level == 0:
            if node.module in HOT or node.module.startswith('utils.op'):
                out.setdefault(node.module, set()).update(a.name for a in node.names)
    return out


def _calls(tree, names):
    """calls `name(...)` for the given variable / class names -> (name, n_positional, keyword names, line)"""
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in names:
            if any(isinstance(a, ast.Starred) for a in node.args) or any(k.arg is None for k in node.keywords):
                continue
            yield node.func.id, len(node.args), [k.arg for k in node.keywords], node.lineno


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference tree only exists in the build container')
def test_reference_scripts_resolve_through_the_shim_and_bind_to_our_signatures():
    trees = {s: ast.parse(open(os.path.join(REF, s)).read()) for s in SCRIPTS}
    wanted = {}
    for s, tree in trees.items():
        if s == 'model_spatial_query.py':
            continue                                   # (replaced as a whole; its own `from utils.op import ...` is checked below)
        for mod, names in _imports(tree).items():
            wanted.setdefault(mod, set()).update(names)
    for mod, names in _imports(trees['model_spatial_query.py']).items():
        wanted.setdefault(mod, set()).update(names)
    assert {'model_spatial_query', 'utils.sample', 'utils.distributed', 'utils.dataset', 'utils.op'} <= set(wanted), wanted
    got = _resolve({m: sorted(n) for m, n in wanted.items()}, with_reference=True)
    for mod in wanted:
        assert got[mod]['missing'] == [], (mod, got[mod])
        assert got[mod]['file'].startswith(os.path.join(ROOT, 'dropin')), (mod, got[mod]['file'])
    # the rest of the reference's `utils` (lpips, editing utils ...) must stay reachable behind the shim package
    assert any(p.startswith(REF) for p in got['__utils_path__']), got['__utils_path__']

    from transeditor_amd.model_spatial_query import Discriminator, Generator
    sigs = {'generator': inspect.signature(Generator.forward), 'g_ema': inspect.signature(Generator.forward),
            'g_module': inspect.signature(Generator.forward), 'discriminator': inspect.signature(Discriminator.forward),
            'Generator': inspect.signature(Generator.__init__), 'Discriminator': inspect.signature(Discriminator.__init__)}
    n = 0
    for s in ('train_spatial_query.py', 'test_spatial_query.py'):
        for name, npos, kws, line in _calls(trees[s], set(sigs)):
            try:
                sigs[name].bind(None, *([0] * npos), **{k: 0 for k in kws})
            except TypeError as e:
                raise AssertionError(f'{s}:{line}: {name}(...) does not bind to our signature: {e}')
            n += 1
    assert n >= 30, n                                  # (both scripts together call the models ~40 times)
    # keyword names the reference's own Generator.forward accepts = ours (same flags, same defaults)
    ref_fwd = next(f for c in ast.walk(trees['model_spatial_query.py']) if isinstance(c, ast.ClassDef) and c.name == 'Generator'
                   for f in c.body if isinstance(f, ast.FunctionDef) and f.name == 'forward')
    ref_args = [a.arg for a in ref_fwd.args.args]
    ref_defaults = [ast.literal_eval(d) for d in ref_fwd.args.defaults]
    ours = inspect.signature(Generator.forward)
    our_args = list(ours.parameters)
    assert our_args[:len(ref_args)] == ref_args, (our_args, ref_args)
    our_defaults = [p.default for p in ours.parameters.values() if p.default is not inspect.Parameter.empty]
    assert our_defaults[:len(ref_defaults)] == ref_defaults


def test_legacy_ddp_outputs_restores_differentiation_between_wrapper_outputs():
    """torch >= 1.9: DistributedDataParallel(find_unused_parameters=True) returns its outputs through an identity autograd node, so
    the reference's path-length step - autograd.grad of one output of the wrapped generator w.r.t. another (train_spatial_query.py:
    226-232, :92-105) - raises; `utils.distributed.legacy_ddp_outputs()` (called by the drop-in `utils.distributed` on import)
    restores the behaviour of the torch the reference pins.  gloo, one rank, CPU, a two-output toy module."""
    import socket

    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from transeditor_amd.utils.distributed import legacy_ddp_outputs

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.unused = torch.nn.Linear(4, 4), torch.nn.Linear(4, 3), torch.nn.Linear(2, 2)

        def forward(self, x):
            latent = self.a(x)
            return torch.tanh(self.b(latent)), latent

    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        net = Toy()
        wrapped = DistributedDataParallel(net, find_unused_parameters=True, broadcast_buffers=False)
        x = torch.randn(5, 4)

        def path_step():
            img, latent = wrapped(x)
            g, = torch.autograd.grad(img.sum(), latent, create_graph=True)
            wrapped.zero_grad()
            g.pow(2).sum().backward()
            return [None if p.grad is None else p.grad.clone() for p in net.parameters()]

        assert legacy_ddp_outputs(False)                 # torch's own behaviour
        with pytest.raises(RuntimeError, match='not have been used in the graph'):
            path_step()
        assert legacy_ddp_outputs(True)
        got = path_step()
        img, latent = net(x)                             # the unwrapped module: same gradients, unused parameters without one
        g, = torch.autograd.grad(img.sum(), latent, create_graph=True)
        net.zero_grad()
        g.pow(2).sum().backward()
        for a, p in zip(got, net.parameters()):
            assert (a is None) == (p.grad is None)
            if a is not None:
                assert torch.allclose(a, p.grad, rtol=0, atol=1e-6)
        assert got[-1] is None and got[-2] is None       # `unused`

        # the switch is SCOPED: a static-graph wrapper, or one without find_unused_parameters, in the same process keeps
        # torch's own sink - its first backward enqueues the delayed all-reduce from _DDPSink.backward
        import torch.nn.parallel.distributed as ddp_mod
        from transeditor_amd.utils.distributed import legacy_ddp_outputs_scope
        seen = []
        orig_apply = ddp_mod._te_original_sink.apply

        net2 = torch.nn.Linear(4, 2)
        static = DistributedDataParallel(net2, static_graph=True, broadcast_buffers=False)
        ddp_mod._te_original_sink.apply = staticmethod(lambda *a: (seen.append(1), orig_apply(*a))[1])
        try:
            static(x).sum().backward()                    # first iteration of a static graph goes through the sink
            assert seen, 'static-graph wrapper bypassed the original _DDPSink'
            assert static._static_graph_delay_allreduce_enqueued
            n = len(seen)
            path_step()                                   # the reference-style wrapper: pass-through, original not called
            assert len(seen) == n
            wrapped._te_keep_ddp_sink = True              # per-instance opt-out
            with pytest.raises(RuntimeError, match='not have been used in the graph'):
                path_step()
            wrapped._te_keep_ddp_sink = False
        finally:
            ddp_mod._te_original_sink.apply = orig_apply
        legacy_ddp_outputs(False)
        with legacy_ddp_outputs_scope() as scope:         # context-manager form: on inside, torch's behaviour back outside
            assert scope.active
            path_step()
        with pytest.raises(RuntimeError, match='not have been used in the graph'):
            path_step()
    finally:
        legacy_ddp_outputs(False)
        dist.destroy_process_group()
