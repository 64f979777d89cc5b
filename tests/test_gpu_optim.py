"""T2: multi-tensor Adam / EMA kernels against torch.optim.Adam and the reference's accumulate() formula
(train_spatial_query.py:56-61, 458-473).  Tolerance 1e-6 relative over three steps (same operation order, fp32)."""
import copy

import pytest
import torch

from transeditor_amd import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _params(seed, shapes):
    return [torch.nn.Parameter(synth.normal(s, f'opt.p{i}', seed).to(DEV)) for i, s in enumerate(shapes)]


SHAPES = [(512, 512, 3, 3), (512,), (1, 3, 1, 1), (16, 17), (3,), (130, 7, 3, 3), (1,), (8192 + 5,), (4, 2048)]


@pytest.mark.parametrize('betas', [(0.0, 0.99 ** 0.8), (0.9, 0.999), (0.3, 0.99 ** (16 / 17))])
def test_fused_adam_matches_torch_adam(betas):
    from transeditor_amd.optim import FusedAdam
    ours, ref = _params(1, SHAPES), _params(1, SHAPES)
    o1 = FusedAdam(ours, lr=0.002 * 0.8, betas=betas)
    o2 = torch.optim.Adam(ref, lr=0.002 * 0.8, betas=betas)
    bucket = torch.zeros(sum(p.numel() for p in ours) + 3, device=DEV)      # gradients as 4-byte aligned bucket slices
    for step in range(3):
        off = 1 if step == 1 else 0                                          # step 1: misaligned views (scalar path)
        for i, (a, b) in enumerate(zip(ours, ref)):
            g = synth.normal(tuple(a.shape), f'opt.g{i}', 10 + step).to(DEV) * (10.0 ** (i % 3 - 1))
            if i == 4 and step < 2:
                a.grad = b.grad = None                                       # a parameter that skips steps (own step count)
                continue
            b.grad = g.clone()
            if step == 0:
                a.grad = g.clone()
            else:
                v = bucket[off:off + a.numel()].view_as(a)
                v.copy_(g)
                a.grad = v
                off += a.numel()
        o1.step()
        o2.step()
        for i, (a, b) in enumerate(zip(ours, ref)):
            err = float((a.detach() - b.detach()).abs().max() / b.detach().abs().max())
            assert err < 1e-6, (step, i, err)
    # state layout == torch.optim.Adam's: a checkpoint written by one loads into the other and continues identically
    sd1, sd2 = o1.state_dict(), o2.state_dict()
    assert sd1['state'].keys() == sd2['state'].keys()
    for k in sd1['state']:
        assert set(sd1['state'][k]) == {'step', 'exp_avg', 'exp_avg_sq'} == set(sd2['state'][k])
        assert float(sd1['state'][k]['step']) == float(sd2['state'][k]['step'])
        for name in ('exp_avg', 'exp_avg_sq'):
            a, b = sd1['state'][k][name], sd2['state'][k][name]
            assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()) + 1e-30, (k, name)
    o3 = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ours], lr=0.002 * 0.8, betas=betas)
    o3.load_state_dict(copy.deepcopy(sd1))
    o4 = FusedAdam([torch.nn.Parameter(p.detach().clone()) for p in ours], lr=0.002 * 0.8, betas=betas)
    o4.load_state_dict(copy.deepcopy(sd2))
    for i, (a, b) in enumerate(zip(o3.param_groups[0]['params'], o4.param_groups[0]['params'])):
        a.grad = synth.normal(tuple(a.shape), f'opt.g{i}', 99).to(DEV)
        b.grad = a.grad.clone()
    o3.step()
    o4.step()
    for a, b in zip(o3.param_groups[0]['params'], o4.param_groups[0]['params']):
        assert float((a.detach() - b.detach()).abs().max() / a.detach().abs().max()) < 1e-6


def test_fused_adam_steps_after_loading_reference_style_state():
    """a `g_optim` written by the reference (torch 1.7: python-int `step`) or loaded with
    torch.load(map_location='cuda') (tensor `step` on the GPU) must load and continue exactly like torch.optim.Adam."""
    from transeditor_amd.optim import FusedAdam
    shapes = [(64, 32, 3, 3), (64,), (7, 5)]
    base, ref = _params(3, shapes), _params(3, shapes)
    o_ref = torch.optim.Adam(ref, lr=0.0016, betas=(0.0, 0.99 ** 0.8))
    for i, b in enumerate(ref):
        b.grad = synth.normal(tuple(b.shape), f'optl.g{i}', 1).to(DEV)
    o_ref.step()
    for layout in ('int', 'cuda'):
        sd = copy.deepcopy(o_ref.state_dict())
        for st in sd['state'].values():
            st['step'] = int(st['step']) if layout == 'int' else torch.tensor(float(st['step']), device=DEV)
        ours = [torch.nn.Parameter(b.detach().clone()) for b in ref]
        o = FusedAdam(ours, lr=0.0016, betas=(0.0, 0.99 ** 0.8))
        o.load_state_dict(sd)
        twin = [torch.nn.Parameter(b.detach().clone()) for b in ref]
        o_t = torch.optim.Adam(twin, lr=0.0016, betas=(0.0, 0.99 ** 0.8))
        o_t.load_state_dict(copy.deepcopy(o_ref.state_dict()))
        for i, (a, b) in enumerate(zip(ours, twin)):
            a.grad = synth.normal(tuple(a.shape), f'optl.g{i}', 2).to(DEV)
            b.grad = a.grad.clone()
        o.step()
        o_t.step()
        for a, b in zip(ours, twin):
            assert float((a.detach() - b.detach()).abs().max() / b.detach().abs().max()) < 1e-6, layout
        assert all(st['step'].device.type == 'cpu' and float(st['step']) == 2.0 for st in o.state.values())


def test_multi_tensor_ema_matches_reference_formula():
    from transeditor_amd.optim import MultiTensorEMA
    from transeditor_amd.train_step import accumulate

    class M(torch.nn.Module):
        def __init__(self, seed):
            super().__init__()
            self.ps = torch.nn.ParameterList(_params(seed, SHAPES))
    a, b = M(3), M(4)
    ref = [p.detach().clone() for p in a.parameters()]
    decay = 0.5 ** (32 / (10 * 1000))
    ema = MultiTensorEMA(a, b)
    for _ in range(3):
        ema.update(decay)
        for r, q in zip(ref, b.parameters()):
            r.mul_(decay).add_(q.detach(), alpha=1 - decay)                  # par1.mul_(decay).add_(1 - decay, par2)
    for p, r in zip(a.parameters(), ref):
        assert float((p.detach() - r).abs().max() / r.abs().max()) < 1e-6
    accumulate(a, b, 0)                                                      # decay 0: plain copy (train_spatial_query.py:455)
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p.detach(), q.detach())


def test_fused_adam_rejects_cpu_parameters():
    from transeditor_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match='GPU'):
        FusedAdam([p], lr=0.1).step()


@pytest.mark.gpu
def test_refresh_packed_weights_equals_single_packs():
    """te_conv_pack_weights_multi_f32 (every stale layout of a cache in one launch, after an optimiser step) against the
    per-layer packing launches: all three layouts, 3x3 and 1x1, ragged channel counts, the [1,Co,Ci,k,k] parameter layout of
    ModulatedConv2d; untouched parameters are left alone and a stale entry is really rewritten."""
    import torch
    from transeditor_amd import _lib
    from transeditor_amd.op.modconv import packed, packed2, packed_weights_cache, refresh_packed_weights
    torch.manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(*s, device='cuda')) for s in [(1, 24, 40, 3, 3), (64, 3, 1, 1), (512, 512, 3, 3), (130, 66, 1, 1)]]
    views = [p[0] if p.dim() == 5 else p for p in ps]
    cache = {}
    with packed_weights_cache(cache):
        for w in views:
            packed2(w, _lib.PACK_FWD, _lib.PACK_DGRAD, 0.37)
        packed(views[2], _lib.PACK_SWAP, 1.5)
        assert len(cache) == 9 and refresh_packed_weights(cache) == 0          # nothing stale yet
        held = {k: v[1] for k, v in cache.items()}
        before = {k: v.clone() for k, v in held.items()}
        with torch.no_grad():
            for p in ps[:3]:
                p.add_(torch.randn_like(p))                                    # an optimiser step (bumps the version counters)
        assert refresh_packed_weights(cache) == 7                              # ps[3]'s two layouts are not touched
        touched = {p.data_ptr() for p in ps[:3]}
        for key, (ver, wp, base) in cache.items():
            assert ver == base._version                                        # entry current again
            # a stale layout is re-derived into a NEW buffer (the old one may still be held by an autograd node, ADVICE
            # round 3) and the old buffer keeps the old weights; an untouched parameter keeps its buffer
            assert (wp is not held[key]) == (base.data_ptr() in touched), key
            assert torch.equal(held[key], before[key]), key
            want = _lib.conv_pack(base.detach().view(key[1]), key[2], key[3])
            assert torch.equal(wp, want), key
        hits = packed(views[0], _lib.PACK_FWD, 0.37)
        assert hits is cache[(views[0].data_ptr(), tuple(views[0].shape), _lib.PACK_FWD, 0.37)][1]
