"""CPU-side checks: the C-ABI library builds, loads and exports every symbol include/te_hip.h declares;
the product path refuses to run without a GPU (no silent fallback); host-side shape logic and the
state_dict schema match the reference."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope='module')
def libpath():
    from transeditor_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(libpath):
    header = open(os.path.join(ROOT, 'include', 'te_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(te_[a-z0-9_]+)\s*\(', header)))
    assert len(declared) >= 14
    lib = ctypes.CDLL(libpath)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    lib.te_arch.restype = ctypes.c_char_p
    assert lib.te_arch() == b'gfx950'
    assert lib.te_version() == 3


def test_binding_table_matches_header(libpath):
    from transeditor_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'te_hip.h')).read()
    declared = set(re.findall(r'\b(te_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_lib.EXPORTS)
    assert _lib.lib().te_version() == 3


def test_argument_validation_returns_error_codes(libpath):
    """No compute without a GPU: only the host-side validation paths are exercised."""
    from transeditor_amd import _lib
    L = _lib.lib()
    assert L.te_bias_act_f32(None, None, None, None, 3, 0, 0.2, 1.0, 16, 1, 1, None) == -1
    assert b'NULL' in L.te_last_error_string()
    assert L.te_conv_packed_numel(0, 5, 6, 3) == 9 * 16 * 128
    assert L.te_conv_packed_numel(1, 512, 256, 3) == 9 * 512 * 256
    assert L.te_wgrad_slab_count(0, 16, 128, 128, 256, 256) >= 1
    assert L.te_wgrad_slab_count(0, 0, 128, 128, 256, 256) < 0
    # host-side plans of the Winograd forms (pure functions of the shape)
    assert L.te_conv_wino_supported(16, 128, 128, 256, 256) == 1 and L.te_conv_wino_supported(16, 128, 128, 16, 16) == 0
    assert L.te_conv_wino_supported(16, 3, 128, 256, 256) == 0 and L.te_conv_wino_supported(4, 32, 32, 1024, 1024) == 1
    assert L.te_conv_packed_numel(_lib.PACK_WFWD, 128, 256, 3) == 12 * 128 * 256
    assert L.te_wgrad_pair_form(0, 128, 128, 256, 256) == 1 and L.te_wgrad_pair_form(0, 64, 64, 512, 512) == 0
    assert L.te_conv_wino6_supported(16, 128, 128, 256, 256) == 1 and L.te_conv_wino6_supported(4, 64, 64, 512, 512) == 1
    assert L.te_conv_wino6_supported(4, 32, 32, 1024, 1024) == 0 and L.te_conv_wino6_supported(16, 512, 512, 8, 8) == 0        # M % 64, W % 32 ...
    assert L.te_conv_wino6_supported(16, 512, 512, 16, 16) == 1 and L.te_conv_wino6_supported(15, 512, 512, 16, 16) == 0    # ... or W == 16, even batch (round 6)
    assert L.te_conv_wino6_supported(16, 48, 64, 32, 32) == 0                                                                    # K % 32
    assert L.te_conv_packed_numel(_lib.PACK_W6FWD, 128, 256, 3) == 18 * 128 * 256 == L.te_conv_packed_numel(_lib.PACK_W6DGRAD, 128, 256, 3)
    assert L.te_wgrad_pair_form(1, 128, 128, 64, 64) == 0 and L.te_wgrad_pair_form(0, 128, 128, 1, 1) == 0


def test_product_path_fails_loudly_on_cpu(libpath):
    from transeditor_amd.op import fused_leaky_relu, upfirdn2d
    from transeditor_amd.op.modconv import modconv
    with pytest.raises(RuntimeError, match='GPU'):
        fused_leaky_relu(torch.randn(2, 4), torch.zeros(4))
    with pytest.raises(RuntimeError, match='GPU'):
        upfirdn2d(torch.randn(1, 1, 8, 8), torch.ones(4, 4) / 16, pad=(1, 1))
    with pytest.raises(RuntimeError, match='GPU'):
        modconv(torch.randn(1, 8, 4, 4), torch.randn(8, 8, 3, 3))
    # the dense / normalisation ops of the mapping and attention stack have no CPU route either
    from transeditor_amd.op.layernorm import pixel_norm, sample_layer_norm
    from transeditor_amd.op.linear import linear_fused
    from transeditor_amd.op.style import demod
    from transeditor_amd.op.token_mlp import token_mlp
    with pytest.raises(RuntimeError, match='GPU'):
        linear_fused(torch.randn(4, 8), torch.randn(6, 8), torch.zeros(6))
    with pytest.raises(RuntimeError, match='GPU'):
        sample_layer_norm(torch.randn(2, 4, 8))
    with pytest.raises(RuntimeError, match='GPU'):
        pixel_norm(torch.randn(2, 8, 4), 1)
    with pytest.raises(RuntimeError, match='GPU'):
        token_mlp(torch.randn(2, 8, 4), [torch.randn(8, 8)] * 4, [torch.zeros(8)] * 4, 1.0, 1.0)
    with pytest.raises(RuntimeError, match='GPU'):
        demod(torch.randn(8, 8, 3, 3), torch.randn(2, 8), 1.0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'transeditor_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), os.path.join(dirpath, f)
                assert '/root/reference' not in src, os.path.join(dirpath, f)


@pytest.mark.parametrize('tag,ctor', [('g64', ('G', 64, 10)), ('g256', ('G', 256, 14)), ('g1024', ('G', 1024, 18)),
                                      ('d64', ('D', 64, 0)), ('d256', ('D', 256, 0))])
def test_state_dict_schema_matches_reference(golden, tag, ctor):
    from transeditor_amd.model_spatial_query import Discriminator, Generator
    g = golden('state_dict_schema')
    kind, size, token = ctor
    m = Generator(size, 512, 512, token, n_trans=8, pixel_norm_op_dim=1) if kind == 'G' else Discriminator(size)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g[tag + '.keys']]
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g[tag + '.shapes']]
    assert sum(p.numel() for p in m.parameters()) == int(g[tag + '.nparams'])
    assert [n for n, _ in m.named_parameters()] == [str(k) for k in g[tag + '.param_names']]


def test_upfirdn2d_geometry_formulas():
    """out size and adjoint padding (utils/op/upfirdn2d.py:101-112 in the reference): integer, exact."""
    from transeditor_amd.op.upfirdn2d import _geometry
    # blur pad (1,1) on 2H+1 -> 2H ; adjoint pads (2,2)
    assert _geometry((9, 9), (4, 4), (1, 1), (1, 1), (1, 1, 1, 1)) == ((8, 8), (2, 2, 2, 2))
    # D blur pad (2,2) -> adjoint (1,1)
    assert _geometry((8, 8), (4, 4), (1, 1), (1, 1), (2, 2, 2, 2)) == ((9, 9), (1, 1, 1, 1))
    # skip upsample x2 pad (2,1) -> adjoint (1,1)
    assert _geometry((5, 7), (4, 4), (2, 2), (1, 1), (2, 1, 2, 1)) == ((10, 14), (1, 1, 1, 1))


def test_synth_is_deterministic_and_normal():
    from transeditor_amd import synth
    a = synth.normal((4096,), 'unit.test', 3)
    b = synth.normal((4096,), 'unit.test', 3)
    assert torch.equal(a, b)
    assert abs(float(a.mean())) < 0.06 and abs(float(a.std()) - 1) < 0.05
    assert not torch.equal(a, synth.normal((4096,), 'unit.test', 4))
    # pinned values: the GPU box must regenerate exactly these
    assert np.allclose(a[:3].numpy(), synth.normal((3,), 'unit.test', 3).numpy())


def test_generator_ctor_surface():
    import inspect
    from transeditor_amd.model_spatial_query import Generator, ModulatedConv2d
    sig = inspect.signature(Generator.__init__)
    assert list(sig.parameters)[1:] == ['size', 'style_dim', 'param_dim', 'token_dim', 'channel_multiplier', 'blur_kernel',
                                        'lr_mlp', 'layer_noise_injection', 'use_spatial_mapping', 'num_region', 'n_trans',
                                        'pixel_norm_op_dim', 'no_trans']
    assert sig.parameters['n_trans'].default == 4 and sig.parameters['pixel_norm_op_dim'].default == 2
    fsig = inspect.signature(Generator.forward)
    assert list(fsig.parameters)[1:] == ['style', 'op_param', 'return_latents', 'input_is_latent', 'noise',
                                         'randomize_noise', 'return_style', 'return_p_latent', 'return_only_style',
                                         'return_only_style_latent', 'return_only_mapped_p', 'return_only_mapped_z',
                                         'use_spatial_mapping', 'use_style_mapping', 'trans_interact',
                                         'return_mapped_codes']
    g = Generator(32, 512, 512, 8, n_trans=2)
    assert g.n_latent == 8 and g.num_layers == 7 and len(g.convs) == 6 and len(g.to_rgbs) == 3
    assert list(inspect.signature(ModulatedConv2d.__init__).parameters)[1:9] == [
        'in_channel', 'out_channel', 'kernel_size', 'style_dim', 'demodulate', 'upsample', 'downsample', 'blur_kernel']


def test_train_step_losses_match_oracle_on_cpu():
    """Loss formulas of the train-step harness (pure torch) against the oracle's restatement."""
    from oracle import te_oracle as O
    from transeditor_amd import synth, train_step as T
    rp, fp = synth.normal((6, 1), 'l.r'), synth.normal((6, 1), 'l.f')
    assert torch.allclose(T.d_logistic_loss(rp, fp), O.d_logistic_loss(rp, fp))
    assert torch.allclose(T.g_nonsaturating_loss(fp), O.g_nonsaturating_loss(fp))
    lat = synth.normal((2, 4, 8), 'l.lat').requires_grad_(True)
    A = synth.normal((8, 3 * 4 * 4), 'l.A')
    img = torch.tanh(lat.sum(1) @ A).reshape(2, 3, 4, 4)
    noise = synth.normal((2, 3, 4, 4), 'l.n')
    a = T.g_path_regularize(img, lat, 0.0, noise)
    b = O.g_path_regularize(img, lat, 0.0, noise / 4.0)
    assert torch.allclose(a[0], b[0]) and torch.allclose(a[2], b[2])
    m1, m2 = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)
    with pytest.raises(RuntimeError, match='no CPU path'):        # the EMA is a HIP kernel (tests/test_gpu_optim.py)
        T.accumulate(m1, m2, 0.9)
    args = T.default_args(size=64)
    assert args.token == 10 and args.d_reg_every == 16 and args.g_reg_every == 4


def test_modconv_host_argument_logic():
    """activation-gain codes and the demodulation argument contract are decided on the host, before any launch"""
    from transeditor_amd.op import modconv as M
    assert M._act_code(False) == 0 and M._act_code(True) == 3 and M._act_code(2 ** 0.5) == 3 and M._act_code(1.0) == 4
    assert M._act_code(2 ** 0.5 * (1 / 2 ** 0.5)) == 4                       # ResBlock: sqrt(2) * 1/sqrt(2)
    with pytest.raises(RuntimeError, match='gain'):
        M._act_code(0.5)
    x, w = torch.randn(1, 8, 4, 4), torch.randn(8, 8, 3, 3)
    with pytest.raises(RuntimeError, match='demod_eps'):
        M.modconv(x, w, isc=torch.ones(1, 8), osc=torch.ones(1, 8), demod_eps=1e-8)
    with pytest.raises(RuntimeError, match='demod_eps'):
        M.modconv(x, w, demod_eps=1e-8)                                       # demodulation without a style scale
    assert M._bwd_pack_kind('up') == M._bwd_pack_kind('down') != M._bwd_pack_kind('3x3')
    with M.second_order(), M.no_weight_grads():
        assert M._STATE == {'second_order': True, 'skip_w': True}
    assert M._STATE == {'second_order': False, 'skip_w': False}
    # forward-side hints and the frozen-weight cache are thread-local (nn.DataParallel worker threads, a sampler next to a
    # training loop); the skip flag travels on a token the nodes keep, because backward runs on the autograd engine's thread
    import threading
    seen = {}

    def other():
        seen['so'] = M._STATE['second_order']
        seen['frozen'] = M._STATE.frozen_on
        seen['graph_is_default'] = M.current_graph() is M._DEFAULT_GRAPH
    with M.second_order() as tok, M.frozen_weights({}):
        assert M.current_graph() is tok and tok is not M._DEFAULT_GRAPH
        t = threading.Thread(target=other)
        t.start()
        t.join()
    assert seen == {'so': False, 'frozen': False, 'graph_is_default': True}
    with M.no_weight_grads():                       # flips the token of the forward that just ended (and the default one)
        assert tok.skip_w and M._DEFAULT_GRAPH.skip_w
    assert not tok.skip_w and not M._DEFAULT_GRAPH.skip_w


def test_upfirdn2d_geometry_matches_reference_formulas():
    """output size and adjoint pads (utils/op/upfirdn2d.py:101-112) for the five configurations the model uses"""
    from transeditor_amd.op.upfirdn2d import _geometry
    # blur pad (1,1) on 257 -> 256, its adjoint pads (1,3): 256 -> 257
    (oh, ow), g_pad = _geometry((257, 257), (4, 4), (1, 1), (1, 1), (1, 1, 1, 1))
    assert (oh, ow) == (256, 256) and g_pad == (2, 2, 2, 2)
    (oh, ow), _ = _geometry((256, 256), (4, 4), (1, 1), (1, 1), g_pad)
    assert (oh, ow) == (257, 257)
    # ToRGB skip upsample: up 2, pad (2,1): 128 -> 256; adjoint is down 2
    (oh, ow), g_pad = _geometry((128, 128), (4, 4), (2, 2), (1, 1), (2, 1, 2, 1))
    assert (oh, ow) == (256, 256) and g_pad == (1, 1, 1, 1)
    # discriminator blur pad (2,2) before the stride-2 3x3: 256 -> 257; skip branch blur (1,1) with down 2: 256 -> 128
    assert _geometry((256, 256), (4, 4), (1, 1), (1, 1), (2, 2, 2, 2))[0] == (257, 257)
    assert _geometry((256, 256), (4, 4), (1, 1), (2, 2), (1, 1, 1, 1))[0] == (128, 128)


def test_tools_and_entry_points_compile():
    """the GPU-side diagnostics under tools/ (run only on the GPU box) at least parse, so a visit is never lost to a typo"""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, 'tools', '*.py')) + [os.path.join(root, 'bench.py'), os.path.join(root, '__graft_entry__.py')]
    assert len(files) > 10
    for f in files:
        compile(open(f).read(), f, 'exec')


def test_roctx_ranges_behind_env_switch():
    """TE_ROCTX=1 wraps every tensor-level entry of _lib in a roctx range (operator names in rocprofv3 marker traces)"""
    import subprocess
    import sys
    code = ("from transeditor_amd import _lib; assert _lib.ROCTX and _lib.conv.__wrapped__.__name__ == 'conv' "
            "and _lib.upfirdn2d_raw.__wrapped__.__name__ == 'upfirdn2d_raw'; print('ok')")
    env = dict(os.environ, TE_ROCTX='1', PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr
    from transeditor_amd import _lib
    assert not _lib.ROCTX and not hasattr(_lib.conv, '__wrapped__')


def test_fused_adam_loads_reference_style_optimizer_state():
    """the reference's torch 1.7 Adam stores `step` as a python int, and torch.load(map_location=device) moves a
    tensor `step` to the GPU; FusedAdam normalises both to a host fp32 scalar at load time (the kernel step itself is GPU-only:
    tests/test_gpu_optim.py)."""
    from transeditor_amd.optim import FusedAdam
    p = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    opt = FusedAdam(p, lr=0.002, betas=(0.0, 0.99))
    ref = torch.optim.Adam(p, lr=0.002, betas=(0.0, 0.99))
    for q in p:
        q.grad = torch.randn_like(q)
    ref.step()
    sd = ref.state_dict()
    for st in sd['state'].values():
        st['step'] = 7                                   # torch 1.7 layout: python int
    opt.load_state_dict(sd)
    for q in p:
        s = opt.state[q]['step']
        assert torch.is_tensor(s) and s.device.type == 'cpu' and s.dtype == torch.float32 and float(s) == 7.0
        assert torch.equal(opt.state[q]['exp_avg_sq'], ref.state[q]['exp_avg_sq'])
    opt.state[p[0]]['step'] = 9                          # assigned behind the optimiser's back: normalised at the next step
    FusedAdam._host_step(opt.state[p[0]])
    assert float(opt.state[p[0]]['step']) == 9.0 and torch.is_tensor(opt.state[p[0]]['step'])


def test_wgrad_group_plan_invariants():
    """te_wgrad_group_plan is host code (no GPU): NB divides B, is 1 wherever grouping must not apply (channel tails - the 16-byte
    staging path masks a tail through the end of the sample's buffer range -, a sample that already splits into chunks, too few
    blocks left), keeps every CU a block when it groups, and a group stays below the 2 GiB the zero-fill offset needs."""
    import ctypes as C
    from transeditor_amd import _lib
    L = _lib.lib()

    def plan(kind, B, Co, Ci, H, W):
        nb, s = C.c_int(0), C.c_int(0)
        assert L.te_wgrad_group_plan(kind, B, Co, Ci, H, W, C.byref(nb), C.byref(s)) == 0
        return nb.value, s.value
    for kind in (_lib.CONV_3X3, _lib.CONV_T2, _lib.CONV_1X1):
        for B in (1, 2, 8, 12, 16, 32, 64):
            for Co, Ci in ((512, 512), (512, 513), (130, 512), (128, 128), (256, 512)):
                for H in (1, 4, 8, 33, 64, 256):
                    nb, s = plan(kind, B, Co, Ci, H, H)
                    assert nb >= 1 and B % nb == 0 and s >= 1, (kind, B, Co, Ci, H, nb, s)
                    if Co % 128 or Ci % 128 or L.te_wgrad_slab_count(kind, B, Co, Ci, H, H) > 1:
                        assert nb == 1, (kind, B, Co, Ci, H, nb)
                    if nb > 1:
                        tiles = -(-Co * Ci // (128 * 64))
                        assert (B // nb) * tiles >= 256
                        biggest = max(Co * ((2 * H + 1) ** 2 if kind == _lib.CONV_T2 else H * H), Ci * H * H) * 4
                        assert nb * biggest < 2 ** 31
    assert plan(_lib.CONV_3X3, 32, 512, 512, 4, 4) == (4, 1) and plan(_lib.CONV_3X3, 16, 512, 512, 16, 16)[0] == 2
    nb, s = C.c_int(0), C.c_int(0)
    assert L.te_wgrad_group_plan(_lib.CONV_3X3, 0, 512, 512, 4, 4, C.byref(nb), C.byref(s)) != 0      # bad dims are refused


def test_minibatch_stddev_chunks_expression():
    """the recorded-backward expression of the minibatch stddev with `chunks` (Discriminator.forward(x, chunks=2): the D step's
    fake and real passes as one batch) is the per-pass statistic of model_spatial_query.py:844-852 laid end to end - and differs
    from the statistic of the joined batch"""
    import torch
    from transeditor_amd.op.stddev import _torch_expr
    from plain_torch import minibatch_stddev as plain
    torch.manual_seed(0)
    a, b = torch.randn(8, 6, 4, 4, dtype=torch.float64), torch.randn(8, 6, 4, 4, dtype=torch.float64)
    joint = _torch_expr(torch.cat([a, b]), 4, 1, chunks=2)
    assert torch.allclose(joint, torch.cat([plain(a, 4), plain(b, 4)]), rtol=1e-12, atol=1e-14)
    assert not torch.allclose(joint, plain(torch.cat([a, b]), 4))
    assert torch.equal(_torch_expr(a, 4, 1, chunks=1), plain(a, 4))


def test_packed_weight_cache_refresh_logic(monkeypatch):
    """host logic of the packed-weight cache (op/modconv.py) with the packing launches replaced by fakes: hits while the version
    counter stands, ONE multi-tensor refresh of exactly the stale entries after an optimiser step (into fresh buffers), entries of partial
    views dropped, temporaries never cached"""
    import torch
    from transeditor_amd import _lib
    from transeditor_amd.op import modconv as mc
    calls = {'single': 0, 'multi': []}

    def conv_pack(w, kind, wscale=1.0):
        calls['single'] += 1
        return (w.detach().flatten() * wscale + kind).clone()

    def conv_pack2(w, ka, kb, wscale=1.0):
        return conv_pack(w, ka, wscale), conv_pack(w, kb, wscale)

    def conv_pack_multi(jobs):
        calls['multi'].append(len(jobs))
        for wp, w, kind, wscale in jobs:
            wp.copy_(w.flatten() * wscale + kind)
    monkeypatch.setattr(_lib, 'conv_pack', conv_pack)
    monkeypatch.setattr(_lib, 'conv_pack2', conv_pack2)
    monkeypatch.setattr(_lib, 'conv_pack_multi', conv_pack_multi)
    p5 = torch.nn.Parameter(torch.randn(1, 4, 3, 3, 3))          # ModulatedConv2d layout: the op sees weight[0]
    p4 = torch.nn.Parameter(torch.randn(4, 3, 1, 1))
    p2 = torch.nn.Parameter(torch.randn(2, 4, 3, 3, 3))          # a PARTIAL view of it is cached but cannot be refreshed in bulk
    cache = {}
    with mc.packed_weights_cache(cache):
        a = mc.packed(p5[0], 0, 0.5)
        mc.packed2(p4, 0, 1, 2.0)
        mc.packed(p2[1], 0, 1.0)
        assert mc.packed(p5[0], 0, 0.5) is a and calls['single'] == 4 and len(cache) == 4
        mc.packed(torch.randn(4, 3, 3, 3), 0, 1.0)                # a temporary: packed, not cached
        assert len(cache) == 4 and calls['single'] == 5
        assert mc.refresh_packed_weights(cache) == 0 and calls['multi'] == [0]
        with torch.no_grad():
            for p in (p5, p4, p2):
                p.mul_(2.0)                                       # "optimiser step": version counters move
        assert mc.refresh_packed_weights(cache) == 3 and calls['multi'][-1] == 3      # p5 (1 layout) + p4 (2 layouts); the partial view is dropped
        assert len(cache) == 3
        a_old = a.clone()
        a2 = mc.packed(p5[0], 0, 0.5)                             # a hit again, on a NEW buffer: the old layout (possibly still
        assert a2 is not a and torch.equal(a, a_old)              # held by an autograd node) is left untouched
        assert torch.equal(a2, p5.detach()[0].flatten() * 0.5) and mc.packed(p5[0], 0, 0.5) is a2
        assert calls['single'] == 5
        mc.packed(p2[1], 0, 1.0)                                  # the dropped entry is repacked at its next use
        assert calls['single'] == 6 and len(cache) == 4


def test_convolution_form_selection_is_a_pure_function_of_the_shape(libpath):
    """op/modconv.fwd_kinds / bwd_kinds (no GPU): split-bf16 Winograd where te_conv_wino6_supported says so, fp32 Winograd where only
    te_conv_wino_supported does, the direct kernels elsewhere; TE_SPLIT_BF16 / USE_WINOGRAD switch the forms off in that order"""
    from transeditor_amd import _lib
    from transeditor_amd.op import modconv
    w = lambda co, ci: torch.empty(co, ci, 3, 3)
    assert modconv.fwd_kinds('3x3', 16, w(128, 128), 256, 256) == (_lib.PACK_W6FWD, _lib.CONV_3X3W6)
    assert modconv.bwd_kinds('3x3', 16, w(256, 512), 64, 64) == (_lib.PACK_W6DGRAD, _lib.CONV_3X3W6)
    assert modconv.fwd_kinds('3x3', 4, w(32, 32), 1024, 1024) == (_lib.PACK_WFWD, _lib.CONV_3X3W)
    assert modconv.fwd_kinds('3x3', 16, w(512, 512), 16, 16) == (_lib.PACK_W6FWD, _lib.CONV_3X3W6)        # round 6: two samples side by side
    assert modconv.fwd_kinds('3x3', 15, w(512, 512), 16, 16) == (_lib.PACK_FWD, _lib.CONV_3X3)            # (odd batch), small batches, smaller images: direct kernel
    assert modconv.fwd_kinds('3x3', 8, w(512, 512), 16, 16) == (_lib.PACK_FWD, _lib.CONV_3X3)
    assert modconv.fwd_kinds('3x3', 16, w(512, 512), 8, 8) == (_lib.PACK_FWD, _lib.CONV_3X3)
    # (round 5) the strided kind on the bf16 pipe where te_conv_s2s6_supported says so: the up-sampling layers' data gradient and the
    # discriminator's down-sampling convolutions from 16 x 16 outputs up; smaller images stay on the fp32 kernel
    assert modconv.fwd_kinds('up', 16, w(256, 512), 64, 64) == (_lib.PACK_T6FWD, _lib.CONV_T2S6)
    assert modconv.fwd_kinds('up', 16, w(512, 512), 8, 8) == (_lib.PACK_FWD, _lib.CONV_T2)
    assert modconv.bwd_kinds('down', 32, w(256, 128), 128, 128) == (_lib.PACK_T6SWAP, _lib.CONV_T2S6)
    assert modconv.bwd_kinds('up', 16, w(256, 512), 64, 64) == (_lib.PACK_S6SWAP, _lib.CONV_S2S6)
    assert modconv.fwd_kinds('down', 32, w(256, 128), 128, 128) == (_lib.PACK_S6FWD, _lib.CONV_S2S6)
    assert modconv.fwd_kinds('down', 32, w(512, 512), 8, 8) == (_lib.PACK_FWD, _lib.CONV_S2)
    assert modconv.bwd_kinds('up', 16, w(512, 512), 4, 4) == (_lib.PACK_SWAP, _lib.CONV_S2)
    old = modconv.USE_SPLIT_BF16, modconv.USE_WINOGRAD
    try:
        modconv.USE_SPLIT_BF16 = False
        assert modconv.fwd_kinds('3x3', 16, w(128, 128), 256, 256) == (_lib.PACK_WFWD, _lib.CONV_3X3W)
        assert _lib.wgrad_split() == 0               # ONE switch: the weight-gradient kernels follow the Python flag
        modconv.USE_WINOGRAD = False
        assert modconv.fwd_kinds('3x3', 16, w(128, 128), 256, 256) == (_lib.PACK_FWD, _lib.CONV_3X3)
    finally:
        modconv.USE_SPLIT_BF16, modconv.USE_WINOGRAD = old
    assert _lib.wgrad_split() == (1 if old[0] else 0)


def test_load_checkpoint_into_is_all_or_nothing():
    """train_step.load_checkpoint_into validates its arguments BEFORE loading anything (the reference's restore,
    train_spatial_query.py:475-492, is all-or-nothing): a partial set of training objects or a file with some but not all training
    entries raises and leaves every module untouched; a 'g_ema'-only file loads the EMA generator alone; a full checkpoint into g_ema only warns."""
    import warnings

    import torch
    from transeditor_amd.train_step import load_checkpoint_into
    mk = lambda: torch.nn.Linear(3, 2)
    src = {k: mk() for k in ('g_ema', 'g', 'd')}
    opt = {k: torch.optim.Adam(src[k].parameters()) for k in ('g', 'd')}
    full = {'g_ema': src['g_ema'].state_dict(), 'g': src['g'].state_dict(), 'd': src['d'].state_dict(),
            'g_optim': opt['g'].state_dict(), 'd_optim': opt['d'].state_dict()}
    ema, g, d = mk(), mk(), mk()
    before = ema.weight.clone()
    go, do = torch.optim.Adam(g.parameters()), torch.optim.Adam(d.parameters())
    with pytest.raises(ValueError):
        load_checkpoint_into(full, ema, generator=g)                    # partial set of training objects
    with pytest.raises(KeyError):
        load_checkpoint_into({k: full[k] for k in ('g_ema', 'g', 'd')}, ema, g, d, go, do)   # neither a training nor a g_ema-only file
    assert torch.equal(ema.weight, before)                                  # nothing was loaded on the failed attempts
    gw = g.weight.clone()
    with pytest.raises(KeyError):                    # a g_ema-only file into a TRAINING restore: the reference's ckpt['g'] raises too (:487)
        load_checkpoint_into({'g_ema': src['d'].state_dict()}, ema, g, d, go, do)
    assert torch.equal(ema.weight, before)
    assert load_checkpoint_into({'g_ema': src['d'].state_dict()}, ema, g, d, go, do, g_ema_only_ok=True) is None   # knowingly: EMA alone
    assert torch.equal(ema.weight, src['d'].weight) and torch.equal(g.weight, gw)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        load_checkpoint_into(full, ema)
    assert any('g_ema only' in str(r.message) for r in rec)
    assert torch.equal(ema.weight, src['g_ema'].weight)
    load_checkpoint_into(full, ema, g, d, go, do)
    assert torch.equal(g.weight, src['g'].weight) and torch.equal(d.weight, src['d'].weight)
