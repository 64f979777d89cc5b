"""SECOND-ORDER parity at the shapes bench.py times.

The path-length regulariser (train_spatial_query.py:92-105, 226-250: batch 8 at 256 px, double backward through the
T2 / S2 / 3x3 trio and the chan_scale / chan_dot pair) and R1 (:77-83, 196-206: batch 16 at 256 px) are 62 ms and 84 ms
sub-steps of the headline number; rounds 1-2 pinned their second order only at 32 px.  Here, at 256 px:

* path-length step, batch 2, against the CPU oracle: path lengths, penalty, every parameter-gradient norm (5e-3, the bar
  of the 32-px test: a second-order quantity of a squared deviation);
* the batch-8 path step through linearity over the batch: with a FIXED target length the penalty is a sum over samples, so
  every parameter gradient of the batch-8 double backward (multi-sample tiles, slab splits chosen under `second_order()`
  at B = 8) equals the sum of four batch-2 double backwards;
* R1 at batch 16 against the reference's own 256-px discriminator fixture (tests/golden/discriminator256_b4.npz): the
  minibatch-stddev groups of a batch of 16 are {j, j+4, j+8, j+12} (model_spatial_query.py:844-852), so a batch whose
  sample j+4k is the fixture's sample k consists of four copies of the fixture's group: predictions, R1 and the R1
  parameter-gradient norms must equal the batch-4 fixture's.
"""
import math

import pytest
import torch

from conftest import rel_err, rel_l2
from oracle import te_oracle as O
from test_oracle_golden import generator_state
from transeditor_amd import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = 1e-3
PIN_SO_TOL = 5e-4      # second-order gradient parity with the leaky-ReLU slopes pinned (tests/pinning.py): element-wise L2


@pytest.fixture(autouse=True)
def _cpu_threads():
    old = torch.get_num_threads()
    torch.set_num_threads(min(old, 48))
    yield
    torch.set_num_threads(old)


def _build(size, seed):
    g, sd = generator_state(size, seed)
    g.load_state_dict(sd)
    return g.to(DEV), sd


def _trainable(k, v):
    return v.is_floating_point() and 'noises' not in k and 'kernel' not in k and not k.startswith('token')


def test_path_length_step_256_batch2_vs_oracle():
    """T1 path regulariser at the FFHQ-256 architecture through the product's own second_order() / no_weight_grads() route
    (exactly what TrainStep.path_step runs) against the CPU oracle's plain double backward."""
    from transeditor_amd.op.modconv import second_order
    from transeditor_amd.train_step import g_path_regularize
    size, B = 256, 2
    G, sd = _build(size, 7)
    z, p = synth.latents(B, 6161)
    noise = synth.normal((B, 3, size, size), 'so.pl256')
    # --- CPU oracle (the reference's arithmetic): forward, path penalty, gradients of the weighted penalty
    P, names = {}, []
    for k, v in sd.items():
        P[k] = v.clone().requires_grad_(True) if _trainable(k, v) else v
        if _trainable(k, v):
            names.append(k)
    from pinning import pinned, record_oracle
    with record_oracle() as bank:
        img_r, lat_r, _ = O.generator_forward(P, z, p, size)
    bank.extend_stacked(16, dim=1)
    pen_r, mean_r, len_r = O.g_path_regularize(img_r, lat_r, 0.0, noise / math.sqrt(size * size))
    gr = torch.autograd.grad(2.0 * 4 * pen_r + 0 * img_r[0, 0, 0, 0], [P[k] for k in names], allow_unused=True)
    ref = dict(zip(names, gr))
    # --- HIP path
    with second_order():
        img, lat, _ = G(z.to(DEV), p.to(DEV), return_latents=True)
    pen, mean, lengths = g_path_regularize(img, lat, 0.0, noise.to(DEV))
    assert rel_err(img, img_r) < TOL
    assert rel_err(lengths, len_r) < TOL, (lengths, len_r)
    assert abs(float(pen) - float(pen_r)) <= 5 * TOL * abs(float(pen_r))
    assert abs(float(mean) - float(mean_r)) <= TOL * abs(float(mean_r))
    gs = torch.autograd.grad(2.0 * 4 * pen + 0 * img[0, 0, 0, 0], list(G.parameters()), allow_unused=True)
    top = max(float(v.double().norm()) for v in ref.values() if v is not None)
    bad, unused = [], []
    for (n, _), got in zip(G.named_parameters(), gs):
        want = ref[n]
        if got is None:
            unused.append(n)
            assert want is None or float(want.abs().max()) == 0.0, n
            continue
        wn = float(want.double().norm())
        if n.endswith('k_transform.bias') or wn <= 1e-7 * top:       # analytically zero / round-off only
            continue
        e = abs(float(got.double().norm()) - wn) / wn
        if e > 5 * TOL:
            bad.append((n, e, wn))
    assert not bad, bad[:8]
    # the penalty does not reach ToRGB of the last layer's bias-free parts differently from the reference: same unused set
    assert all(n.endswith('noise.weight') for n in unused), unused
    # PINNED variant (tests/pinning.py): slopes taken from the oracle's forward, so the double backward differentiates the same
    # piecewise-linear function on both sides: path lengths 1e-4, every parameter gradient element-wise (L2) at 5e-4
    with pinned(bank) as st:
        with second_order():
            img, lat, _ = G(z.to(DEV), p.to(DEV), return_latents=True)
        pen, mean, lengths = g_path_regularize(img, lat, 0.0, noise.to(DEV))
        gs = torch.autograd.grad(2.0 * 4 * pen + 0 * img[0, 0, 0, 0], list(G.parameters()), allow_unused=True)
    assert not st['unmatched'], st['unmatched']
    assert rel_err(lengths, len_r) < 1e-4, (lengths, len_r)
    errs = {}
    for (n, _), got in zip(G.named_parameters(), gs):
        want = ref[n]
        if got is None or n.endswith('k_transform.bias') or float(want.double().norm()) <= 1e-7 * top:
            continue
        errs[n] = rel_l2(got, want)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(f'path step 256 px pinned: {st["flips"]} of {st["elements"]} slopes pinned; lengths {rel_err(lengths, len_r):.2e}; '
          f'worst {[(k, float(f"{v:.2e}")) for k, v in worst]}')
    bad = [(k, v) for k, v in errs.items() if v > PIN_SO_TOL]
    assert not bad, bad[:8]


def test_path_length_step_256_batch8_is_sum_of_batch2_steps():
    """The timed path step runs at batch 8 (batch 16 // path_batch_shrink 2).  With a fixed target length c the penalty
    sum_b (len_b - c)^2 is a sum over samples, so the batch-8 double backward must equal the sum of four batch-2 ones."""
    from transeditor_amd.op.modconv import no_weight_grads, second_order
    size, B = 256, 8
    G, _ = _build(size, 7)
    z, p = synth.latents(B, 6262)
    noise = (synth.normal((B, 3, size, size), 'so.pl256b8') / math.sqrt(size * size)).to(DEV)
    params = list(G.parameters())
    names = [n for n, _ in G.named_parameters()]
    c = 0.5

    def run(sl):
        with second_order():
            img, lat, _ = G(z[sl].to(DEV), p[sl].to(DEV), return_latents=True)
        with no_weight_grads():
            grad, = torch.autograd.grad((img * noise[sl]).sum(), lat, create_graph=True)
        lengths = torch.sqrt(grad.pow(2).sum(2).mean(1))
        loss = (lengths - c).pow(2).sum()
        return lengths.detach(), torch.autograd.grad(loss, params, allow_unused=True)

    from pinning import capture, pinned
    with capture() as bank8:
        len8, g8 = run(slice(0, B))
    acc, lens = None, []
    for k in range(4):
        l2, g2 = run(slice(2 * k, 2 * k + 2))
        lens.append(l2)
        acc = [None if t is None else t.double() for t in g2] if acc is None else \
              [None if a is None else a + t.double() for a, t in zip(acc, g2)]
    assert rel_err(len8, torch.cat(lens)) < TOL
    top = max(float(b.norm()) for b in acc if b is not None)
    bad = []
    for n, a, b in zip(names, g8, acc):
        assert (a is None) == (b is None), n
        if a is None or n.endswith('k_transform.bias') or float(b.norm()) < 1e-7 * top:
            continue
        e = abs(float(a.double().norm()) - float(b.norm())) / float(b.norm())
        # second-order gradients through different tile / split choices and flipped leaky-ReLU slopes (see
        # test_gpu_timed_shapes.py): the 5e-3 bar of the second-order tests on the norm, 1e-2 element-wise (L2)
        if e > 5 * TOL or rel_l2(a, b) > 10 * TOL:
            bad.append((n, e, rel_l2(a, b)))
    assert not bad, bad[:8]
    # PINNED variant: the batch-2 steps take the slopes of the batch-8 step
    acc, lens, flips = None, [], 0
    for k in range(4):
        with pinned(bank8.batch_slice(slice(2 * k, 2 * k + 2), B)) as st:
            l2, g2 = run(slice(2 * k, 2 * k + 2))
        assert not st['unmatched'], st['unmatched']
        flips += st['flips']
        lens.append(l2)
        acc = [None if t is None else t.double() for t in g2] if acc is None else \
              [None if a is None else a + t.double() for a, t in zip(acc, g2)]
    assert rel_err(len8, torch.cat(lens)) < 1e-4
    errs = {}
    for n, a, b in zip(names, g8, acc):
        if a is None or n.endswith('k_transform.bias') or float(b.norm()) < 1e-7 * top:
            continue
        errs[n] = rel_l2(a, b)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print(f'batch-8 path linearity pinned: {flips} slopes pinned; worst {[(k, float(f"{v:.2e}")) for k, v in worst]}')
    bad = [(k, v) for k, v in errs.items() if v > PIN_SO_TOL]
    assert not bad, bad[:8]


def test_r1_256_batch16_vs_reference_fixture(golden):
    """R1 at the timed batch (16) against the reference's 256-px discriminator fixture (batch 4), see the module docstring."""
    from transeditor_amd.model_spatial_query import Discriminator
    from transeditor_amd.op.modconv import second_order
    from transeditor_amd.train_step import d_r1_loss
    gold = golden('discriminator256_b4')
    Dn = Discriminator(256)
    sd = Dn.state_dict()
    synth.fill_state_dict(sd, 5)
    Dn.load_state_dict(sd)
    Dn = Dn.to(DEV)
    img4 = synth.normal((4, 3, 256, 256), 'd.img').clamp(-1, 1)
    idx = torch.arange(16) // 4                       # sample j + 4k of the batch of 16 = fixture sample k
    img16 = img4[idx].to(DEV).requires_grad_(True)
    with second_order():
        pred = Dn(img16)
    assert rel_err(pred.cpu(), gold['pred'][idx]) < TOL
    r1 = d_r1_loss(pred, img16)
    assert abs(float(r1) - float(gold['r1'])) / float(gold['r1']) < TOL
    gs = torch.autograd.grad(10 / 2 * r1 * 16 + 0 * pred[0], list(Dn.parameters()), allow_unused=True)
    bad = []
    for n, got, want in zip([str(k) for k in gold['r1_grad_names']], gs, gold['r1_grad_norms']):
        if want > 1e-7:
            e = abs(float(got.double().norm()) - want) / want
            if e > 5 * TOL:
                bad.append((n, e))
    assert not bad, bad[:8]
