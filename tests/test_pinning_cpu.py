"""The pinned-slope test infrastructure itself (tests/pinning.py) on the CPU tier: the wrappers sit on the `_lib` binding, so
they can be driven with stand-in binding functions - matching by sign agreement (scale-invariant), the in-place sign fix, the
unmatched-activation report, which call sites count as a fused leaky-ReLU, stacking of per-token masks, batch slices."""
import torch

import pinning
from transeditor_amd import _lib


def _fake_lib(monkeypatch, outs):
    """binding stand-ins that return prepared tensors (on the CPU; the bank lives wherever pinning.DEV says)"""
    monkeypatch.setattr(pinning, 'DEV', 'cpu')
    it = iter(outs)
    monkeypatch.setattr(_lib, 'conv', lambda x, wp, kind, M, H, W, isc=None, osc=None, bias=None, act=0, res=None, mask_ref=None,
                        mask_gain=1.0: next(it))
    monkeypatch.setattr(_lib, 'bias_act', lambda x, b, ref, act, grad, alpha, scale: next(it))
    monkeypatch.setattr(_lib, 'small_gemm', lambda I, J, K, a, sai, sak, b, sbk, sbj, bias=None, residual=None, alpha=1.0, beta=1.0,
                        act=0, want_pre=False, rowsum_scale=None: (next(it), None, None))


def test_pinning_matches_by_sign_and_fixes_flips(monkeypatch):
    torch.manual_seed(0)
    ref = torch.randn(2, 8, 4, 4)
    other = torch.randn(2, 8, 4, 4)                       # same shape, unrelated signs: must not be chosen
    bank = pinning.SignBank()
    monkeypatch.setattr(pinning, 'DEV', 'cpu')
    bank.add(other)
    bank.add(ref)
    ours = ref * 0.70710678                               # a folded constant (ResBlock's 1/sqrt(2)): signs agree, values do not
    ours[0, 0, 0, 0] = -ours[0, 0, 0, 0] * 1e-7           # one slope flip near the kink
    ours[1, 3, 2, 1] = -ours[1, 3, 2, 1] * 1e-7
    want_sign = ref > 0
    _fake_lib(monkeypatch, [ours, torch.randn(3, 5), ours.clone(), torch.randn(2, 8, 4, 4)])
    with pinning.pinned(bank) as st:
        a = _lib.conv(None, None, 0, 8, 4, 4, None, None, None, 3)                   # fused leaky-ReLU: visited
        b = _lib.conv(None, None, 0, 8, 4, 4, None, None, None, 0)                   # no activation: not visited
        c = _lib.bias_act(None, None, None, 3, 1, 0.2, 1.0)                          # grad mode: a backward call, not visited
        d = _lib.small_gemm(2, 8, 4, None, 1, 1, None, 1, 1, act=3)[0]               # activation without a twin in the bank
    assert st['activations'] == 2 and st['flips'] == 2 and st['unmatched'] == [(2, 8, 4, 4)]
    assert torch.equal(a > 0, want_sign)                  # signs now the bank's; untouched elements keep their values
    keep = torch.ones_like(ref, dtype=torch.bool)
    keep[0, 0, 0, 0] = keep[1, 3, 2, 1] = False
    assert float((a - ref * 0.70710678).abs()[keep].max()) < 1e-6 and float(a.abs()[~keep].max()) < 2e-30
    assert b.shape == (3, 5) and c is not None and d is not None
    assert _lib.conv.__name__ == '<lambda>'               # the wrappers are gone again


def test_sign_bank_stacking_slices_and_capture(monkeypatch):
    monkeypatch.setattr(pinning, 'DEV', 'cpu')
    torch.manual_seed(1)
    bank = pinning.SignBank()
    toks = [torch.randn(3, 6) for _ in range(16)]         # the oracle maps 16 tokens one by one: [B, 512] each
    for t in toks:
        bank.add(t)
    bank.add(torch.randn(3, 2, 4, 4))
    bank.extend_stacked(16, dim=1)
    stacked = torch.stack(toks, dim=1)                    # the HIP path writes [B, 16, 512] in one launch
    ref, pos = bank.match(stacked * 3.0)
    assert ref is not None and torch.equal(ref, stacked > 0)
    sub = bank.batch_slice(slice(1, 3), 3)
    assert all(m.shape[0] == 2 for m in sub.masks) and len(sub.masks) == len(bank.masks)
    rep = bank.mapped(lambda m: m[torch.tensor([0, 0, 1, 1, 2, 2])])
    assert rep.masks[0].shape[0] == 6
    # capture(): the same call sites, recording instead of pinning
    out = torch.randn(2, 4)
    _fake_lib(monkeypatch, [out])
    with pinning.capture() as got:
        _lib.conv(None, None, 0, 4, 1, 1, None, None, None, 4)
    assert len(got.masks) == 1 and torch.equal(got.masks[0], out > 0)


def test_oracle_tap_records_every_leaky_relu():
    from oracle import te_oracle as O
    x, b = torch.randn(2, 4, 3, 3), torch.randn(4)
    old_dev = pinning.DEV
    pinning.DEV = 'cpu'
    try:
        with pinning.record_oracle() as bank:
            y = O.fused_leaky_relu(x, b)
            O.equal_linear(torch.randn(2, 8), torch.randn(5, 8), torch.randn(5), activation=True)
        assert O.ACT_TAP is None and len(bank.masks) == 2 and torch.equal(bank.masks[0], y > 0)
    finally:
        pinning.DEV = old_dev
