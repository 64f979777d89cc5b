"""Per-shape time of the MFMA convolution launches inside the training iteration (HIP events around every te_conv / te_wgrad
call of 16 iterations): which (kind, K -> M @ HxW, B) shapes carry the time and at what rate each runs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib, synth       # noqa: E402
from transeditor_amd.train_step import TrainStep, default_args       # noqa: E402

DEV = 'cuda'
recs = []
names = {_lib.CONV_3X3: 'conv3x3', _lib.CONV_T2: 'convT2', _lib.CONV_S2: 'convS2', _lib.CONV_1X1: 'conv1x1', _lib.CONV_3X3W: 'conv3x3w',
         _lib.CONV_3X3W6: 'conv3x3w6', _lib.CONV_S2S6: 'convS2s6', _lib.CONV_T2S6: 'convT2s6', _lib.CONV_1X1S6: 'conv1x1s6'}
orig_conv, orig_wgrad = _lib.conv, _lib.wgrad_slabs
ON = [False]


def conv(x, wp, kind, M, H, W, *a, **k):
    if not ON[0]:
        return orig_conv(x, wp, kind, M, H, W, *a, **k)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = orig_conv(x, wp, kind, M, H, W, *a, **k)
    e.record()
    taps = 1 if kind in (_lib.CONV_1X1, _lib.CONV_1X1S6) else 9
    recs.append(((names[kind], x.shape[0], x.shape[1], M, H, W), 2.0 * taps * x.shape[1] * M * H * W * x.shape[0], s, e))
    return out


def wgrad(g, x, kind, H, W, *a, **k):
    if not ON[0]:
        return orig_wgrad(g, x, kind, H, W, *a, **k)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = orig_wgrad(g, x, kind, H, W, *a, **k)
    e.record()
    taps = 1 if kind in (_lib.CONV_1X1, _lib.CONV_1X1S6) else 9
    recs.append((('wgrad_' + names[kind], g.shape[0], x.shape[1], g.shape[1], H, W), 2.0 * taps * g.shape[1] * x.shape[1] * H * W * g.shape[0], s, e))
    return out


_lib.conv, _lib.wgrad_slabs = conv, wgrad


def main():
    torch.cuda.set_device(0)
    targs = default_args(size=256, batch=16)
    ts = TrainStep(targs, DEV)
    torch.manual_seed(1)
    reals = [torch.randn(16, 3, 256, 256, device=DEV).clamp(-1, 1) for i in range(2)]
    for i in range(3):
        ts.iteration(100 + i, reals[i % 2])
    torch.cuda.synchronize()
    ON[0] = True
    for i in range(16):
        ts.iteration(i, reals[i % 2])
    torch.cuda.synchronize()
    agg = {}
    for key, fl, s, e in recs:
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1; a[1] += fl; a[2] += s.elapsed_time(e)
    tot = sum(a[2] for a in agg.values())
    print(f'{len(recs)} launches, {tot:.1f} ms over 16 iterations')
    print(f'{"kind":14s} {"B":>3s} {"K":>4s} {"M":>4s} {"H":>4s} {"W":>4s} {"n":>5s} {"ms":>8s} {"share":>6s} {"TF/s":>7s}')
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:110]:
        print(f'{key[0]:14s} {key[1]:3d} {key[2]:4d} {key[3]:4d} {key[4]:4d} {key[5]:4d} {a[0]:5d} {a[2]:8.1f} {100 * a[2] / tot:5.1f}% {a[1] / a[2] / 1e9:7.1f}')


if __name__ == '__main__':
    main()
