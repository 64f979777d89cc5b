"""TE_CONV_1X1S6 (csrc/p1s6.hip) against the fp32 1x1 kernel at the discriminator's skip-branch shapes (forward and data gradient, batch 32 =
the joint pass of the D step, and batch 16): deviation from each other and HIP-event time.   python tools/p1s6_check.py"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transeditor_amd import _lib      # noqa: E402
from tools.exp_time import timeit     # noqa: E402

DEV = 'cuda'
if os.environ.get('VARIANT'):
    _lib.LIB_PATH = os.path.join(ROOT, 'tools', 'exp', f"libte_{os.environ['VARIANT']}.so")


def main():
    shapes = [(32, 128, 256, 128), (32, 256, 512, 64), (32, 512, 512, 32), (32, 512, 512, 16), (16, 128, 256, 128), (16, 256, 512, 64),
              (32, 256, 128, 128), (32, 512, 256, 64)]          # (the last two: data gradients, K = Co, M = Ci)
    for B, K, M, H in shapes:
        ok = _lib.p1s6_ok(B, K, M, H, H)
        x = torch.randn(B, K, H, H, device=DEV)
        w = torch.randn(M, K, 1, 1, device=DEV) / math.sqrt(K)
        res = torch.randn(B, M, H, H, device=DEV)
        w32 = _lib.conv_pack(w, _lib.PACK_FWD)
        f32 = lambda: _lib.conv(x, w32, _lib.CONV_1X1, M, H, H, res=res)
        flops = 2.0 * K * M * H * H * B
        t32 = timeit(f32, n=20)
        msg = f'B{B} {K}->{M} @{H}x{H}: fp32 kernel {t32 * 1e3:7.1f} us {flops / t32 / 1e9:6.1f} TF/s'
        if ok:
            w6 = _lib.conv_pack(w, _lib.PACK_P6FWD)
            f6 = lambda: _lib.conv(x, w6, _lib.CONV_1X1S6, M, H, H, res=res)
            a, b = f6(), f32()
            d = float((a.double() - b.double()).norm() / b.double().norm())
            t6 = min(timeit(f6, n=20), timeit(f6, n=20))
            msg += f' | split {t6 * 1e3:7.1f} us {flops / t6 / 1e9:6.1f} TF/s ({6 * flops / t6 / 1e9:6.0f} executed) | split vs fp32 kernel (L2) {d:.2e}'
        else:
            msg += ' | not covered'
        print(msg, flush=True)


if __name__ == '__main__':
    main()
