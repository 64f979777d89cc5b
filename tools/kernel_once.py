"""Launch every hot kernel a few times at its FFHQ-256 / batch-16 top shape (for rocprofv3 --pmc runs)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transeditor_amd import _lib  # noqa: E402

DEV, B = 'cuda', 16
REP = int(os.environ.get('REP', '3'))


def main():
    x = torch.randn(B, 128, 256, 256, device=DEV)
    w = torch.randn(128, 128, 3, 3, device=DEV) / 34
    isc, osc, bias = torch.rand(B, 128, device=DEV) + 0.5, torch.rand(B, 128, device=DEV) + 0.5, torch.randn(128, device=DEV)
    from transeditor_amd.op.modconv import fwd_kinds
    pk, ck = fwd_kinds('3x3', B, w, 256, 256)                                         # the form the model runs (Winograd where it applies)
    wp = _lib.conv_pack(w, pk)
    for _ in range(REP):
        y = _lib.conv(x, wp, ck, 128, 256, 256, isc, osc, bias, 3)                    # conv3x3 fwd (fused epilogue)
    if ck == _lib.CONV_3X3W6:
        wpw = _lib.conv_pack(w, _lib.PACK_WFWD)
        for _ in range(REP):
            _lib.conv(x, wpw, _lib.CONV_3X3W, 128, 256, 256, isc, osc, bias, 3)       # the fp32 Winograd kernel at the same shape
    if ck != _lib.CONV_3X3:
        wpd = _lib.conv_pack(w, _lib.PACK_FWD)
        for _ in range(REP):
            _lib.conv(x, wpd, _lib.CONV_3X3, 128, 256, 256, isc, osc, bias, 3)        # the direct kernel at the same shape (small layers run it)
    for _ in range(REP):
        sl = _lib.wgrad_slabs(y, x, _lib.CONV_3X3, 256, 256)                           # wgrad 3x3 (split-bf16 kernel where the switch is on)
    if _lib.wgrad_split():
        _lib.wgrad_split(0)
        for _ in range(REP):
            _lib.wgrad_slabs(y, x, _lib.CONV_3X3, 256, 256)                            # the fp32 pair-form kernel at the same shape
        _lib.wgrad_split(1)
    for _ in range(REP):
        _lib.wgrad_reduce(sl, w.reshape(128, 128, 9), 1.0, isc, osc, True, True, True)
    xl = torch.randn(B, 256, 128, 128, device=DEV)
    wu = torch.randn(128, 256, 3, 3, device=DEV) / 48
    iscu = torch.rand(B, 256, device=DEV) + 0.5
    from transeditor_amd.op.modconv import bwd_kinds
    pku, cku = fwd_kinds('up', B, wu, 128, 128)                                        # the forms the model runs (split-bf16 where they apply)
    wpu = _lib.conv_pack(wu, pku)
    for _ in range(REP):
        t = _lib.conv(xl, wpu, cku, 128, 128, 128, iscu, osc)                          # T2 fwd -> [B,128,257,257]
    pks, cks = bwd_kinds('up', B, wu, 128, 128)
    wps = _lib.conv_pack(wu, pks)
    for _ in range(REP):
        _lib.conv(t, wps, cks, 256, 128, 128, osc, iscu)                               # S2 (dgrad of T2)
    if cku != _lib.CONV_T2:                                                            # the fp32 kernels at the same shapes
        wpu32 = _lib.conv_pack(wu, _lib.PACK_FWD)
        for _ in range(REP):
            _lib.conv(xl, wpu32, _lib.CONV_T2, 128, 128, 128, iscu, osc)
    if cks != _lib.CONV_S2:
        wps32 = _lib.conv_pack(wu, _lib.PACK_SWAP)
        for _ in range(REP):
            _lib.conv(t, wps32, _lib.CONV_S2, 256, 128, 128, osc, iscu)
    for _ in range(REP):
        _lib.wgrad_slabs(t, xl, _lib.CONV_T2, 128, 128)                                # wgrad T2
    if _lib.wgrad_split():
        _lib.wgrad_split(0)
        for _ in range(REP):
            _lib.wgrad_slabs(t, xl, _lib.CONV_T2, 128, 128)                            # the fp32 kernel at the same shape
        _lib.wgrad_split(1)
    k = torch.tensor([1., 3., 3., 1.], device=DEV)
    k = torch.outer(k, k) / 16
    for _ in range(REP):
        _lib.upfirdn2d_raw(t, k, (1, 1), (1, 1), (1, 1, 1, 1), bias=bias, act=3, scale=2 ** 0.5)   # blur + bias + act
    for _ in range(REP):
        _lib.bias_act_bwd(x, y, 0.2, 2 ** 0.5)
    for _ in range(REP):
        _lib.blur_actgrad(x, y, k, (1, 3, 1, 3), 0.2, 2 ** 0.5)                       # backward of blur + bias + act, one pass
    kf = torch.flip(k, [0, 1]).contiguous()
    for _ in range(REP):
        _lib.upfirdn2d_raw(x, kf, (1, 1), (1, 1), (2, 2, 2, 2))                       # adjoint blur 256^2 -> 257^2
    for _ in range(REP):
        _lib.blur_gradact(t, y, kf, (1, 1, 1, 1), 0.2, 2 ** 0.5)                      # adjoint blur 257^2 -> 256^2 + activation gradient (D ResBlock)
    wr = torch.randn(3, 128, device=DEV)
    for _ in range(REP):
        r = _lib.rgb_fwd(x, wr, isc, bias[:3].contiguous())
    for _ in range(REP):
        _lib.rgb_dgrad(r, wr, isc, 128)
    for _ in range(REP):
        _lib.rgb_wgrad_slabs(r, x)
    for _ in range(REP):
        _lib.bias_act_bwd_rgb(x, y, r, wr, isc, 1.0, 0.2, 2 ** 0.5)                   # activation gradient + ToRGB data gradient
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
