"""All style modulations of a synthesis pass as a few batched launches (te_small_gemm_batched_f32).

Reference: every ModulatedConv2d owns `self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)`
(model_spatial_query.py:286) and evaluates it at the start of its forward (:299) on the layer's latent `latent[:, i]`
(Generator.forward :702-714): 20 tiny GEMMs [B,512] x [512,Cin] per pass at 256 px (26 at 1024 px), each followed in the
backward by a dx and a dW launch and by an accumulation of the latent's gradient.  None of them depends on the synthesis
itself - only on the latent - so here they run up front, grouped by width (three widths at 256 px):

    forward    one launch per width group over the gathered latents      (20 launches -> 3)
    backward   one dx launch + one dW launch + one bias reduction per group; the scatter back onto the latent entries is the
               backward of ONE index_select                                (40 launches + ~20 accumulation adds -> ~10)

A backward that is itself recorded (create_graph) goes layer by layer through the closed small-GEMM trio of op/linear.py;
under `second_order()` the model does not take this route at all.
"""
import torch
from torch.autograd import Function

from .. import _lib
from .linear import _closed_expr


def _offsets(ts):
    base = ts[0].data_ptr()
    return [(t.data_ptr() - base) // 4 for t in ts]


def supported(latent, mods):
    return (latent.is_cuda and latent.dtype == torch.float32 and latent.dim() == 3
            and all(m.activation is None and m.bias is not None and m.weight.is_contiguous() and m.bias.is_contiguous()
                    and m.weight.shape[1] == latent.shape[2] and m.scale == mods[0].scale and m.lr_mul == mods[0].lr_mul
                    and m.weight.device == latent.device for m in mods))


def _groups(widths):
    """layers grouped by output width, at most 16 per launch (the kernel's per-launch parameter table): [(width, [layer ids])]"""
    by = {}
    for i, w in enumerate(widths):
        by.setdefault(w, []).append(i)
    out = []
    for w, ids in by.items():
        for c in range(0, len(ids), 16):
            out.append((w, ids[c:c + 16]))
    return out


class _BatchedModulation(Function):
    @staticmethod
    def forward(ctx, lat_g, alpha, beta, n, *params):
        """lat_g [B, Z, K]: the latent of every modulation, GROUP-MAJOR (the caller gathered it in the order of `_groups`);
        params = Z weights [J_z, K] then Z biases [J_z], in the same order.  Returns Z tensors [B, J_z]."""
        weights, biases = params[:n], params[n:]
        lat_in, lat_g = lat_g, lat_g.contiguous()       # saved: the input itself (history for a recorded backward)
        B, Z, K = lat_g.shape
        groups = _groups([w.shape[0] for w in weights])
        outs = [None] * n
        z0 = 0
        for J, ids in groups:
            assert ids == list(range(z0, z0 + len(ids))), 'parameters must arrive group-major'
            nz = len(ids)
            buf = torch.empty(nz, B, J, device=lat_g.device, dtype=lat_g.dtype)
            ws, bs = [weights[i] for i in ids], [biases[i] for i in ids]
            # A_z(b, k) = lat_g[b, z0 + z, k];  B_z(k, j) = W_z[j, k];  C_z(b, j) = buf[z, b, j]
            _lib.small_gemm_batched(buf, lat_g[:, z0:], ws[0], bs[0], nz, K, B * J, B, J, K, Z * K, 1, 1, K, J, 1,
                                    b_tab=_offsets(ws), bias_tab=_offsets(bs), alpha=alpha, beta=beta, act=0)
            for z, i in enumerate(ids):
                outs[i] = buf[z]
            z0 += nz
        ctx.save_for_backward(lat_in, *params)
        ctx.cfg = (alpha, beta, n, groups)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        lat_g, *params = ctx.saved_tensors
        alpha, beta, n, groups = ctx.cfg
        weights, biases = params[:n], params[n:]
        need = ctx.needs_input_grad
        B, Z, K = lat_g.shape
        if torch.is_grad_enabled():              # recorded backward: layer by layer through the closed trio
            with torch.enable_grad():
                la = lat_g.view_as(lat_g)
                pa = [p.view_as(p) for p in params]
                ys = [_closed_expr(la[:, z], pa[z], pa[n + z], alpha, beta, None, None) for z in range(n)]
                ins = [t for t, f in zip([la] + pa, [need[0]] + list(need[4:])) if f]
                res = iter(torch.autograd.grad(ys, ins, list(gs), create_graph=True, allow_unused=True))
            glat = next(res) if need[0] else None
            return (glat, None, None, None) + tuple(next(res) if f else None for f in need[4:])
        lat_g = lat_g.contiguous()
        glat = torch.empty_like(lat_g) if need[0] else None
        gws, gbs = [None] * n, [None] * n
        z0 = 0
        for J, ids in groups:
            nz = len(ids)
            g = torch.stack([gs[i] for i in ids])                      # [nz, B, J]
            ws = [weights[i] for i in ids]
            if glat is not None:      # dlat_g[b, z0 + z, k] = alpha * sum_j g[z, b, j] W_z[j, k]
                _lib.small_gemm_batched(glat[:, z0:], g, ws[0], None, nz, B * J, K, B, K, J, J, 1, K, 1, Z * K, 1,
                                        b_tab=_offsets(ws), alpha=alpha)
            if any(need[4 + i] for i in ids):      # dW_z[j, k] = alpha * sum_b g[z, b, j] lat_g[b, z0 + z, k]
                gw = torch.empty(nz, J, K, device=g.device, dtype=g.dtype)
                _lib.small_gemm_batched(gw, g, lat_g[:, z0:], None, nz, B * J, J * K, J, K, B, 1, J, Z * K, 1, K, 1, zb=K, alpha=alpha)
                for z, i in enumerate(ids):
                    gws[i] = gw[z] if need[4 + i] else None
            if any(need[4 + n + i] for i in ids):
                gb = g.sum(dim=1)
                gb = gb * beta if beta != 1.0 else gb
                for z, i in enumerate(ids):
                    gbs[i] = gb[z] if need[4 + n + i] else None
            z0 += nz
        return (glat, None, None, None) + tuple(gws) + tuple(gbs)


_SEL = {}          # (device, latent entries in group-major order) -> index tensor: built once (never inside a hipGraph capture)


def batched_modulation(latent, mods, index):
    """latent [B, L, K]; mods: EqualLinear modules (activation None, with bias); index[i] = the latent entry layer i reads.
    Returns the list of style scales s_i = mods[i](latent[:, index[i]])  [B, J_i]."""
    n = len(mods)
    widths = [m.weight.shape[0] for m in mods]
    order = [i for _, ids in _groups(widths) for i in ids]                 # group-major order of the layers
    key = (latent.device, tuple(index[i] for i in order))
    sel = _SEL.get(key)
    if sel is None:
        sel = _SEL[key] = torch.as_tensor(key[1], device=latent.device)
    lat_g = latent.index_select(1, sel)                                    # ONE gather; its backward is the scatter-add
    outs = _BatchedModulation.apply(lat_g, float(mods[0].scale), float(mods[0].lr_mul), n,
                                    *[mods[i].weight for i in order], *[mods[i].bias for i in order])
    res = [None] * n
    for pos, i in enumerate(order):
        res[i] = outs[pos]
    return res
