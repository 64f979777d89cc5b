"""The stack of dual-space cross-attention blocks as ONE launch per direction (te_attn_stack_fwd/bwd_f32).

Reference: Generator.forward, model_spatial_query.py:670-679 (`x = interact[0](cat(Z, eye), cat(P, eye))`, then
`x = interact[i](x, P)` for the other blocks) over AttentionBlock.forward (:920-936) / Attention.forward (:883-901).
Forward: one workgroup per sample walks through all blocks with the sample's activations in LDS (csrc/attn_block.hip).
Backward: one launch produces the input gradients and the per-layer gradient matrices; the weight / bias gradients of a
layer of ALL blocks are then one batched GEMM (te_small_gemm_batched_rs_f32) — about a dozen launches in total instead of
~80 per block.  A recorded backward (create_graph: the path-length regulariser) re-derives the gradients through the
block-by-block composition of the twice-differentiable ops (op/linear.py, op/layernorm.py, op/attention.py).
"""
import math

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib
from .attention import attention_core
from .layernorm import sample_layer_norm
from .linear import linear_fused

PLANES, GROUPS, OUT = 128, 4, 512
EPS = 1e-5
# Which form Generator.forward uses.  Measured on MI355X (profiles/README.md, FFHQ-256 generator fwd+bwd, batch 16): the
# fused stack cuts the step from 640 to 369 dispatches but costs 2.3 ms of GPU time per step against ~1.45 ms for the
# layer-by-layer launches, which spread every 7 us GEMM over 100+ CUs while a per-sample workgroup is bound by ONE CU's
# MFMA rate (25 MFLOP x 8 blocks per sample = 0.33 ms at 614 GFLOP/s per CU, before any stall).  The train step is
# GPU-bound, so the default is the faster one; `use_fused(True)` selects the single-launch form (host-bound callers).
FUSED = False


def use_fused(flag=True):
    global FUSED
    FUSED = bool(flag)

_NAMES = ('wq', 'bq', 'wk', 'bk', 'wv', 'bv', 'wp', 'bp', 'w1', 'b1', 'w2', 'b2', 'w0', 'b0')


def block_params(block):
    """the 12 (14 with the block's own skip projection) parameters of an AttentionBlock in kernel order"""
    a = block.atten
    ps = [a.q_transform.weight, a.q_transform.bias, a.k_transform.weight, a.k_transform.bias, a.v_transform.weight,
          a.v_transform.bias, a.proj.weight, a.proj.bias, block.mlp[0].weight, block.mlp[0].bias, block.mlp[2].weight,
          block.mlp[2].bias]
    if block.out_dim != block.in_dim:
        ps += [block.proj.weight, block.proj.bias]
    return ps


def _composite(x0, p0, p, blocks, lr_mul, scale):
    """the same function from the twice-differentiable single ops (also the reference of the parity tests)"""
    x = x0
    for bi, prm in enumerate(blocks):
        wq, bq, wk, bk, wv, bv, wp, bp, w1, b1, w2, b2 = prm[:12]
        pp = p0 if bi == 0 else p
        lin = lambda t, w, b, act=None, res=None: linear_fused(t, w, b, lr_mul / math.sqrt(w.shape[1]), lr_mul, act, res)
        skip = lin(x, prm[12], prm[13]) if len(prm) == 14 else x
        xn = sample_layer_norm(x)
        o, _ = attention_core(lin(pp, wq, bq), lin(xn, wk, bk), lin(xn, wv, bv), scale, GROUPS)
        x1 = lin(o, wp, bp, None, skip)
        h = lin(sample_layer_norm(x1), w1, b1, 'gelu')
        x = lin(h, w2, b2, None, x1)
    return x


class _AttnStack(Function):
    @staticmethod
    def forward(ctx, x0, p0, p, lr_mul, scale, counts, *params):
        x0, p0 = x0.contiguous(), p0.contiguous()
        p = p.contiguous() if p is not None else None
        nb, N = len(counts), x0.shape[0]
        blocks, off = [], 0
        for c in counts:
            blocks.append(list(params[off:off + c]))
            off += c
        dims = [(blk[2].shape[1], blk[0].shape[1]) for blk in blocks]          # (cin, cp) = in-features of k_ / q_transform
        flat = []
        for blk in blocks:
            flat += [t.contiguous() for t in blk] + [None] * (14 - len(blk))
        dev, dt = x0.device, x0.dtype
        need_bwd = any(ctx.needs_input_grad)
        save = None
        if need_bwd:
            mk = lambda *s: torch.empty(nb, N, *s, device=dev, dtype=dt)
            save = [mk(16, 528), mk(16, PLANES), mk(16, PLANES), mk(16, PLANES), mk(16, PLANES), mk(GROUPS, 16, 16), mk(16, OUT),
                    mk(16, OUT), mk(16, OUT), mk(16, OUT), mk(4)]
        out = _lib.attn_stack_fwd(x0, p0, p, flat, dims, lr_mul, scale, EPS, save)
        if need_bwd:
            ctx.save_for_backward(x0, p0, p, *params, *save)
        ctx.cfg = (lr_mul, scale, counts, dims)
        return out

    @staticmethod
    def backward(ctx, gout):
        lr_mul, scale, counts, dims = ctx.cfg
        nb = len(counts)
        npar = sum(counts)
        x0, p0, p = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:3 + npar]
        save = list(ctx.saved_tensors[3 + npar:])
        need = ctx.needs_input_grad
        blocks, off = [], 0
        for c in counts:
            blocks.append(list(params[off:off + c]))
            off += c
        if torch.is_grad_enabled():
            # double backward requested: differentiate the composition of the twice-differentiable single ops
            with torch.enable_grad():
                al = [t.view_as(t) if t is not None else None for t in (x0, p0, p)]
                pa = [[t.view_as(t) for t in blk] for blk in blocks]
                y = _composite(al[0], al[1], al[2], pa, lr_mul, scale)
                cand = al + [t for blk in pa for t in blk]
                flags = list(need[:3]) + list(need[6:])
                ins = [t for t, n in zip(cand, flags) if n and t is not None]
                gs = iter(torch.autograd.grad(y, ins, gout, create_graph=True, allow_unused=True))
            res = [next(gs) if (n and t is not None) else None for t, n in zip(cand, flags)]
            return tuple(res[:3]) + (None, None, None) + tuple(res[3:])
        N = x0.shape[0]
        dev, dt = gout.device, gout.dtype
        mk = lambda w: torch.empty(nb, N, 16, w, device=dev, dtype=dt)
        gmat = [mk(OUT), mk(OUT), mk(OUT), mk(PLANES), mk(PLANES), mk(PLANES)]         # g_x2, g_hpre, g_x1, g_q, g_k, g_v
        flat = []
        for blk in blocks:
            flat += [t.contiguous() for t in blk] + [None] * (14 - len(blk))
        gx0, gp0, gp = _lib.attn_stack_bwd(gout.contiguous(), flat, dims, lr_mul, scale, save, gmat, x0.shape[2], p0.shape[2])
        s_xn, s_o, s_xn1, s_h = save[0], save[4], save[7], save[9]
        g_x2, g_hpre, g_x1, g_q, g_k, g_v = gmat
        R = N * 16

        def wgrad(g, gw, act, arow, K, b_lo, b_hi, zb=None, act_off=0):
            """dW[b] = alpha * g[b]^T act[b] and db[b] = lr_mul * column sums of g[b] for blocks b_lo..b_hi-1, one launch.
            g [nb, R, J]; act: tensor whose rows (n, t) are `arow` floats apart, K used columns."""
            nz = b_hi - b_lo
            if nz <= 0:
                return None, None
            J = g.shape[-1]
            dW = torch.empty(nz, J, K, device=dev, dtype=dt)
            db = torch.empty(nz, J, device=dev, dtype=dt)
            alpha = lr_mul / math.sqrt(K)
            za = R * J
            if zb is None:
                zb = R * arow
            a = g.view(-1)[b_lo * za:]
            b = act.reshape(-1)[act_off:]
            _lib.small_gemm_batched_rs(dW, a, b, nz, za, J * K, zb, J, K, R, 1, J, arow, 1, K, 1, alpha, arowsum=db, zrs=J,
                                       rs_scale=lr_mul)
            return dW, db

        grads = [[None] * c for c in counts]

        def put(bi, wi, dW, db, z):
            if dW is not None:
                grads[bi][wi], grads[bi][wi + 1] = dW[z], db[z]

        # mlp.2 / mlp.0 / atten.proj: the same shapes in every block -> one launch each over all blocks
        for wi, g, act, arow, K in ((10, g_x2, s_h, OUT, OUT), (8, g_hpre, s_xn1, OUT, OUT), (6, g_x1, s_o, PLANES, PLANES)):
            dW, db = wgrad(g, None, act, arow, K, 0, nb)
            for bi in range(nb):
                put(bi, wi, dW, db, bi)
        # k / v / q transforms: block 0 may be 528 wide, the others are 512 wide and (for q) share the same P
        for wi, g in ((2, g_k), (4, g_v)):
            dW, db = wgrad(g, None, s_xn, 528, dims[0][0], 0, 1)
            put(0, wi, dW, db, 0)
            dW, db = wgrad(g, None, s_xn, 528, OUT, 1, nb, act_off=R * 528)
            for bi in range(1, nb):
                put(bi, wi, dW, db, bi - 1)
        dW, db = wgrad(g_q, None, p0, dims[0][1], dims[0][1], 0, 1)
        put(0, 0, dW, db, 0)
        if nb > 1:
            dW, db = wgrad(g_q, None, p, OUT, OUT, 1, nb, zb=0)
            for bi in range(1, nb):
                put(bi, 0, dW, db, bi - 1)
        if counts[0] == 14:        # block 0's skip projection: dW0 = alpha0 g_x1[0]^T x0
            dW, db = wgrad(g_x1, None, x0, dims[0][0], dims[0][0], 0, 1)
            put(0, 12, dW, db, 0)
        flat_g = [t for blk in grads for t in blk]
        flat_g = [t if n else None for t, n in zip(flat_g, need[6:])]
        return (gx0 if need[0] else None, gp0 if need[1] else None, gp if (need[2] and gp is not None) else None,
                None, None, None) + tuple(flat_g)


def attention_stack(x0, p0, p, blocks, lr_mul, scale, second_order=False):
    """x0 [N,16,cin0], p0 [N,16,cp0], p [N,16,512] (None for a single block); blocks: per block the list from
    block_params().  Returns the last block's output [N,16,512]."""
    for t in (x0, p0):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError('te_hip: expected fp32 tensors on the GPU (no CPU path exists)')
    if second_order and torch.is_grad_enabled():
        return _composite(x0, p0, p, blocks, lr_mul, scale)
    counts = tuple(len(b) for b in blocks)
    flat = [t for b in blocks for t in b]
    return _AttnStack.apply(x0, p0, p, float(lr_mul), float(scale), counts, *flat)


def supported(blocks):
    """shapes the fused kernels cover: 16 tokens, planes 128 / 4 heads, output 512, first block <= 528 wide (multiples of 16)"""
    if not 1 <= len(blocks) <= 8:
        return False
    for i, b in enumerate(blocks):
        a = b.atten
        if (b.out_dim, a.planes, a.groups) != (OUT, PLANES, GROUPS):
            return False
        if b.in_dim % 16 or b.param_dim % 16 or b.in_dim > 528 or b.param_dim > 528:
            return False
        if i > 0 and (b.in_dim != OUT or b.param_dim != OUT):
            return False
    return True
