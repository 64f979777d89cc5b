// A1 + A2 fused: the whole stack of dual-space cross-attention blocks in ONE launch (forward) — one workgroup per sample.
//
// Reference: AttentionBlock.forward + Attention.forward, model_spatial_query.py:883-901, 920-936, chained n_trans times by
// Generator.forward (:670-679).  Per block, with x [16 tokens, C] the Z-space code and P [16, Cp] the P-space code:
//     xn = LN(x)                      (over all 16 x C elements of the sample, eps 1e-5, no affine)
//     q = Lq(P), k = Lk(xn), v = Lv(xn)                      EqualLinear: y = (lr_mul / sqrt(K)) x W^T + lr_mul b
//     o = softmax(q k^T * 128^-0.5) v   per head (4 x 32)
//     x1 = Lproj(o) + (Lskip(x) if C != 512 else x)
//     x2 = L2(GELU(L1(LN(x1)))) + x1
// The unfused form is ~26 launches per block (7 us GEMMs, layer norms, attention core, residual adds): 0.2 % of the FLOPs
// but most of the step's dispatches.  Here a sample's activations never leave the CU: a 512-thread workgroup keeps
// x / xn / x1 / h (16 x 528 fp32 each) and q / k / v / o (16 x 128) in LDS (147 KB of the 160 KB), streams every weight
// matrix once from L2 straight into MFMA B operands (v_mfma_f32_16x16x4_f32, exact fp32; a wave owns 16-column tiles of
// the output and reads 16 bytes per lane and load, 16 loads in flight per wave), runs the layer norms as workgroup reductions and the attention core
// on four of its waves, and walks through all n_trans blocks.  What the backward needs (normalised inputs, q/k/v/o, the
// attention matrix, x1, the pre-GELU activations, LN statistics) is written to caller-provided save buffers.
// LDS rows are 584 / 136 floats apart (= 8 mod 64): the 16-byte A-operand reads of the 16 x 4 lane grid are conflict-free.
#include "te_common.h"
#include <math.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 512, NWAVES = THREADS / 64;     // 8 waves = 2 per SIMD: 256 VGPRs each for the in-flight weight loads
constexpr int T = 16;                 // tokens
constexpr int CO = 512;               // block output width
constexpr int PL = 128;               // planes (q / k / v width), 4 heads x 32
constexpr int SB = 584;               // LDS row stride of the wide buffers (>= 528, = 8 mod 64)
constexpr int SS = 136;               // LDS row stride of the q / k / v / o buffers
constexpr int MAXBLK = 8;

struct BlockW {                       // device pointers of one AttentionBlock's parameters (state_dict names in te_hip.h)
    const float *wq, *bq, *wk, *bk, *wv, *bv, *wp, *bp, *w1, *b1, *w2, *b2, *w0, *b0;
    int cin, cp;                      // width of x / of P seen by this block (528 for block 0, else 512)
};

struct StackArgs {
    float* xout;                      // [N, 16, 512] output of the last block
    const float* x0;                  // [N, 16, cin0]
    const float* p0;                  // [N, 16, cp0]   P as seen by block 0 (with the one-hot tokens appended)
    const float* p;                   // [N, 16, 512]   P as seen by the other blocks
    // save buffers for the backward (all may be NULL: inference), laid out [block][N][16][width]
    float *s_xn, *s_q, *s_k, *s_v, *s_o, *s_sim, *s_x1, *s_xn1, *s_hpre, *s_h, *s_stats;
    float* sim_out;                   // optional [block][N][4][16][16] (return_similarity); may alias s_sim
    int N, nblocks;
    float lr_mul, attn_scale, eps;
    BlockW blk[MAXBLK];
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) t += red[w];
    return t;
}

// C[16][J] = alpha * A[16][K] W[J][K]^T for the 16-column tiles this wave owns (tiles go round the waves);
// epi(row, col, value) is called for the 4 results of every lane.  A lives in LDS (row stride sa), W in global memory / L2.
// Lane t = (lo = t & 15, hi = t >> 4) supplies A[lo][k0 + 4 hi + i] and W[j0 + lo][k0 + 4 hi + i] to MFMA i of a 16-wide k step
// and receives C[4 hi + r][j0 + lo].  K % 16 == 0, J % 16 == 0.  The kernel is bound by the latency of the weight stream
// (16 workgroups x 3 MB per block out of L2), so a wave issues the 16-byte loads of CH k-steps back to back before it
// touches the first one: 8 KB in flight per wave, 128 KB per CU.
#ifndef TE_ATT_CH
#define TE_ATT_CH 16
#define TE_ATT_CHT 16
#endif
constexpr int CH = TE_ATT_CH;

__device__ __forceinline__ void gemm16_acc(f32x4& acc, const float* __restrict__ A, int sa, const float* __restrict__ W, int K, int j0) {
    const int t = threadIdx.x & 63, lo = t & 15, hi = t >> 4;
    const float* ar = A + lo * sa + 4 * hi;
    const float* wr = W + (size_t)(j0 + lo) * K + 4 * hi;
    for (int kc = 0; kc < K; kc += 16 * CH) {
        f32x4 b[CH];
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int k = kc + 16 * s;
            b[s] = *reinterpret_cast<const f32x4*>(wr + (k < K ? k : 0));       // clamped: the tail step is skipped below
        }
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int k = kc + 16 * s;
            if (k < K) {                                                          // wave-uniform
                const f32x4 a = *reinterpret_cast<const f32x4*>(ar + k);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[s][i], acc, 0, 0, 0);
            }
        }
    }
}

template <typename Epi>
__device__ __forceinline__ void gemm16(const float* __restrict__ A, int sa, const float* __restrict__ W, int K, int J, float alpha,
                                       const float* __restrict__ A2, int sa2, const float* __restrict__ W2, int K2, float alpha2,
                                       int wave0, int nw, Epi epi) {
    const int t = threadIdx.x & 63, lo = t & 15, hi = t >> 4;
    const int wid = (threadIdx.x >> 6) - wave0;
    if (wid < 0 || wid >= nw) return;
    for (int j0 = wid * 16; j0 < J; j0 += nw * 16) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        gemm16_acc(acc, A, sa, W, K, j0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= alpha;
        if (A2) {                                                                 // a second product into the same tile
            f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
            gemm16_acc(acc2, A2, sa2, W2, K2, j0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += acc2[r] * alpha2;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) epi(4 * hi + r, j0 + lo, acc[r]);
    }
}

// LN over the 16 x C elements of buf (row stride SB) -> dst (row stride SB, may alias buf); optional global copy
__device__ __forceinline__ void layer_norm16(float* dst, const float* src, int C, float eps, float* red, float* g_copy, int g_stride, float* g_stats) {
    const int n = T * C;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += THREADS) s += src[(e / C) * SB + e % C];
    const float mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    for (int e = threadIdx.x; e < n; e += THREADS) {
        const float d = src[(e / C) * SB + e % C] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(block_sum(q, red) / (float)n + eps);
    for (int e = threadIdx.x; e < n; e += THREADS) {
        const int r = e / C, c = e % C;
        const float y = (src[r * SB + c] - mean) * rstd;
        dst[r * SB + c] = y;
        if (g_copy) g_copy[r * g_stride + c] = y;
    }
    if (g_stats && threadIdx.x == 0) { g_stats[0] = mean; g_stats[1] = rstd; }
    __syncthreads();
}

__global__ __launch_bounds__(THREADS) void attn_stack_fwd_kernel(const StackArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufX = smem;                    // x, later LN(x1), later x2
    float* bufN = bufX + T * SB;           // P, later LN(x), later x1
    float* bufH = bufN + T * SB;           // GELU(L1(.))
    float* bufQ = bufH + T * SB;
    float* bufK = bufQ + T * SS;
    float* bufV = bufK + T * SS;
    float* bufO = bufV + T * SS;
    float* ps = bufO + T * SS;             // [4 heads][16][17] softmax scratch
    float* red = ps + 4 * 16 * 17;         // [NWAVES]

    const int n = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, t = tid & 63, lo = t & 15, hi = t >> 4;

    // block 0 input
    {
        const int C = p.blk[0].cin;
        const float* xg = p.x0 + (size_t)n * T * C;
        for (int e = tid; e < T * C; e += THREADS) bufX[(e / C) * SB + e % C] = xg[e];
    }
    for (int bi = 0; bi < p.nblocks; ++bi) {
        const BlockW& w = p.blk[bi];
        const int C = w.cin, CP = w.cp;
        const size_t sn = (size_t)bi * p.N + n;          // (block, sample) index into the save buffers
        // ---- P -> bufN, block input -> save
        {
            const float* pg = (bi == 0 ? p.p0 : p.p) + (size_t)n * T * CP;
            for (int e = tid; e < T * CP; e += THREADS) bufN[(e / CP) * SB + e % CP] = pg[e];
        }
        __syncthreads();
        // ---- q = Lq(P)
        {
            const float alpha = p.lr_mul * rsqrtf((float)CP);
            float* sq = p.s_q ? p.s_q + sn * T * PL : nullptr;
            gemm16(bufN, SB, w.wq, CP, PL, alpha, nullptr, 0, nullptr, 0, 0.f, 0, NWAVES, [&](int r, int c, float v) {
                v += w.bq[c] * p.lr_mul;
                bufQ[r * SS + c] = v;
                if (sq) sq[r * PL + c] = v;
            });
        }
        __syncthreads();
        // ---- xn = LN(x) -> bufN
        layer_norm16(bufN, bufX, C, p.eps, red, p.s_xn ? p.s_xn + sn * T * 528 : nullptr, 528, p.s_stats ? p.s_stats + sn * 4 : nullptr);
        // ---- k, v = Lk(xn), Lv(xn): the first half of the waves takes k, the second half v
        {
            const float alpha = p.lr_mul * rsqrtf((float)C);
            float* sk = p.s_k ? p.s_k + sn * T * PL : nullptr;
            float* sv = p.s_v ? p.s_v + sn * T * PL : nullptr;
            gemm16(bufN, SB, w.wk, C, PL, alpha, nullptr, 0, nullptr, 0, 0.f, 0, NWAVES / 2, [&](int r, int c, float v) {
                v += w.bk[c] * p.lr_mul;
                bufK[r * SS + c] = v;
                if (sk) sk[r * PL + c] = v;
            });
            gemm16(bufN, SB, w.wv, C, PL, alpha, nullptr, 0, nullptr, 0, 0.f, NWAVES / 2, NWAVES / 2, [&](int r, int c, float v) {
                v += w.bv[c] * p.lr_mul;
                bufV[r * SS + c] = v;
                if (sv) sv[r * PL + c] = v;
            });
        }
        __syncthreads();
        // ---- attention core: wave g < 4 owns head g (Attention.forward :888-894; scale = planes^-0.5, :873)
        if (wave < 4) {
            const int g = wave;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const float a = bufQ[lo * SS + g * 32 + 4 * st + hi];
                const float b = bufK[lo * SS + g * 32 + 4 * st + hi];
                s = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, s, 0, 0, 0);
            }
            float* psg = ps + g * 16 * 17;
            float* simg = p.s_sim ? p.s_sim + (sn * 4 + g) * 256 : nullptr;
            float* simo = p.sim_out ? p.sim_out + (sn * 4 + g) * 256 : nullptr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {            // lane holds S[m = 4 hi + r][l = lo]; a row lives in 16 lanes of a quarter wave
                const float x = s[r] * p.attn_scale;
                float mx = x;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                const float e = expf(x - mx);
                float sm = e;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o, 64);
                const float pr = e / sm;
                psg[(4 * hi + r) * 17 + lo] = pr;
                if (simg) simg[(4 * hi + r) * 16 + lo] = pr;
                if (simo && simo != simg) simo[(4 * hi + r) * 16 + lo] = pr;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's LDS writes have landed (the wave is the only reader)
            __builtin_amdgcn_wave_barrier();
            float* so = p.s_o ? p.s_o + sn * T * PL : nullptr;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const float a = psg[lo * 17 + 4 * st + hi];
                    const float b = bufV[(4 * st + hi) * SS + g * 32 + nb * 16 + lo];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bufO[(4 * hi + r) * SS + g * 32 + nb * 16 + lo] = acc[r];
                    if (so) so[(4 * hi + r) * PL + g * 32 + nb * 16 + lo] = acc[r];
                }
            }
        }
        __syncthreads();
        // ---- x1 = Lproj(o) + skip -> bufN   (skip = Lskip(x) for the 528-wide block, else x)
        {
            const float alpha = p.lr_mul * rsqrtf((float)PL);
            const bool proj = (C != CO);
            const float alpha0 = proj ? p.lr_mul * rsqrtf((float)C) : 0.f;
            float* sx1 = p.s_x1 ? p.s_x1 + sn * T * CO : nullptr;
            gemm16(bufO, SS, w.wp, PL, CO, alpha, proj ? bufX : nullptr, SB, proj ? w.w0 : nullptr, C, alpha0, 0, NWAVES,
                   [&](int r, int c, float v) {
                       v += w.bp[c] * p.lr_mul;
                       v += proj ? w.b0[c] * p.lr_mul : bufX[r * SB + c];
                       bufN[r * SB + c] = v;
                       if (sx1) sx1[r * CO + c] = v;
                   });
        }
        __syncthreads();
        // ---- LN(x1) -> bufX
        layer_norm16(bufX, bufN, CO, p.eps, red, p.s_xn1 ? p.s_xn1 + sn * T * CO : nullptr, CO, p.s_stats ? p.s_stats + sn * 4 + 2 : nullptr);
        // ---- h = GELU(L1(LN(x1))) -> bufH
        {
            const float alpha = p.lr_mul * rsqrtf((float)CO);
            float* shp = p.s_hpre ? p.s_hpre + sn * T * CO : nullptr;
            float* sh = p.s_h ? p.s_h + sn * T * CO : nullptr;
            gemm16(bufX, SB, w.w1, CO, CO, alpha, nullptr, 0, nullptr, 0, 0.f, 0, NWAVES, [&](int r, int c, float v) {
                v += w.b1[c] * p.lr_mul;
                const float h = gelu_erf(v);
                bufH[r * SB + c] = h;
                if (shp) shp[r * CO + c] = v;
                if (sh) sh[r * CO + c] = h;
            });
        }
        __syncthreads();
        // ---- x2 = L2(h) + x1 -> bufX (next block's input) / global output of the last block
        {
            const float alpha = p.lr_mul * rsqrtf((float)CO);
            const bool last = bi == p.nblocks - 1;
            float* xo = p.xout + (size_t)n * T * CO;
            gemm16(bufH, SB, w.w2, CO, CO, alpha, nullptr, 0, nullptr, 0, 0.f, 0, NWAVES, [&](int r, int c, float v) {
                v += w.b2[c] * p.lr_mul + bufN[r * SB + c];
                bufX[r * SB + c] = v;
                if (last) xo[r * CO + c] = v;
            });
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Backward of the stack, one workgroup per sample, blocks in reverse order.  It produces the input gradients (x0, P of
// block 0, P of the other blocks) and, per block, the gradient MATRICES of the six (seven) linear layers' outputs
// (g_x2, g_hpre, g_x1, g_q, g_k, g_v: [block][N][16][width]); the weight / bias gradients are then plain batched GEMMs
// over all samples (dW = alpha * g^T act, host side, te_small_gemm_batched_f32) — a per-sample workgroup cannot own them.
struct BwdArgs {
    const float* gout;                // [N, 16, 512]
    float *gx0, *gp0, *gp;            // [N, 16, cin0], [N, 16, cp0], [N, 16, 512] (gp: sum over the blocks after the first)
    const float *s_xn, *s_q, *s_k, *s_v, *s_sim, *s_xn1, *s_hpre, *s_stats;
    float *g_x2, *g_hpre, *g_x1, *g_q, *g_k, *g_v;
    int N, nblocks;
    float lr_mul, attn_scale;
    BlockW blk[MAXBLK];
};

// out[16][Kout] = alpha * A[16][J] W[J][Kout]  (W row-major [J][ldw]): the data gradient of y = x W^T, reading W as stored.
// A wave owns 16 * NQ consecutive output columns as NQ interleaved tiles: lane (lo, hi) loads the NQ consecutive weights
// W[j0 + hi][kb + NQ lo .. + NQ - 1] (one 4 / 8 / 16-byte load) and feeds value q to the tile of the columns kb + NQ lo + q, with
// A[lo][j0 + hi] (one LDS read) as the other operand of all NQ MFMAs of the 4-deep j step.  Loads of CHT j steps are issued
// together.  Kout % 16 == 0 (columns beyond Kout are masked), J % 4 == 0.
constexpr int CHT = TE_ATT_CHT;

template <int NQ>
__device__ __forceinline__ void gemm16_t_acc(f32x4 (&acc)[NQ], const float* __restrict__ A, int sa, const float* __restrict__ W, int ldw,
                                             int J, int kb, bool valid) {
    typedef float vq __attribute__((ext_vector_type(NQ == 1 ? 2 : NQ)));          // (NQ == 1 uses element 0 of a dummy pair)
    const int t = threadIdx.x & 63, lo = t & 15, hi = t >> 4;
    const float* ar = A + lo * sa + hi;
    const float* wr = W + (size_t)hi * ldw + kb + NQ * lo;
    for (int jc = 0; jc < J; jc += 4 * CHT) {
        float b[CHT][NQ];
#pragma unroll
        for (int s = 0; s < CHT; ++s) {
            const int j = jc + 4 * s;
            const float* src = wr + (size_t)((j < J && valid) ? j : 0) * ldw;
            if constexpr (NQ == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(valid ? src : W);
#pragma unroll
                for (int q = 0; q < 4; ++q) b[s][q] = v[q];
            } else if constexpr (NQ == 2) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 v = *reinterpret_cast<const f32x2*>(valid ? src : W);
                b[s][0] = v[0]; b[s][1] = v[1];
            } else {
                b[s][0] = *(valid ? src : W);
            }
        }
#pragma unroll
        for (int s = 0; s < CHT; ++s) {
            const int j = jc + 4 * s;
            if (j < J) {                                                          // wave-uniform
                const float a = ar[j];
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, valid ? b[s][q] : 0.f, acc[q], 0, 0, 0);
            }
        }
    }
}

template <int NQ, typename Epi>
__device__ __forceinline__ void gemm16_t(const float* __restrict__ A, int sa, const float* __restrict__ W, int ldw, int J, int Kout,
                                         float alpha, const float* __restrict__ A2, const float* __restrict__ W2, float alpha2, Epi epi) {
    const int t = threadIdx.x & 63, lo = t & 15, hi = t >> 4;
    const int wid = threadIdx.x >> 6;
    for (int kb = wid * 16 * NQ; kb < Kout; kb += NWAVES * 16 * NQ) {
        const bool valid = kb + NQ * lo + NQ - 1 < Kout;
        f32x4 acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm16_t_acc<NQ>(acc, A, sa, W, ldw, J, kb, valid);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][r] *= alpha;
        if (A2) {
            f32x4 acc2[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc2[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm16_t_acc<NQ>(acc2, A2, sa, W2, ldw, J, kb, valid);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[q][r] += acc2[q][r] * alpha2;
        }
        if (valid) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) epi(4 * hi + r, kb + NQ * lo + q, acc[q][r]);
        }
    }
}

__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// dst[r][c] (+)= rstd * (g - mean(g) - y * mean(g y)) over the 16 x C elements; g in LDS (stride SB), y in global (stride ys)
__device__ __forceinline__ void layer_norm16_bwd(float* dst, bool accumulate, const float* g, const float* __restrict__ y, int ys, int C,
                                                 float rstd, float* red) {
    const int n = T * C;
    float sg = 0.f, sgy = 0.f;
    for (int e = threadIdx.x; e < n; e += THREADS) {
        const int r = e / C, c = e % C;
        const float gv = g[r * SB + c];
        sg += gv;
        sgy += gv * y[r * ys + c];
    }
    const float mg = block_sum(sg, red) / (float)n;
    const float mgy = block_sum(sgy, red) / (float)n;
    for (int e = threadIdx.x; e < n; e += THREADS) {
        const int r = e / C, c = e % C;
        const float v = rstd * (g[r * SB + c] - mg - y[r * ys + c] * mgy);
        if (accumulate) dst[r * SB + c] += v;
        else dst[r * SB + c] = v;
    }
    __syncthreads();
}

__global__ __launch_bounds__(THREADS) void attn_stack_bwd_kernel(const BwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufG = smem;                    // gradient w.r.t. the current block's output, later w.r.t. x1
    float* bufA = bufG + T * SB;
    float* bufB = bufA + T * SB;
    float* bufQ = bufB + T * SB;
    float* bufK = bufQ + T * SS;
    float* bufV = bufK + T * SS;
    float* bufO = bufV + T * SS;
    float* ps = bufO + T * SS;             // [4][16][17] attention matrix
    float* red = ps + 4 * 16 * 17;         // [NWAVES]
    float* gs = bufA;                      // [4][16][17] gradient of the logits (bufA is free while the attention core runs)

    const int n = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, t = tid & 63;
    {
        const float* gg = p.gout + (size_t)n * T * CO;
        for (int e = tid; e < T * CO; e += THREADS) bufG[(e / CO) * SB + e % CO] = gg[e];
    }
    __syncthreads();
    for (int bi = p.nblocks - 1; bi >= 0; --bi) {
        const BlockW& w = p.blk[bi];
        const int C = w.cin, CP = w.cp;
        const size_t sn = (size_t)bi * p.N + n;
        const float* stats = p.s_stats + sn * 4;
        // ---- 0. g_x2 -> global (dW2 / db2)
        for (int e = tid; e < T * CO; e += THREADS) p.g_x2[sn * T * CO + e] = bufG[(e / CO) * SB + e % CO];
        // ---- 1. g_hpre = (alpha2 g_x2 W2) * gelu'(hpre) -> bufA
        {
            const float* hp = p.s_hpre + sn * T * CO;
            float* gh = p.g_hpre + sn * T * CO;
            gemm16_t<2>(bufG, SB, w.w2, CO, CO, CO, p.lr_mul * rsqrtf((float)CO), nullptr, nullptr, 0.f, [&](int r, int c, float v) {
                v *= gelu_grad(hp[r * CO + c]);
                bufA[r * SB + c] = v;
                gh[r * CO + c] = v;
            });
        }
        __syncthreads();
        // ---- 2. g_xn1 = alpha1 g_hpre W1 -> bufB
        gemm16_t<2>(bufA, SB, w.w1, CO, CO, CO, p.lr_mul * rsqrtf((float)CO), nullptr, nullptr, 0.f,
                 [&](int r, int c, float v) { bufB[r * SB + c] = v; });
        __syncthreads();
        // ---- 3. g_x1 = g_x2 + LN'(g_xn1) -> bufG ; -> global (dWproj, dWskip)
        layer_norm16_bwd(bufG, true, bufB, p.s_xn1 + sn * T * CO, CO, CO, stats[3], red);
        for (int e = tid; e < T * CO; e += THREADS) p.g_x1[sn * T * CO + e] = bufG[(e / CO) * SB + e % CO];
        // ---- 4. g_o = alpha_p g_x1 Wproj -> bufO ; q, k, v, sim of the forward -> LDS
        gemm16_t<1>(bufG, SB, w.wp, PL, CO, PL, p.lr_mul * rsqrtf((float)PL), nullptr, nullptr, 0.f,
                 [&](int r, int c, float v) { bufO[r * SS + c] = v; });
        for (int e = tid; e < T * PL; e += THREADS) {
            const int r = e / PL, c = e % PL;
            bufQ[r * SS + c] = p.s_q[sn * T * PL + e];
            bufK[r * SS + c] = p.s_k[sn * T * PL + e];
            bufV[r * SS + c] = p.s_v[sn * T * PL + e];
        }
        for (int e = tid; e < 4 * 256; e += THREADS) ps[(e >> 8) * 272 + ((e >> 4) & 15) * 17 + (e & 15)] = p.s_sim[sn * 1024 + e];
        __syncthreads();
        // ---- 5. attention core backward, wave g < 4 owns head g:
        //      gP = gO V^T ; gS = scale * P .* (gP - rowsum(gP .* P)) ; gQ = gS K ; gK = gS^T Q ; gV = P^T gO
        if (wave < 4) {
            const int g = wave, hc = g * 32;
            float* psg = ps + g * 272;
            float* gsg = gs + g * 272;
            {
                const int m = t >> 2, l0 = 4 * (t & 3);
                float gp[4], dot = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = 0.f;
                    for (int d = 0; d < 32; ++d) a += bufO[m * SS + hc + d] * bufV[(l0 + j) * SS + hc + d];
                    gp[j] = a;
                    dot += a * psg[m * 17 + l0 + j];
                }
                dot += __shfl_xor(dot, 1, 64);
                dot += __shfl_xor(dot, 2, 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) gsg[m * 17 + l0 + j] = psg[m * 17 + l0 + j] * (gp[j] - dot) * p.attn_scale;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            float rq[8], rk[8], rv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = t + 64 * i, r = e >> 5, d = e & 31;
                float aq = 0.f, ak = 0.f, av = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    aq += gsg[r * 17 + j] * bufK[j * SS + hc + d];
                    ak += gsg[j * 17 + r] * bufQ[j * SS + hc + d];
                    av += psg[j * 17 + r] * bufO[j * SS + hc + d];
                }
                rq[i] = aq; rk[i] = ak; rv[i] = av;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();              // every lane of this wave has read q / k / v / gO of the head
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = t + 64 * i, r = e >> 5, d = e & 31;
                bufQ[r * SS + hc + d] = rq[i];
                bufK[r * SS + hc + d] = rk[i];
                bufV[r * SS + hc + d] = rv[i];
                p.g_q[sn * T * PL + r * PL + hc + d] = rq[i];
                p.g_k[sn * T * PL + r * PL + hc + d] = rk[i];
                p.g_v[sn * T * PL + r * PL + hc + d] = rv[i];
            }
        }
        __syncthreads();
        // ---- 6. g_xn = alpha (g_k Wk + g_v Wv) -> bufB ;  g_P = alpha_q g_q Wq -> global (summed over the blocks that share P)
        {
            const float alpha = p.lr_mul * rsqrtf((float)C);
            gemm16_t<2>(bufK, SS, w.wk, C, PL, C, alpha, bufV, w.wv, alpha, [&](int r, int c, float v) { bufB[r * SB + c] = v; });
            float* gpp = (bi == 0 ? p.gp0 : p.gp) + (size_t)n * T * CP;
            const bool first = (bi == 0) || (bi == p.nblocks - 1);
            gemm16_t<2>(bufQ, SS, w.wq, CP, PL, CP, p.lr_mul * rsqrtf((float)CP), nullptr, nullptr, 0.f, [&](int r, int c, float v) {
                if (first) gpp[r * CP + c] = v;
                else gpp[r * CP + c] += v;
            });
        }
        __syncthreads();
        // ---- 7. gradient w.r.t. the block input: LN'(g_xn) + (g_x1 W0 alpha0 | g_x1) -> bufA, which becomes the next bufG
        layer_norm16_bwd(bufA, false, bufB, p.s_xn + sn * T * 528, 528, C, stats[1], red);
        if (C != CO) {
            gemm16_t<2>(bufG, SB, w.w0, C, CO, C, p.lr_mul * rsqrtf((float)C), nullptr, nullptr, 0.f,
                     [&](int r, int c, float v) { bufA[r * SB + c] += v; });
        } else {
            for (int e = tid; e < T * CO; e += THREADS) bufA[(e / CO) * SB + e % CO] += bufG[(e / CO) * SB + e % CO];
        }
        __syncthreads();
        { float* tmp = bufG; bufG = bufA; bufA = tmp; gs = bufA; }
        if (bi == 0) {
            float* gx = p.gx0 + (size_t)n * T * C;
            for (int e = tid; e < T * C; e += THREADS) gx[e] = bufG[(e / C) * SB + e % C];
        }
    }
}

}  // namespace

extern "C" int te_attn_stack_lds_bytes(void) {
    return (int)sizeof(float) * (3 * T * SB + 4 * T * SS + 4 * 16 * 17 + NWAVES);
}

// weights: host array of nblocks x 14 device pointers in the order wq, bq, wk, bk, wv, bv, wp, bp, w1, b1, w2, b2, w0, b0
// (w0 / b0 = the block's own `proj`, NULL when cin == 512); dims: host array of nblocks x 2 ints (cin, cp)
extern "C" int te_attn_stack_fwd_f32(float* xout, const float* x0, const float* p0, const float* p, const float* const* weights,
                                     const int* dims, int nblocks, int N, float lr_mul, float attn_scale, float eps,
                                     float* const* save, float* sim_out, te_stream_t stream_) {
    TE_REQUIRE(xout && x0 && p0 && weights && dims, TE_ERR_NULL, "te_attn_stack_fwd_f32: NULL pointer");
    TE_REQUIRE(nblocks > 0 && nblocks <= MAXBLK && N > 0, TE_ERR_SHAPE, "te_attn_stack_fwd_f32: 1..8 blocks, N > 0");
    TE_REQUIRE(nblocks == 1 || p, TE_ERR_NULL, "te_attn_stack_fwd_f32: p is NULL");
    StackArgs a{};
    a.xout = xout; a.x0 = x0; a.p0 = p0; a.p = p; a.N = N; a.nblocks = nblocks;
    a.lr_mul = lr_mul; a.attn_scale = attn_scale; a.eps = eps; a.sim_out = sim_out;
    if (save) {
        a.s_xn = save[0]; a.s_q = save[1]; a.s_k = save[2]; a.s_v = save[3]; a.s_o = save[4]; a.s_sim = save[5];
        a.s_x1 = save[6]; a.s_xn1 = save[7]; a.s_hpre = save[8]; a.s_h = save[9]; a.s_stats = save[10];
    }
    for (int b = 0; b < nblocks; ++b) {
        const float* const* w = weights + 14 * b;
        BlockW& k = a.blk[b];
        k.wq = w[0]; k.bq = w[1]; k.wk = w[2]; k.bk = w[3]; k.wv = w[4]; k.bv = w[5]; k.wp = w[6]; k.bp = w[7];
        k.w1 = w[8]; k.b1 = w[9]; k.w2 = w[10]; k.b2 = w[11]; k.w0 = w[12]; k.b0 = w[13];
        k.cin = dims[2 * b]; k.cp = dims[2 * b + 1];
        for (int i = 0; i < 12; ++i) TE_REQUIRE(w[i], TE_ERR_NULL, "te_attn_stack_fwd_f32: block %d parameter %d is NULL", b, i);
        TE_REQUIRE((k.cin == CO && !k.w0) || (k.cin != CO && k.w0 && k.b0), TE_ERR_SHAPE,
                   "te_attn_stack_fwd_f32: block %d: a skip projection exactly when cin != 512", b);
        TE_REQUIRE(k.cin % 16 == 0 && k.cp % 16 == 0 && k.cin <= 528 && k.cp <= 528 && k.cin >= 16 && k.cp >= 16, TE_ERR_UNSUPPORTED,
                   "te_attn_stack_fwd_f32: widths must be multiples of 16 up to 528 (got %d, %d)", k.cin, k.cp);
        TE_REQUIRE(b == 0 || k.cin == CO, TE_ERR_SHAPE, "te_attn_stack_fwd_f32: blocks after the first take the 512-wide output");
    }
    const int lds = te_attn_stack_lds_bytes();
    static std::atomic<uint64_t> done{0};
    te::allow_big_lds(done, (const void*)attn_stack_fwd_kernel, 160 * 1024);
    attn_stack_fwd_kernel<<<N, THREADS, lds, (hipStream_t)stream_>>>(a);
    return te::launch_status("te_attn_stack_fwd_f32");
}

// save: the 11 forward save buffers (order of te_attn_stack_fwd_f32); gmat: g_x2, g_hpre, g_x1, g_q, g_k, g_v
extern "C" int te_attn_stack_bwd_f32(float* gx0, float* gp0, float* gp, const float* gout, const float* const* weights, const int* dims,
                                     int nblocks, int N, float lr_mul, float attn_scale, float* const* save, float* const* gmat,
                                     te_stream_t stream_) {
    TE_REQUIRE(gx0 && gp0 && gout && weights && dims && save && gmat, TE_ERR_NULL, "te_attn_stack_bwd_f32: NULL pointer");
    TE_REQUIRE(nblocks > 0 && nblocks <= MAXBLK && N > 0, TE_ERR_SHAPE, "te_attn_stack_bwd_f32: 1..8 blocks, N > 0");
    TE_REQUIRE(nblocks == 1 || gp, TE_ERR_NULL, "te_attn_stack_bwd_f32: gp is NULL");
    BwdArgs a{};
    a.gout = gout; a.gx0 = gx0; a.gp0 = gp0; a.gp = gp; a.N = N; a.nblocks = nblocks; a.lr_mul = lr_mul; a.attn_scale = attn_scale;
    a.s_xn = save[0]; a.s_q = save[1]; a.s_k = save[2]; a.s_v = save[3]; a.s_sim = save[5]; a.s_xn1 = save[7]; a.s_hpre = save[8];
    a.s_stats = save[10];
    a.g_x2 = gmat[0]; a.g_hpre = gmat[1]; a.g_x1 = gmat[2]; a.g_q = gmat[3]; a.g_k = gmat[4]; a.g_v = gmat[5];
    for (int i = 0; i < 6; ++i) TE_REQUIRE(gmat[i], TE_ERR_NULL, "te_attn_stack_bwd_f32: gmat[%d] is NULL", i);
    TE_REQUIRE(a.s_xn && a.s_q && a.s_k && a.s_v && a.s_sim && a.s_xn1 && a.s_hpre && a.s_stats, TE_ERR_NULL,
               "te_attn_stack_bwd_f32: a forward save buffer is NULL");
    for (int b = 0; b < nblocks; ++b) {
        const float* const* w = weights + 14 * b;
        BlockW& k = a.blk[b];
        k.wq = w[0]; k.bq = w[1]; k.wk = w[2]; k.bk = w[3]; k.wv = w[4]; k.bv = w[5]; k.wp = w[6]; k.bp = w[7];
        k.w1 = w[8]; k.b1 = w[9]; k.w2 = w[10]; k.b2 = w[11]; k.w0 = w[12]; k.b0 = w[13];
        k.cin = dims[2 * b]; k.cp = dims[2 * b + 1];
        TE_REQUIRE(k.wq && k.wk && k.wv && k.wp && k.w1 && k.w2, TE_ERR_NULL, "te_attn_stack_bwd_f32: block %d weight is NULL", b);
        TE_REQUIRE((k.cin == CO) == (k.w0 == nullptr), TE_ERR_SHAPE, "te_attn_stack_bwd_f32: block %d skip projection mismatch", b);
        TE_REQUIRE(k.cin % 16 == 0 && k.cp % 16 == 0 && k.cin <= 528 && k.cp <= 528 && (b == 0 || k.cin == CO), TE_ERR_UNSUPPORTED,
                   "te_attn_stack_bwd_f32: unsupported widths (%d, %d)", k.cin, k.cp);
    }
    const int lds = te_attn_stack_lds_bytes();
    static std::atomic<uint64_t> done{0};
    te::allow_big_lds(done, (const void*)attn_stack_bwd_kernel, 160 * 1024);
    attn_stack_bwd_kernel<<<N, THREADS, lds, (hipStream_t)stream_>>>(a);
    return te::launch_status("te_attn_stack_bwd_f32");
}
