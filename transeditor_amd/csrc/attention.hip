// F2: attention core of the dual-space cross-attention block (Q from the P space, K/V from the Z space).
// Reference: Attention.forward, model_spatial_query.py:888-894 — per sample n and head g (4 heads x 32 dims,
// 16 query tokens x 16 key tokens):   sim = softmax_l( scale * q k^T ),   o = sim v.
//
// One wave64 per (sample, head).  Both contractions run on v_mfma_f32_16x16x4_f32 (exact fp32):
//   S = Q K^T : 16x16, K-dim 32  -> 8 MFMAs;   A[i=m][k=d] = q, B[k=d][j=l] = k
//   O = P V   : 16x32, K-dim 16  -> 2 x 4 MFMAs;  A[i=m][k=l] = P (re-laid out through LDS), B[k=l][j=d] = v
// Operand layout (16x16x4): lane t supplies A[t&15][t>>4] and B[t>>4][t&15]; result reg r of lane t is
// C[4*(t>>4)+r][t&15].  Softmax runs in registers: a row of S lives in the 16 lanes t&15 of one quarter-wave,
// so max / sum are 4 xor-shuffles.  Latency-bound by construction (0.2 % of the generator FLOPs).
#include "te_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float quarter_max(float v) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float quarter_sum(float v) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// M = L = 16, D = 32
__global__ __launch_bounds__(64) void attn_fwd_kernel(float* __restrict__ o, float* __restrict__ sim,
                                                      const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, float scale, int G) {
    __shared__ float ps[16][17];
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int t = threadIdx.x, lo = t & 15, hi = t >> 4;
    const int C = G * 32;
    const float* qn = q + (size_t)n * 16 * C + g * 32;
    const float* kn = k + (size_t)n * 16 * C + g * 32;
    const float* vn = v + (size_t)n * 16 * C + g * 32;

    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const float a = qn[lo * C + 4 * st + hi];     // A[m = lo][d = 4 st + hi]
        const float b = kn[lo * C + 4 * st + hi];     // B[d = 4 st + hi][l = lo] = k[l][d]
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, s, 0, 0, 0);
    }
    // lane holds S[m = 4 hi + r][l = lo]
    float pr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float x = s[r] * scale;
        const float mx = quarter_max(x);
        const float e = expf(x - mx);
        const float sm = quarter_sum(e);
        pr[r] = e / sm;
        ps[4 * hi + r][lo] = pr[r];
        sim[(((size_t)n * G + g) * 16 + 4 * hi + r) * 16 + lo] = pr[r];
    }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const float a = ps[lo][4 * st + hi];                       // A[m = lo][l = 4 st + hi]
            const float b = vn[(4 * st + hi) * C + nb * 16 + lo];      // B[l][d = nb*16 + lo]
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) o[((size_t)n * 16 + 4 * hi + r) * C + g * 32 + nb * 16 + lo] = acc[r];
    }
}

// Backward (first order).  Small dense loops through LDS; one wave per (sample, head).
//   gV = P^T gO ; gP = gO V^T (+ gsim) ; gS = P .* (gP - rowsum(gP .* P)) ; gQ = scale gS K ; gK = scale gS^T Q
__global__ __launch_bounds__(64) void attn_bwd_kernel(float* __restrict__ gq, float* __restrict__ gk, float* __restrict__ gv,
                                                      const float* __restrict__ go, const float* __restrict__ gsim,
                                                      const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, const float* __restrict__ sim,
                                                      float scale, int G) {
    __shared__ float sq[16][33], sk[16][33], sv[16][33], sgo[16][33], sp[16][17], sgs[16][17];
    const int n = blockIdx.x / G, g = blockIdx.x % G, t = threadIdx.x;
    const int C = G * 32;
    const size_t base = (size_t)n * 16 * C + g * 32;
    for (int e = t; e < 16 * 32; e += 64) {
        const int r = e >> 5, c = e & 31;
        sq[r][c] = q[base + r * C + c];
        sk[r][c] = k[base + r * C + c];
        sv[r][c] = v[base + r * C + c];
        sgo[r][c] = go[base + r * C + c];
    }
    const size_t sbase = ((size_t)n * G + g) * 256;
    for (int e = t; e < 256; e += 64) sp[e >> 4][e & 15] = sim[sbase + e];
    __syncthreads();
    // gP and gS: thread t -> row m = t >> 2, 4 columns l = 4 (t & 3) .. +3
    {
        const int m = t >> 2, l0 = 4 * (t & 3);
        float gp[4], dot = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = gsim ? gsim[sbase + m * 16 + l0 + j] : 0.f;
            for (int d = 0; d < 32; ++d) a += sgo[m][d] * sv[l0 + j][d];
            gp[j] = a;
            dot += a * sp[m][l0 + j];
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
#pragma unroll
        for (int j = 0; j < 4; ++j) sgs[m][l0 + j] = sp[m][l0 + j] * (gp[j] - dot) * scale;
    }
    __syncthreads();
    // gQ[m][d], gK[l][d], gV[l][d]: 512 outputs each, 8 per thread
    for (int e = t; e < 512; e += 64) {
        const int r = e >> 5, d = e & 31;
        float aq = 0.f, ak = 0.f, av = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            aq += sgs[r][j] * sk[j][d];
            ak += sgs[j][r] * sq[j][d];
            av += sp[j][r] * sgo[j][d];
        }
        gq[base + r * C + d] = aq;
        gk[base + r * C + d] = ak;
        gv[base + r * C + d] = av;
    }
}

}  // namespace

extern "C" int te_attn_fwd_f32(float* o, float* sim, const float* q, const float* k, const float* v, float scale, int N,
                               int G, int M, int L, int D, te_stream_t stream_) {
    TE_REQUIRE(o && sim && q && k && v, TE_ERR_NULL, "te_attn_fwd_f32: NULL pointer");
    TE_REQUIRE(N > 0 && G > 0, TE_ERR_SHAPE, "te_attn_fwd_f32: bad dims");
    TE_REQUIRE(M == 16 && L == 16 && D == 32, TE_ERR_UNSUPPORTED, "te_attn_fwd_f32: only M=L=16, D=32 (got %d,%d,%d)", M, L, D);
    attn_fwd_kernel<<<N * G, 64, 0, (hipStream_t)stream_>>>(o, sim, q, k, v, scale, G);
    return te::launch_status("te_attn_fwd_f32");
}

extern "C" int te_attn_bwd_f32(float* gq, float* gk, float* gv, const float* go, const float* gsim, const float* q,
                               const float* k, const float* v, const float* sim, float scale, int N, int G, int M, int L,
                               int D, te_stream_t stream_) {
    TE_REQUIRE(gq && gk && gv && go && q && k && v && sim, TE_ERR_NULL, "te_attn_bwd_f32: NULL pointer");
    TE_REQUIRE(N > 0 && G > 0, TE_ERR_SHAPE, "te_attn_bwd_f32: bad dims");
    TE_REQUIRE(M == 16 && L == 16 && D == 32, TE_ERR_UNSUPPORTED, "te_attn_bwd_f32: only M=L=16, D=32 (got %d,%d,%d)", M, L, D);
    attn_bwd_kernel<<<N * G, 64, 0, (hipStream_t)stream_>>>(gq, gk, gv, go, gsim, q, k, v, sim, scale, G);
    return te::launch_status("te_attn_bwd_f32");
}
