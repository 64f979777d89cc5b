// A2: the parameter-free layer norm of the attention blocks (reference: AttentionBlock.forward, model_spatial_query.py:
// 924 / 931 — F.layer_norm(x, x.size()[1:]): every sample is normalised over ALL of its 16 x 512 (16 x 528 in block 0)
// token-channel elements, eps 1e-5, no affine).  Only B = 16 rows of 8192 elements: the framework kernel runs one block
// per row (7 us forward, 14 us backward); here a row is one 1024-thread block with 16-byte loads kept in registers
// between the reduction and the normalisation pass.
//   forward : mean, rstd = 1/sqrt(var + eps) (two-pass variance), y = (x - mean) * rstd
//   backward: gx = rstd * (g - mean(g) - y * mean(g * y))
#include "te_common.h"

namespace {

constexpr int LN_THREADS = 1024;
constexpr int LN_MAXV = 4;            // float4 per thread held in registers: rows up to 16384 elements

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                                   // red may still be read from the previous reduction
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < LN_THREADS / 64; ++w) t += red[w];
    return t;
}

__global__ __launch_bounds__(LN_THREADS) void layer_norm_fwd_kernel(float* __restrict__ y, float* __restrict__ stats,
                                                                    const float* __restrict__ x, int N, float eps) {
    __shared__ float red[LN_THREADS / 64];
    const int row = blockIdx.x, tid = threadIdx.x, N4 = N >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * N);
    f32x4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int e = tid + LN_THREADS * i;
        v[i] = e < N4 ? xr[e] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = block_sum(s, red) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int e = tid + LN_THREADS * i;
        if (e < N4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float d = v[i][c] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(block_sum(q, red) / (float)N + eps);
    f32x4* yr = reinterpret_cast<f32x4*>(y + (size_t)row * N);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int e = tid + LN_THREADS * i;
        if (e < N4) yr[e] = (v[i] - mean) * rstd;
    }
    if (tid == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

__global__ __launch_bounds__(LN_THREADS) void layer_norm_bwd_kernel(float* __restrict__ gx, const float* __restrict__ g,
                                                                    const float* __restrict__ y,
                                                                    const float* __restrict__ stats, int N) {
    __shared__ float red[LN_THREADS / 64];
    const int row = blockIdx.x, tid = threadIdx.x, N4 = N >> 2;
    const f32x4* gr = reinterpret_cast<const f32x4*>(g + (size_t)row * N);
    const f32x4* yr = reinterpret_cast<const f32x4*>(y + (size_t)row * N);
    f32x4 gv[LN_MAXV], yv[LN_MAXV];
    float sg = 0.f, sgy = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int e = tid + LN_THREADS * i;
        gv[i] = e < N4 ? gr[e] : f32x4{0.f, 0.f, 0.f, 0.f};
        yv[i] = e < N4 ? yr[e] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) { sg += gv[i][c]; sgy += gv[i][c] * yv[i][c]; }
    }
    const float mg = block_sum(sg, red) / (float)N;
    const float mgy = block_sum(sgy, red) / (float)N;
    const float rstd = stats[2 * row + 1];
    f32x4* or_ = reinterpret_cast<f32x4*>(gx + (size_t)row * N);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int e = tid + LN_THREADS * i;
        if (e < N4) or_[e] = (gv[i] - mg - yv[i] * mgy) * rstd;
    }
}

inline bool ln_ok(int64_t R, int N) { return R > 0 && N > 0 && (N & 3) == 0 && N <= 4 * LN_THREADS * LN_MAXV; }

}  // namespace

extern "C" int te_layer_norm_supported(int64_t R, int N) { return ln_ok(R, N) ? 1 : 0; }

extern "C" int te_layer_norm_fwd_f32(float* y, float* stats, const float* x, int64_t R, int N, float eps, te_stream_t stream_) {
    TE_REQUIRE(y && stats && x, TE_ERR_NULL, "te_layer_norm_fwd_f32: NULL pointer");
    TE_REQUIRE(ln_ok(R, N) && R <= 0x7FFFFFFF, TE_ERR_UNSUPPORTED, "te_layer_norm_fwd_f32: need N %% 4 == 0 and N <= 16384");
    TE_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, TE_ERR_UNSUPPORTED,
               "te_layer_norm_fwd_f32: 16-byte aligned rows required");
    layer_norm_fwd_kernel<<<(unsigned)R, LN_THREADS, 0, (hipStream_t)stream_>>>(y, stats, x, N, eps);
    return te::launch_status("te_layer_norm_fwd_f32");
}

extern "C" int te_layer_norm_bwd_f32(float* gx, const float* g, const float* y, const float* stats, int64_t R, int N,
                                     te_stream_t stream_) {
    TE_REQUIRE(gx && g && y && stats, TE_ERR_NULL, "te_layer_norm_bwd_f32: NULL pointer");
    TE_REQUIRE(ln_ok(R, N) && R <= 0x7FFFFFFF, TE_ERR_UNSUPPORTED, "te_layer_norm_bwd_f32: need N %% 4 == 0 and N <= 16384");
    TE_REQUIRE(((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gx)) & 15) == 0,
               TE_ERR_UNSUPPORTED, "te_layer_norm_bwd_f32: 16-byte aligned rows required");
    layer_norm_bwd_kernel<<<(unsigned)R, LN_THREADS, 0, (hipStream_t)stream_>>>(gx, g, y, stats, N);
    return te::launch_status("te_layer_norm_bwd_f32");
}

// G1: PixelNorm over the channel axis of the [B, D, C] latent codes (reference: PixelNorm.forward,
// model_spatial_query.py:80-81 with pixel_norm_op_dim = 1):  y[b,d,c] = x[b,d,c] * rsqrt(mean_d x[b,d,c]^2 + 1e-8).
// One block per sample; thread (group = tid / C, c = tid % C) strides over d, C must divide 256.
//   backward: gx = r * (g - y * mean_d(g * y)),  r[b,c] saved by the forward
namespace {

__global__ __launch_bounds__(256) void pixel_norm_fwd_kernel(float* __restrict__ y, float* __restrict__ r,
                                                             const float* __restrict__ x, int D, int C, float eps) {
    __shared__ float part[256];
    __shared__ float rs[256];
    const int b = blockIdx.x, tid = threadIdx.x, c = tid % C, grp = tid / C, ng = 256 / C;
    const float* xb = x + (size_t)b * D * C;
    float s = 0.f;
    for (int d = grp; d < D; d += ng) { const float v = xb[d * C + c]; s += v * v; }
    part[tid] = s;
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
        for (int gI = 0; gI < ng; ++gI) t += part[gI * C + tid];
        const float rv = rsqrtf(t / (float)D + eps);
        rs[tid] = rv;
        r[(size_t)b * C + tid] = rv;
    }
    __syncthreads();
    const float rv = rs[c];
    float* yb = y + (size_t)b * D * C;
    for (int d = grp; d < D; d += ng) yb[d * C + c] = xb[d * C + c] * rv;
}

__global__ __launch_bounds__(256) void pixel_norm_bwd_kernel(float* __restrict__ gx, const float* __restrict__ g,
                                                             const float* __restrict__ y, const float* __restrict__ r, int D,
                                                             int C) {
    __shared__ float part[256];
    __shared__ float ms[256];
    const int b = blockIdx.x, tid = threadIdx.x, c = tid % C, grp = tid / C, ng = 256 / C;
    const float* gb = g + (size_t)b * D * C;
    const float* yb = y + (size_t)b * D * C;
    float s = 0.f;
    for (int d = grp; d < D; d += ng) s += gb[d * C + c] * yb[d * C + c];
    part[tid] = s;
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
        for (int gI = 0; gI < ng; ++gI) t += part[gI * C + tid];
        ms[tid] = t / (float)D;
    }
    __syncthreads();
    const float m = ms[c], rv = r[(size_t)b * C + c];
    float* ob = gx + (size_t)b * D * C;
    for (int d = grp; d < D; d += ng) ob[d * C + c] = rv * (gb[d * C + c] - yb[d * C + c] * m);
}

inline bool pn_ok(int64_t B, int D, int C) { return B > 0 && B <= 0x7FFFFFFF && D > 0 && C > 0 && C <= 256 && 256 % C == 0; }

}  // namespace

extern "C" int te_pixel_norm_supported(int64_t B, int D, int C) { return pn_ok(B, D, C) ? 1 : 0; }

extern "C" int te_pixel_norm_fwd_f32(float* y, float* r, const float* x, int64_t B, int D, int C, float eps, te_stream_t stream_) {
    TE_REQUIRE(y && r && x, TE_ERR_NULL, "te_pixel_norm_fwd_f32: NULL pointer");
    TE_REQUIRE(pn_ok(B, D, C), TE_ERR_UNSUPPORTED, "te_pixel_norm_fwd_f32: C must divide 256");
    pixel_norm_fwd_kernel<<<(unsigned)B, 256, 0, (hipStream_t)stream_>>>(y, r, x, D, C, eps);
    return te::launch_status("te_pixel_norm_fwd_f32");
}

extern "C" int te_pixel_norm_bwd_f32(float* gx, const float* g, const float* y, const float* r, int64_t B, int D, int C,
                                     te_stream_t stream_) {
    TE_REQUIRE(gx && g && y && r, TE_ERR_NULL, "te_pixel_norm_bwd_f32: NULL pointer");
    TE_REQUIRE(pn_ok(B, D, C), TE_ERR_UNSUPPORTED, "te_pixel_norm_bwd_f32: C must divide 256");
    pixel_norm_bwd_kernel<<<(unsigned)B, 256, 0, (hipStream_t)stream_>>>(gx, g, y, r, D, C);
    return te::launch_status("te_pixel_norm_bwd_f32");
}
