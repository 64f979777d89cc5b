// D1: minibatch standard deviation of the discriminator as ONE launch, concat included.
//
// Reference: Discriminator.forward, model_spatial_query.py:844-852 —
//     group = min(batch, 4); stddev = out.view(group, -1, 1, C, H, W)
//     stddev = sqrt(stddev.var(0, unbiased=False) + 1e-8).mean([2, 3, 4], keepdims=True).squeeze(2)
//     out = cat([out, stddev.repeat(group, 1, H, W)], 1)
// (view / var / add / sqrt / mean / repeat / cat = 8 launches forward, ~20 backward).  With n = B / group, sample
// b = g * n + j belongs to set j; per set:  s_j = mean_p sqrt(var_g x[g*n + j, p] + eps), p over the C*H*W positions.
//   forward : y[b, :C] = x[b], y[b, C, :, :] = s_{b mod n}                                   (one block per set j)
//   backward: gx[g*n+j, p] = gy[g*n+j, p] + G_j (x[g*n+j, p] - mu_p) / (P * group * sqrt(var_p + eps)),
//             G_j = sum over the set's group samples and H*W positions of gy[., C, .]
// Shapes are tiny ([B, 512, 4, 4]): latency-bound, one 1024-thread block per set.
#include "te_common.h"

namespace {

constexpr int THREADS = 1024;
constexpr int MAXG = 4;

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < THREADS / 64; ++w) t += red[w];
    return t;
}

__global__ __launch_bounds__(THREADS) void stddev_fwd_kernel(float* __restrict__ y, const float* __restrict__ x, int n, int group,
                                                             int C, int HW, float eps) {
    __shared__ float red[THREADS / 64];
    const int j = blockIdx.x;
    const int P = C * HW;
    const float inv_g = 1.f / group;
    float acc = 0.f;
    for (int p = threadIdx.x; p < P; p += THREADS) {
        float v[MAXG], mu = 0.f;
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
            if (g < group) { v[g] = x[(size_t)(g * n + j) * P + p]; mu += v[g]; }
        mu *= inv_g;
        float var = 0.f;
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
            if (g < group) { const float d = v[g] - mu; var += d * d; y[(size_t)(g * n + j) * (P + HW) + p] = v[g]; }
        acc += sqrtf(var * inv_g + eps);
    }
    const float s = block_sum(acc, red) / P;
    for (int e = threadIdx.x; e < group * HW; e += THREADS) {
        const int g = e / HW, q = e - g * HW;
        y[(size_t)(g * n + j) * (P + HW) + P + q] = s;
    }
}

__global__ __launch_bounds__(THREADS) void stddev_bwd_kernel(float* __restrict__ gx, const float* __restrict__ gy,
                                                             const float* __restrict__ x, int n, int group, int C, int HW, float eps) {
    __shared__ float red[THREADS / 64];
    const int j = blockIdx.x;
    const int P = C * HW;
    float gs = 0.f;
    for (int e = threadIdx.x; e < group * HW; e += THREADS) {
        const int g = e / HW, q = e - g * HW;
        gs += gy[(size_t)(g * n + j) * (P + HW) + P + q];
    }
    const float G = block_sum(gs, red);
    const float inv_g = 1.f / group;
    const float k = G / ((float)P * group);
    for (int p = threadIdx.x; p < P; p += THREADS) {
        float v[MAXG], mu = 0.f;
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
            if (g < group) { v[g] = x[(size_t)(g * n + j) * P + p]; mu += v[g]; }
        mu *= inv_g;
        float var = 0.f;
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
            if (g < group) { const float d = v[g] - mu; var += d * d; }
        const float r = k / sqrtf(var * inv_g + eps);
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
            if (g < group) gx[(size_t)(g * n + j) * P + p] = gy[(size_t)(g * n + j) * (P + HW) + p] + r * (v[g] - mu);
    }
}

}  // namespace

extern "C" int te_minibatch_stddev_fwd_f32(float* y, const float* x, int B, int group, int C, int HW, float eps, te_stream_t stream_) {
    TE_REQUIRE(y && x, TE_ERR_NULL, "te_minibatch_stddev_fwd_f32: NULL pointer");
    TE_REQUIRE(B > 0 && C > 0 && HW > 0 && group > 0 && group <= MAXG && B % group == 0, TE_ERR_SHAPE,
               "te_minibatch_stddev_fwd_f32: need 0 < group <= 4 dividing B");
    stddev_fwd_kernel<<<B / group, THREADS, 0, (hipStream_t)stream_>>>(y, x, B / group, group, C, HW, eps);
    return te::launch_status("te_minibatch_stddev_fwd_f32");
}

extern "C" int te_minibatch_stddev_bwd_f32(float* gx, const float* gy, const float* x, int B, int group, int C, int HW, float eps,
                                           te_stream_t stream_) {
    TE_REQUIRE(gx && gy && x, TE_ERR_NULL, "te_minibatch_stddev_bwd_f32: NULL pointer");
    TE_REQUIRE(B > 0 && C > 0 && HW > 0 && group > 0 && group <= MAXG && B % group == 0, TE_ERR_SHAPE,
               "te_minibatch_stddev_bwd_f32: need 0 < group <= 4 dividing B");
    stddev_bwd_kernel<<<B / group, THREADS, 0, (hipStream_t)stream_>>>(gx, gy, x, B / group, group, C, HW, eps);
    return te::launch_status("te_minibatch_stddev_bwd_f32");
}
