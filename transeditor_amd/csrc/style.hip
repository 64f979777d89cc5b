// M1: demodulation coefficients of the modulated convolution (reference: ModulatedConv2d.forward,
// model_spatial_query.py:300-304 — weight = scale * W * style; demod = rsqrt(sum_{ci,k} weight^2 + 1e-8), taken there on
// B materialised weight copies).  With the shared-weight formulation only the [B,Co] coefficient is needed:
//
//     wsq[co,ci] = wscale^2 * sum_t w[co,ci,t]^2
//     d[b,co]    = rsqrt( sum_ci s[b,ci]^2 * wsq[co,ci] + eps )
//
// and its gradient, with u[b,co] = -0.5 * gd[b,co] * d[b,co]^3:
//
//     gw[co,ci,t] = 2 * wscale^2 * w[co,ci,t] * sum_b u[b,co] * s[b,ci]^2
//     gs[b,ci]    = 2 * s[b,ci] * sum_co u[b,co] * wsq[co,ci]
//
// Tiny, launch-latency-bound problems (<= 2.4 M weights): the point of these kernels is replacing ~15 framework
// launches per layer (pow, sum, mm, add, rsqrt and their backward) by three.
#include "te_common.h"

namespace {

constexpr int BMAX = 64;      // samples per launch held in LDS by the backward kernels

__global__ __launch_bounds__(256) void demod_fwd_kernel(float* __restrict__ d, float* __restrict__ wsq,
                                                        const float* __restrict__ w, const float* __restrict__ s,
                                                        float wscale, float eps, int B, int Co, int Ci, int T) {
    extern __shared__ float wq[];                         // [Ci]
    const int co = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* wr = w + (size_t)co * Ci * T;
    const float ws2 = wscale * wscale;
    if (T == 0) {                                         // w already holds wsq[Co, Ci] (cached for frozen weights)
        for (int ci = tid; ci < Ci; ci += 256) wq[ci] = w[(size_t)co * Ci + ci];
    } else {
        for (int ci = tid; ci < Ci; ci += 256) {
            float a = 0.f;
            for (int t = 0; t < T; ++t) {
                const float v = wr[(size_t)ci * T + t];
                a += v * v;
            }
            a *= ws2;
            wq[ci] = a;
            wsq[(size_t)co * Ci + ci] = a;
        }
    }
    __syncthreads();
    for (int b = wid; b < B; b += 4) {                    // one wave per sample: no block barrier in the loop
        const float* sr = s + (size_t)b * Ci;
        float a = 0.f;
        for (int ci = lane; ci < Ci; ci += 64) {
            const float v = sr[ci];
            a += v * v * wq[ci];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) d[(size_t)b * Co + co] = rsqrtf(a + eps);
    }
}

// block per co: gw rows
__global__ __launch_bounds__(256) void demod_bwd_w_kernel(float* __restrict__ gw, const float* __restrict__ gd,
                                                          const float* __restrict__ d, const float* __restrict__ w,
                                                          const float* __restrict__ s, float wscale, int B, int Co, int Ci,
                                                          int T, int acc) {
    __shared__ float u[BMAX];
    const int co = blockIdx.x, tid = threadIdx.x;
    if (tid < B) {
        const float dv = d[(size_t)tid * Co + co];
        u[tid] = -0.5f * gd[(size_t)tid * Co + co] * dv * dv * dv;
    }
    __syncthreads();
    const float c2 = 2.f * wscale * wscale;
    for (int ci = tid; ci < Ci; ci += 256) {
        float q = 0.f;
        for (int b = 0; b < B; ++b) {
            const float v = s[(size_t)b * Ci + ci];
            q += u[b] * v * v;
        }
        q *= c2;
        const size_t o = ((size_t)co * Ci + ci) * T;
        for (int t = 0; t < T; ++t) gw[o + t] = (acc ? gw[o + t] : 0.f) + w[o + t] * q;
    }
}

// grid (ceil(Ci/64), B), 256 threads: lane -> ci, the 4 waves split Co and combine through LDS
__global__ __launch_bounds__(256) void demod_bwd_s_kernel(float* __restrict__ gs, const float* __restrict__ gd,
                                                          const float* __restrict__ d, const float* __restrict__ wsq,
                                                          const float* __restrict__ s, int B, int Co, int Ci, int acc) {
    extern __shared__ float uu[];                         // [Co] + [4][64]
    float* part = uu + Co;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ci = min(blockIdx.x * 64 + lane, Ci - 1);
    for (int co = tid; co < Co; co += 256) {
        const float dv = d[(size_t)b * Co + co];
        uu[co] = -0.5f * gd[(size_t)b * Co + co] * dv * dv * dv;
    }
    __syncthreads();
    const int cq = (Co + 3) / 4, c0 = wid * cq, c1 = min(Co, c0 + cq);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int co = c0;
    for (; co + 3 < c1; co += 4) {
        a0 += uu[co] * wsq[(size_t)co * Ci + ci];
        a1 += uu[co + 1] * wsq[(size_t)(co + 1) * Ci + ci];
        a2 += uu[co + 2] * wsq[(size_t)(co + 2) * Ci + ci];
        a3 += uu[co + 3] * wsq[(size_t)(co + 3) * Ci + ci];
    }
    for (; co < c1; ++co) a0 += uu[co] * wsq[(size_t)co * Ci + ci];
    part[wid * 64 + lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wid == 0 && blockIdx.x * 64 + lane < Ci) {
        const size_t o = (size_t)b * Ci + ci;
        gs[o] = (acc ? gs[o] : 0.f) + 2.f * s[o] * ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]));
    }
}

}  // namespace

extern "C" int te_demod_fwd_f32(float* d, float* wsq, const float* w, const float* s, float wscale, float eps, int B, int Co,
                                int Ci, int T, te_stream_t stream_) {
    TE_REQUIRE(d && w && s && (wsq || T == 0), TE_ERR_NULL, "te_demod_fwd_f32: NULL pointer");
    TE_REQUIRE(B > 0 && Co > 0 && Ci > 0 && T >= 0, TE_ERR_SHAPE, "te_demod_fwd_f32: bad dims");
    TE_REQUIRE(Ci <= 8192, TE_ERR_UNSUPPORTED, "te_demod_fwd_f32: Ci <= 8192");
    demod_fwd_kernel<<<Co, 256, sizeof(float) * Ci, (hipStream_t)stream_>>>(d, wsq, w, s, wscale, eps, B, Co, Ci, T);
    return te::launch_status("te_demod_fwd_f32");
}

extern "C" int te_demod_bwd_f32(float* gw, float* gs, const float* gd, const float* d, const float* w, const float* wsq,
                                const float* s, float wscale, int B, int Co, int Ci, int T, int accumulate, te_stream_t stream_) {
    TE_REQUIRE(gd && d && w && wsq && s, TE_ERR_NULL, "te_demod_bwd_f32: NULL pointer");
    TE_REQUIRE(B > 0 && Co > 0 && Ci > 0 && T > 0, TE_ERR_SHAPE, "te_demod_bwd_f32: bad dims");
    TE_REQUIRE(B <= BMAX && Co <= 8192, TE_ERR_UNSUPPORTED, "te_demod_bwd_f32: B <= 64, Co <= 8192");
    hipStream_t st = (hipStream_t)stream_;
    if (gw) demod_bwd_w_kernel<<<Co, 256, 0, st>>>(gw, gd, d, w, s, wscale, B, Co, Ci, T, accumulate);
    if (gs) {
        dim3 grid((unsigned)te::cdiv(Ci, 64), (unsigned)B);
        demod_bwd_s_kernel<<<grid, 256, sizeof(float) * (Co + 256), st>>>(gs, gd, d, wsq, s, B, Co, Ci, accumulate);
    }
    return te::launch_status("te_demod_bwd_f32");
}
