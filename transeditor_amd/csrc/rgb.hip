// M3: ToRGB — 1x1 modulated convolution to 3 channels (reference: ToRGB.forward, model_spatial_query.py:416-425,
// a grouped 1x1 F.conv2d with 3 output channels per sample).  ~1.5 FLOP/byte: HBM-bound, so these kernels are pure
// streaming (16 B per lane, every activation element touched once) instead of a 3-of-128-rows MFMA tile.
//
//   fwd   : out[b,o,p] = sum_k (w[o,k] * s[b,k]) * x[b,k,p] + bias[o]
//   dgrad : gx[b,k,p]  = s[b,k] * sum_o w[o,k] * g[b,o,p]
//   wgrad : slab[b][c][o][k] = sum_{p in chunk c} g[b,o,p] * x[b,k,p]      (unmodulated; te_wgrad_reduce_f32 finishes)
#include "te_common.h"

namespace {

constexpr int KMAX = 512;
constexpr int NOUT = 3;

__global__ __launch_bounds__(256) void rgb_fwd_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                      const float* __restrict__ w, const float* __restrict__ isc,
                                                      const float* __restrict__ bias, float wscale, int K, int HW) {
    __shared__ float ws[NOUT][KMAX];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int k = tid; k < K; k += 256) {
        const float s = (isc ? isc[(size_t)b * K + k] : 1.f) * wscale;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) ws[o][k] = w[o * K + k] * s;
    }
    __syncthreads();
    const int HW4 = HW >> 2;
    const int p4 = blockIdx.x * 256 + tid;
    if (p4 >= HW4) return;
    const float4* xp = reinterpret_cast<const float4*>(x + (size_t)b * K * HW) + p4;
    float4 acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int k = 0; k < K; ++k) {
        const float4 v = xp[(size_t)k * HW4];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float c = ws[o][k];
            acc[o].x += c * v.x; acc[o].y += c * v.y; acc[o].z += c * v.z; acc[o].w += c * v.w;
        }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const float bb = bias ? bias[o] : 0.f;
        acc[o].x += bb; acc[o].y += bb; acc[o].z += bb; acc[o].w += bb;
        reinterpret_cast<float4*>(out + ((size_t)b * NOUT + o) * HW)[p4] = acc[o];
    }
}

// (also the forward of the discriminator's from-RGB stem, a 1x1 convolution FROM 3 channels: `bias` [K] and the scaled
// leaky-ReLU `act` (0 none, 3 gain sqrt(2), 4 gain 1) ride in the epilogue there)
__global__ __launch_bounds__(256) void rgb_dgrad_kernel(float* __restrict__ gx, const float* __restrict__ g,
                                                        const float* __restrict__ w, const float* __restrict__ isc,
                                                        float wscale, int K, int HW, const float* __restrict__ bias = nullptr,
                                                        int act = 0) {
    __shared__ float ws[NOUT][KMAX];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int k = tid; k < K; k += 256) {
        const float s = (isc ? isc[(size_t)b * K + k] : 1.f) * wscale;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) ws[o][k] = w[o * K + k] * s;
    }
    __syncthreads();
    const int HW4 = HW >> 2;
    const int p4 = blockIdx.x * 256 + tid;
    if (p4 >= HW4) return;
    float4 gv[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) gv[o] = reinterpret_cast<const float4*>(g + ((size_t)b * NOUT + o) * HW)[p4];
    float4* op = reinterpret_cast<float4*>(gx + (size_t)b * K * HW) + p4;
    const float gain = act == 3 ? 1.4142135623730951f : 1.f;
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
        const float c0 = ws[0][k], c1 = ws[1][k], c2 = ws[2][k];
        const float bb = bias ? bias[k] : 0.f;
        float4 r;
        r.x = c0 * gv[0].x + c1 * gv[1].x + c2 * gv[2].x + bb;
        r.y = c0 * gv[0].y + c1 * gv[1].y + c2 * gv[2].y + bb;
        r.z = c0 * gv[0].z + c1 * gv[1].z + c2 * gv[2].z + bb;
        r.w = c0 * gv[0].w + c1 * gv[1].w + c2 * gv[2].w + bb;
        if (act >= 3) {          // block-uniform
            r.x = (r.x > 0.f ? r.x : 0.2f * r.x) * gain; r.y = (r.y > 0.f ? r.y : 0.2f * r.y) * gain;
            r.z = (r.z > 0.f ? r.z : 0.2f * r.z) * gain; r.w = (r.w > 0.f ? r.w : 0.2f * r.w) * gain;
        }
        op[(size_t)k * HW4] = r;
    }
}

// one block per (sample, pixel chunk); x tile [K][TPX] staged in LDS (row stride TPX + 1: conflict-free column reads).
// Round 4: the tile width scales with the channel count (K * TPX ~ 8192 elements: 32 pixels at 256+ channels, 256 at 32) and
// ALL 256 threads multiply: 256 / K2 threads share a channel (K2 = K rounded up to a power of two), each takes every
// (256 / K2)-th pixel of the tile, and their partial dot products meet through LDS in a fixed order at the end.  Before, thread k
// owned channel k alone: at the 32 / 64-channel layers of the FFHQ-1024 tail 1/8 - 1/4 of the block computed, behind a barrier
// pair every 32 pixels (1.9 TB/s at 32 channels @1024^2; 3.9 at 128 channels @256^2).
// SUM: a 4th slab row holds sum_p x[b,k,p] (the bias gradient of the from-RGB stem comes out of the same pass over x)
inline int rgb_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
inline int rgb_tile_px(int K) { return std::max(32, std::min(256, 8192 / rgb_pow2(K))); }

template <bool SUM>
__global__ __launch_bounds__(256) void rgb_wgrad_kernel(float* __restrict__ slabs, const float* __restrict__ g,
                                                        const float* __restrict__ x, int K, int HW, int S, int TPX, int lgTPX, int K2) {
    constexpr int NROW = SUM ? NOUT + 1 : NOUT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xl = smem;                      // [K][TPX + 1]
    float* gl = smem + K * (TPX + 1);      // [NOUT][TPX]
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const int ntile = (HW + TPX - 1) / TPX;
    const int t0 = (int)((int64_t)ntile * c / S), t1 = (int)((int64_t)ntile * (c + 1) / S);
    const float* xb = x + (size_t)b * K * HW;
    const float* gb = g + (size_t)b * NOUT * HW;
    const bool vec4 = (HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);      // block-uniform
    // compute-phase geometry: K2 <= 256: `parts` threads per channel; K2 == 512: every thread owns two channels
    const int parts = K2 >= 256 ? 1 : 256 / K2;
    const int kk = K2 >= 256 ? tid : (tid & (K2 - 1)), part = K2 >= 256 ? 0 : tid / K2;
    const int nh = K2 > 256 ? 2 : 1;
    float acc[2][NOUT + 1] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int tl = t0; tl < t1; ++tl) {
        const int p0 = tl * TPX;
        __syncthreads();
        if (vec4) {        // rows are 16-byte aligned (HW % 4 == 0): 8 float4 loads in flight per lane = a whole 8192-element tile in ONE
                           // round (the scalar form below needs four rounds, and the kernel is bound by their latency, not by HBM)
            for (int e0 = 0; e0 < K * TPX; e0 += 256 * 8 * 4) {
                float4 st[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + 4 * (tid + 256 * j), k = e >> lgTPX, pp = e & (TPX - 1);
                    const bool ok = e < K * TPX && p0 + pp < HW;
                    st[j] = ok ? *reinterpret_cast<const float4*>(xb + (size_t)k * HW + p0 + pp) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + 4 * (tid + 256 * j), k = e >> lgTPX, pp = e & (TPX - 1);
                    if (e < K * TPX) {
                        float* d = xl + k * (TPX + 1) + pp;
                        d[0] = st[j].x; d[1] = st[j].y; d[2] = st[j].z; d[3] = st[j].w;
                    }
                }
            }
        } else {
            for (int e0 = 0; e0 < K * TPX; e0 += 256 * 8) {        // 8 loads in flight per lane, then the LDS writes
                float st[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + tid + 256 * j, k = e >> lgTPX, pp = e & (TPX - 1);
                    const bool ok = e < K * TPX && p0 + pp < HW;
                    const float v = xb[ok ? (size_t)k * HW + p0 + pp : 0];
                    st[j] = ok ? v : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + tid + 256 * j, k = e >> lgTPX, pp = e & (TPX - 1);
                    if (e < K * TPX) xl[k * (TPX + 1) + pp] = st[j];
                }
            }
        }
        for (int e = tid; e < NOUT * TPX; e += 256) {
            const int o = e >> lgTPX, pp = e & (TPX - 1);
            gl[e] = (p0 + pp < HW) ? gb[(size_t)o * HW + p0 + pp] : 0.f;
        }
        __syncthreads();
        for (int h = 0; h < nh; ++h) {
            const int k = kk + 256 * h;
            if (k < K) {
#pragma unroll 4
                for (int pp = part; pp < TPX; pp += parts) {
                    const float xv = xl[k * (TPX + 1) + pp];
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) acc[h][o] += gl[o * TPX + pp] * xv;
                    if (SUM) acc[h][NOUT] += xv;
                }
            }
        }
    }
    float* sl = slabs + ((size_t)b * S + c) * NROW * K;
    if (parts > 1) {                       // the channel's `parts` partial sums meet in LDS, added in a fixed order
        __syncthreads();
        float* red = smem;                 // [parts][NROW][K2]  (<= 256 * NROW floats: fits in the x tile)
#pragma unroll
        for (int o = 0; o < NROW; ++o) red[(part * NROW + o) * K2 + kk] = acc[0][o];
        __syncthreads();
        if (part == 0 && kk < K) {
#pragma unroll
            for (int o = 0; o < NROW; ++o) {
                float a = 0.f;
                for (int q = 0; q < parts; ++q) a += red[(q * NROW + o) * K2 + kk];
                sl[o * K + kk] = a;
            }
        }
        return;
    }
    for (int h = 0; h < nh; ++h) {
        const int k = kk + 256 * h;
        if (k < K) {
#pragma unroll
            for (int o = 0; o < NROW; ++o) sl[o * K + k] = acc[h][o];
        }
    }
}

inline bool ok_shape(int K, int HW) { return K > 0 && K <= KMAX && HW > 0 && (HW & 3) == 0; }

}  // namespace

extern "C" int te_rgb_supported(int M, int K, int HW) { return (M == NOUT && ok_shape(K, HW)) ? 1 : 0; }

extern "C" int te_rgb_fwd_f32(float* out, const float* x, const float* w, const float* isc, const float* bias, float wscale,
                              int B, int K, int HW, te_stream_t stream_) {
    TE_REQUIRE(out && x && w, TE_ERR_NULL, "te_rgb_fwd_f32: NULL pointer");
    TE_REQUIRE(B > 0 && ok_shape(K, HW), TE_ERR_UNSUPPORTED, "te_rgb_fwd_f32: need K <= 512 and H*W %% 4 == 0");
    dim3 grid((unsigned)te::cdiv(HW / 4, 256), (unsigned)B);
    rgb_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>(out, x, w, isc, bias, wscale, K, HW);
    return te::launch_status("te_rgb_fwd_f32");
}

extern "C" int te_rgb_dgrad_f32(float* gx, const float* g, const float* w, const float* isc, float wscale, int B, int K,
                                int HW, te_stream_t stream_) {
    TE_REQUIRE(gx && g && w, TE_ERR_NULL, "te_rgb_dgrad_f32: NULL pointer");
    TE_REQUIRE(B > 0 && ok_shape(K, HW), TE_ERR_UNSUPPORTED, "te_rgb_dgrad_f32: need K <= 512 and H*W %% 4 == 0");
    dim3 grid((unsigned)te::cdiv(HW / 4, 256), (unsigned)B);
    rgb_dgrad_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>(gx, g, w, isc, wscale, K, HW);
    return te::launch_status("te_rgb_dgrad_f32");
}

extern "C" int te_rgb_expand_f32(float* out, const float* x3, const float* w, const float* bias, int act, float wscale, int B,
                                 int K, int HW, te_stream_t stream_) {
    TE_REQUIRE(out && x3 && w, TE_ERR_NULL, "te_rgb_expand_f32: NULL pointer");
    TE_REQUIRE(B > 0 && ok_shape(K, HW), TE_ERR_UNSUPPORTED, "te_rgb_expand_f32: need K <= 512 and H*W %% 4 == 0");
    TE_REQUIRE(act == 0 || act == 3 || act == 4, TE_ERR_UNSUPPORTED, "te_rgb_expand_f32: act must be 0, 3 or 4");
    dim3 grid((unsigned)te::cdiv(HW / 4, 256), (unsigned)B);
    rgb_dgrad_kernel<<<grid, 256, 0, (hipStream_t)stream_>>>(out, x3, w, nullptr, wscale, K, HW, bias, act);
    return te::launch_status("te_rgb_expand_f32");
}

extern "C" int te_rgb_wgrad_slab_count(int B, int K, int HW) {
    if (B <= 0 || K <= 0 || HW <= 0) return TE_ERR_SHAPE;
    const int64_t ntile = te::cdiv(HW, rgb_tile_px(K));
    int64_t S = te::cdiv(4 * te::kNumCU, B);
    return (int)std::max<int64_t>(1, std::min<int64_t>(S, ntile));
}

extern "C" int te_rgb_wgrad_f32(float* slabs, const float* g, const float* x, int B, int K, int HW, int S, te_stream_t stream_) {
    TE_REQUIRE(slabs && g && x, TE_ERR_NULL, "te_rgb_wgrad_f32: NULL pointer");
    TE_REQUIRE(B > 0 && K > 0 && K <= KMAX && HW > 0 && S > 0, TE_ERR_UNSUPPORTED, "te_rgb_wgrad_f32: need 0 < K <= 512");
    const int TPX = rgb_tile_px(K), K2 = rgb_pow2(K);
    int lg = 0;
    while ((1 << lg) < TPX) ++lg;
    const size_t lds = sizeof(float) * std::max<size_t>((size_t)K * (TPX + 1) + NOUT * TPX, (size_t)256 * (NOUT + 1));
    static std::atomic<uint64_t> attr_done{0};
    te::allow_big_lds(attr_done, (const void*)rgb_wgrad_kernel<false>, 96 * 1024);
    dim3 grid((unsigned)S, (unsigned)B);
    rgb_wgrad_kernel<false><<<grid, 256, lds, (hipStream_t)stream_>>>(slabs, g, x, K, HW, S, TPX, lg, K2);
    return te::launch_status("te_rgb_wgrad_f32");
}

extern "C" int te_rgb_wgrad_sum_f32(float* slabs, const float* g, const float* x, int B, int K, int HW, int S, te_stream_t stream_) {
    TE_REQUIRE(slabs && g && x, TE_ERR_NULL, "te_rgb_wgrad_sum_f32: NULL pointer");
    TE_REQUIRE(B > 0 && K > 0 && K <= KMAX && HW > 0 && S > 0, TE_ERR_UNSUPPORTED, "te_rgb_wgrad_sum_f32: need 0 < K <= 512");
    const int TPX = rgb_tile_px(K), K2 = rgb_pow2(K);
    int lg = 0;
    while ((1 << lg) < TPX) ++lg;
    const size_t lds = sizeof(float) * std::max<size_t>((size_t)K * (TPX + 1) + NOUT * TPX, (size_t)256 * (NOUT + 1));
    static std::atomic<uint64_t> attr_done{0};
    te::allow_big_lds(attr_done, (const void*)rgb_wgrad_kernel<true>, 96 * 1024);
    dim3 grid((unsigned)S, (unsigned)B);
    rgb_wgrad_kernel<true><<<grid, 256, lds, (hipStream_t)stream_>>>(slabs, g, x, K, HW, S, TPX, lg, K2);
    return te::launch_status("te_rgb_wgrad_sum_f32");
}
