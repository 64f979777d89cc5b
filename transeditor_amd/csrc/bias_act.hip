// K1: fused bias + leaky-ReLU family for gfx950 (wave64).
//   te_bias_act_f32      forward / grad / grad-grad in one kernel family (HBM streaming, 16 B per lane)
//   te_bias_act_bwd_f32  gi = g * slope(ref) * scale and the per-channel bias gradient in the same pass
//                        (wave64 shuffle reduction -> LDS across the 4 waves -> one partial per block in the caller's
//                        workspace -> fixed-order second pass: bit-reproducible, no atomics)
// Semantics follow the reference op (fused_bias_act_kernel.cu:26-47); the channel index of flat
// element i is (i / step_b) % size_b with integer arithmetic.
#include "te_common.h"
#include <hip/hip_fp16.h>

namespace {

__device__ __forceinline__ float act_one(float x, float r, int mode, float alpha) {
    // mode = act*10 + grad
    switch (mode) {
        case 30: return x > 0.f ? x : x * alpha;
        case 31: return r > 0.f ? x : x * alpha;
        case 12:
        case 32: return 0.f;
        default: return x;  // 10, 11: linear
    }
}

// one float4 per lane per iteration; requires step_b % 4 == 0 (so the 4 lanes of a vector share a channel)
__global__ __launch_bounds__(256) void bias_act_vec4_kernel(float4* __restrict__ out, const float4* __restrict__ x,
                                                            const float* __restrict__ b, const float4* __restrict__ ref,
                                                            int mode, float alpha, float scale, uint32_t n4,
                                                            uint32_t step_b4, uint32_t size_b) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = x[i];
        if (b) {
            const float bb = b[(i / step_b4) % size_b];
            v.x += bb; v.y += bb; v.z += bb; v.w += bb;
        }
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ref) r = ref[i];
        float4 y;
        y.x = act_one(v.x, r.x, mode, alpha) * scale;
        y.y = act_one(v.y, r.y, mode, alpha) * scale;
        y.z = act_one(v.z, r.z, mode, alpha) * scale;
        y.w = act_one(v.w, r.w, mode, alpha) * scale;
        out[i] = y;
    }
}

__global__ __launch_bounds__(256) void bias_act_scalar_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                              const float* __restrict__ b, const float* __restrict__ ref,
                                                              int mode, float alpha, float scale, int64_t n,
                                                              int64_t step_b, int64_t size_b) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = x[i];
        if (b) v += b[(i / step_b) % size_b];
        const float r = ref ? ref[i] : 0.f;
        out[i] = act_one(v, r, mode, alpha) * scale;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

__device__ __forceinline__ float block_sum_256(float v, float* lds4) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) lds4[wid] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];  // every thread gets the total
}

// [outer][C][inner], inner % 4 == 0: grid (chunks, C, outer); each block streams CH float4-vectors of one row
constexpr int kBwdVecPerThread = 4;
__global__ __launch_bounds__(256) void bias_act_bwd_rows_kernel(float4* __restrict__ gi, float* __restrict__ part,
                                                                const float4* __restrict__ g, const float4* __restrict__ ref,
                                                                float alpha, float scale, uint32_t C, uint32_t inner4) {
    __shared__ float lds4[4];
    const uint32_t c = blockIdx.y, n = blockIdx.z;
    const size_t row = ((size_t)n * C + c) * inner4;
    const uint32_t base = blockIdx.x * (256 * kBwdVecPerThread);
    float acc = 0.f;
#pragma unroll
    for (int it = 0; it < kBwdVecPerThread; ++it) {
        const uint32_t i = base + it * 256 + threadIdx.x;
        if (i < inner4) {
            const float4 gv = g[row + i], rv = ref[row + i];
            float4 o;
            o.x = gv.x * (rv.x > 0.f ? 1.f : alpha) * scale;
            o.y = gv.y * (rv.y > 0.f ? 1.f : alpha) * scale;
            o.z = gv.z * (rv.z > 0.f ? 1.f : alpha) * scale;
            o.w = gv.w * (rv.w > 0.f ? 1.f : alpha) * scale;
            gi[row + i] = o;
            acc += (o.x + o.y) + (o.z + o.w);
        }
    }
    if (part) {
        const float tot = block_sum_256(acc, lds4);
        if (threadIdx.x == 0) part[((size_t)c * gridDim.z + n) * gridDim.x + blockIdx.x] = tot;
    }
}

// second pass of the row kernels: gb[c] = sum of the channel's per-block partials, in a FIXED order (lane-strided partial
// sums, then the wave's shuffle tree): the bias gradient is bit-identical from run to run
__global__ __launch_bounds__(64) void bias_parts_sum_kernel(float* __restrict__ gb, const float* __restrict__ part, uint32_t parts) {
    const float* p = part + (size_t)blockIdx.x * parts;
    float acc = 0.f;
    for (uint32_t j = threadIdx.x; j < parts; j += 64) acc += p[j];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) gb[blockIdx.x] = acc;
}

// The same pass with the data gradient of a ToRGB layer folded in (generator, model_spatial_query.py:416-425: the activation
// a2 feeds the next layer AND a 1x1 modulated convolution to 3 channels):
//     gi = (g + coef[0] * grgb[n,0] + coef[1] * grgb[n,1] + coef[2] * grgb[n,2]) * slope(ref),   coef[o] = wscale * srgb[n,c] * wrgb[o,c]
// i.e. te_rgb_dgrad_f32 + the framework's gradient-accumulation add + te_bias_act_bwd_f32 in one read of g / ref and one
// write of gi (12 instead of 28 bytes per element; the 3-channel grgb rows are re-read by every channel from L2).  g may be
// NULL (the last layer's activation feeds ToRGB only).
__global__ __launch_bounds__(256) void bias_act_bwd_rgb_rows_kernel(float4* __restrict__ gi, float* __restrict__ part,
                                                                    const float4* __restrict__ g, const float4* __restrict__ ref,
                                                                    const float4* __restrict__ grgb, const float* __restrict__ wrgb,
                                                                    const float* __restrict__ srgb, float wscale, float alpha,
                                                                    float scale, uint32_t C, uint32_t inner4) {
    __shared__ float lds4[4];
    const uint32_t c = blockIdx.y, n = blockIdx.z;
    const size_t row = ((size_t)n * C + c) * inner4;
    const size_t rrow = (size_t)n * 3 * inner4;
    const float sc = (srgb ? srgb[(size_t)n * C + c] : 1.f) * wscale;
    const float c0 = sc * wrgb[c], c1 = sc * wrgb[C + c], c2 = sc * wrgb[2 * C + c];
    const uint32_t base = blockIdx.x * (256 * kBwdVecPerThread);
    float acc = 0.f;
#pragma unroll
    for (int it = 0; it < kBwdVecPerThread; ++it) {
        const uint32_t i = base + it * 256 + threadIdx.x;
        if (i < inner4) {
            const float4 rv = ref[row + i];
            const float4 r0 = grgb[rrow + i], r1 = grgb[rrow + inner4 + i], r2 = grgb[rrow + 2 * (size_t)inner4 + i];
            float4 gv = g ? g[row + i] : make_float4(0.f, 0.f, 0.f, 0.f);
            gv.x += c0 * r0.x + c1 * r1.x + c2 * r2.x;
            gv.y += c0 * r0.y + c1 * r1.y + c2 * r2.y;
            gv.z += c0 * r0.z + c1 * r1.z + c2 * r2.z;
            gv.w += c0 * r0.w + c1 * r1.w + c2 * r2.w;
            float4 o;
            o.x = gv.x * (rv.x > 0.f ? 1.f : alpha) * scale;
            o.y = gv.y * (rv.y > 0.f ? 1.f : alpha) * scale;
            o.z = gv.z * (rv.z > 0.f ? 1.f : alpha) * scale;
            o.w = gv.w * (rv.w > 0.f ? 1.f : alpha) * scale;
            gi[row + i] = o;
            acc += (o.x + o.y) + (o.z + o.w);
        }
    }
    if (part) {
        const float tot = block_sum_256(acc, lds4);
        if (threadIdx.x == 0) part[((size_t)c * gridDim.z + n) * gridDim.x + blockIdx.x] = tot;
    }
}

// generic layout: one block per channel, loops over (n, i): the block owns the channel's sum (plain store)
__global__ __launch_bounds__(256) void bias_act_bwd_chan_kernel(float* __restrict__ gi, float* __restrict__ gb,
                                                                const float* __restrict__ g, const float* __restrict__ ref,
                                                                float alpha, float scale, int64_t outer, int64_t C,
                                                                int64_t inner) {
    __shared__ float lds4[4];
    const int64_t c = blockIdx.x;
    const int64_t per = outer * inner;
    float acc = 0.f;
    for (int64_t e = threadIdx.x; e < per; e += 256) {
        const int64_t n = e / inner, i = e - n * inner;
        const int64_t idx = (n * C + c) * inner + i;
        const float o = g[idx] * (ref[idx] > 0.f ? 1.f : alpha) * scale;
        gi[idx] = o;
        acc += o;
    }
    if (gb) {
        const float tot = block_sum_256(acc, lds4);
        if (threadIdx.x == 0) gb[c] = tot;
    }
}

// 2-D [N][C] (inner == 1): thread per channel, coalesced across c
__global__ __launch_bounds__(256) void bias_act_bwd_2d_kernel(float* __restrict__ gi, float* __restrict__ gb,
                                                              const float* __restrict__ g, const float* __restrict__ ref,
                                                              float alpha, float scale, int64_t N, int64_t C) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float acc = 0.f;
    for (int64_t n = 0; n < N; ++n) {
        const int64_t idx = n * C + c;
        const float o = g[idx] * (ref[idx] > 0.f ? 1.f : alpha) * scale;
        gi[idx] = o;
        acc += o;
    }
    if (gb) gb[c] = acc;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// The reference dispatches the op over half / float / double (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// fused_bias_act_kernel.cu:79) and converts its float alpha / scale arguments to scalar_t.  The model runs in fp32 (the
// streaming kernels above); the other two types take this element-per-lane kernel: arithmetic in T's own precision for
// double, in fp32 for half (one rounding at the store).
template <typename T> struct AccOf { typedef T type; };
template <> struct AccOf<__half> { typedef float type; };

template <typename T>
__global__ __launch_bounds__(256) void bias_act_any_kernel(T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ b,
                                                           const T* __restrict__ ref, int mode, float alpha_, float scale_,
                                                           int64_t n, int64_t step_b, int64_t size_b) {
    typedef typename AccOf<T>::type A;
    const A alpha = (A)(T)alpha_, scale = (A)(T)scale_;          // float -> scalar_t, as the reference's kernel arguments
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        A v = (A)x[i];
        if (b) v += (A)b[(i / step_b) % size_b];
        const A r = ref ? (A)ref[i] : (A)0;
        A y;
        switch (mode) {
            case 30: y = v > (A)0 ? v : v * alpha; break;
            case 31: y = r > (A)0 ? v : v * alpha; break;
            case 12:
            case 32: y = (A)0; break;
            default: y = v;
        }
        out[i] = (T)(y * scale);
    }
}

template <typename T>
int bias_act_any(T* out, const T* x, const T* b, const T* ref, int act, int grad, float alpha, float scale, int64_t size_x,
                 int64_t step_b, int64_t size_b, te_stream_t stream_, const char* what) {
    TE_REQUIRE(out && x, TE_ERR_NULL, "%s: out/x is NULL", what);
    TE_REQUIRE(size_x >= 0, TE_ERR_SHAPE, "%s: size_x < 0", what);
    TE_REQUIRE(!b || (step_b > 0 && size_b > 0), TE_ERR_SHAPE, "%s: bias given but step_b/size_b <= 0", what);
    if (size_x == 0) return 0;
    const int grid = (int)std::min<int64_t>(te::cdiv(size_x, 256), te::kNumCU * 8);
    bias_act_any_kernel<T><<<grid, 256, 0, (hipStream_t)stream_>>>(out, x, b, ref, act * 10 + grad, alpha, scale, size_x,
                                                                   b ? step_b : 1, b ? size_b : 1);
    return te::launch_status(what);
}

}  // namespace

extern "C" int te_bias_act_f16(void* out, const void* x, const void* b, const void* ref, int act, int grad, float alpha,
                               float scale, int64_t size_x, int64_t step_b, int64_t size_b, te_stream_t stream) {
    return bias_act_any<__half>((__half*)out, (const __half*)x, (const __half*)b, (const __half*)ref, act, grad, alpha, scale,
                                size_x, step_b, size_b, stream, "te_bias_act_f16");
}

extern "C" int te_bias_act_f64(double* out, const double* x, const double* b, const double* ref, int act, int grad, float alpha,
                               float scale, int64_t size_x, int64_t step_b, int64_t size_b, te_stream_t stream) {
    return bias_act_any<double>(out, x, b, ref, act, grad, alpha, scale, size_x, step_b, size_b, stream, "te_bias_act_f64");
}

extern "C" int te_bias_act_f32(float* out, const float* x, const float* b, const float* ref, int act, int grad,
                               float alpha, float scale, int64_t size_x, int64_t step_b, int64_t size_b,
                               te_stream_t stream_) {
    TE_REQUIRE(out && x, TE_ERR_NULL, "te_bias_act_f32: out/x is NULL");
    TE_REQUIRE(size_x >= 0, TE_ERR_SHAPE, "te_bias_act_f32: size_x < 0");
    TE_REQUIRE(!b || (step_b > 0 && size_b > 0), TE_ERR_SHAPE, "te_bias_act_f32: bias given but step_b/size_b <= 0");
    if (size_x == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const int mode = act * 10 + grad;
    const bool vec = (size_x % 4 == 0) && (!b || step_b % 4 == 0) && aligned16(out) && aligned16(x) &&
                     (!ref || aligned16(ref)) && (size_x / 4 < (int64_t)0xFFFFFFFF);
    if (vec) {
        const uint32_t n4 = (uint32_t)(size_x / 4);
        const int grid = (int)std::min<int64_t>(te::cdiv(n4, 256), te::kNumCU * 8);
        bias_act_vec4_kernel<<<grid, 256, 0, stream>>>((float4*)out, (const float4*)x, b, (const float4*)ref, mode, alpha,
                                                       scale, n4, b ? (uint32_t)(step_b / 4) : 1u,
                                                       b ? (uint32_t)size_b : 1u);
    } else {
        const int grid = (int)std::min<int64_t>(te::cdiv(size_x, 256), te::kNumCU * 8);
        bias_act_scalar_kernel<<<grid, 256, 0, stream>>>(out, x, b, ref, mode, alpha, scale, size_x, b ? step_b : 1,
                                                         b ? size_b : 1);
    }
    return te::launch_status("te_bias_act_f32");
}

static bool bwd_rows_path(int64_t outer, int64_t C, int64_t inner) {
    return inner % 4 == 0 && inner >= 1024 && C <= 65535 && outer <= 65535;
}

extern "C" int64_t te_bias_act_bwd_ws_floats(int64_t outer, int64_t C, int64_t inner) {
    if (outer <= 0 || C <= 0 || inner <= 0 || !bwd_rows_path(outer, C, inner)) return 0;
    return C * outer * te::cdiv(inner / 4, 256 * kBwdVecPerThread);
}

extern "C" int te_bias_act_bwd_f32(float* gi, float* gb, float* ws, const float* g, const float* ref, float alpha, float scale,
                                   int64_t outer, int64_t C, int64_t inner, te_stream_t stream_) {
    TE_REQUIRE(gi && g && ref, TE_ERR_NULL, "te_bias_act_bwd_f32: gi/g/ref is NULL");
    TE_REQUIRE(outer >= 0 && C > 0 && inner > 0, TE_ERR_SHAPE, "te_bias_act_bwd_f32: bad dims");
    if (outer == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    if (inner == 1) {
        bias_act_bwd_2d_kernel<<<(int)te::cdiv(C, 256), 256, 0, stream>>>(gi, gb, g, ref, alpha, scale, outer, C);
    } else if (bwd_rows_path(outer, C, inner) && aligned16(gi) && aligned16(g) && aligned16(ref) && (!gb || ws)) {
        const uint32_t inner4 = (uint32_t)(inner / 4);
        dim3 grid((unsigned)te::cdiv(inner4, 256 * kBwdVecPerThread), (unsigned)C, (unsigned)outer);
        bias_act_bwd_rows_kernel<<<grid, 256, 0, stream>>>((float4*)gi, gb ? ws : nullptr, (const float4*)g, (const float4*)ref, alpha,
                                                           scale, (uint32_t)C, inner4);
        if (gb) bias_parts_sum_kernel<<<(int)C, 64, 0, stream>>>(gb, ws, (uint32_t)(grid.x * grid.z));
    } else {
        bias_act_bwd_chan_kernel<<<(int)C, 256, 0, stream>>>(gi, gb, g, ref, alpha, scale, outer, C, inner);
    }
    return te::launch_status("te_bias_act_bwd_f32");
}

extern "C" int te_bias_act_bwd_rgb_supported(int64_t outer, int64_t C, int64_t inner) {
    return (outer > 0 && C > 0 && bwd_rows_path(outer, C, inner)) ? 1 : 0;
}

extern "C" int te_bias_act_bwd_rgb_f32(float* gi, float* gb, float* ws, const float* g, const float* ref, const float* grgb,
                                       const float* wrgb, const float* srgb, float wscale, float alpha, float scale, int64_t outer,
                                       int64_t C, int64_t inner, te_stream_t stream_) {
    TE_REQUIRE(gi && ref && grgb && wrgb, TE_ERR_NULL, "te_bias_act_bwd_rgb_f32: gi/ref/grgb/wrgb is NULL");
    TE_REQUIRE(!gb || ws, TE_ERR_NULL, "te_bias_act_bwd_rgb_f32: the bias gradient needs the workspace (te_bias_act_bwd_ws_floats)");
    TE_REQUIRE(te_bias_act_bwd_rgb_supported(outer, C, inner), TE_ERR_UNSUPPORTED,
               "te_bias_act_bwd_rgb_f32: needs inner %% 4 == 0 and inner >= 1024 (use te_rgb_dgrad_f32 + te_bias_act_bwd_f32)");
    TE_REQUIRE(aligned16(gi) && (!g || aligned16(g)) && aligned16(ref) && aligned16(grgb), TE_ERR_UNSUPPORTED,
               "te_bias_act_bwd_rgb_f32: 16-byte aligned tensors required");
    const uint32_t inner4 = (uint32_t)(inner / 4);
    dim3 grid((unsigned)te::cdiv(inner4, 256 * kBwdVecPerThread), (unsigned)C, (unsigned)outer);
    hipStream_t stream = (hipStream_t)stream_;
    bias_act_bwd_rgb_rows_kernel<<<grid, 256, 0, stream>>>((float4*)gi, gb ? ws : nullptr, (const float4*)g, (const float4*)ref,
                                                          (const float4*)grgb, wrgb, srgb, wscale, alpha, scale, (uint32_t)C, inner4);
    if (gb) bias_parts_sum_kernel<<<(int)C, 64, 0, stream>>>(gb, ws, (uint32_t)(grid.x * grid.z));
    return te::launch_status("te_bias_act_bwd_rgb_f32");
}
