// Per-(sample, channel) scaling of an activation tensor and its adjoint, the two elementwise pieces of the any-order
// composite of the modulated convolution (op/modconv.py::_composite: `x * s[:, :, None, None]` and `y * d[:, :, None, None]`,
// model_spatial_query.py:299-304 applied to activations instead of B weight copies):
//
//     te_chan_scale_f32 : out[r, j] = x[r, j] * s[r]                 r = (sample, channel) row, j = pixel
//     te_chan_dot_f32   : out[r]    = sum_j a[r, j] * b[r, j]        (gradient of the scale; also the gradient of a gradient)
//
// Each is the other's derivative, so the pair is closed under differentiation (path-length regulariser: double backward).
// HBM streaming, 16 bytes per lane; the dot is one block per row with a fixed-order reduction (deterministic).
#include "te_common.h"

namespace {

__global__ __launch_bounds__(256) void chan_scale_vec4_kernel(float4* __restrict__ out, const float4* __restrict__ x,
                                                              const float* __restrict__ s, int64_t n4, uint32_t hw4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float sc = s[i / hw4];
        float4 v = x[i];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        out[i] = v;
    }
}

// rows of any length (the transposed convolution's (2H+1)^2 planes): 16 bytes per lane on the FLAT index; the four elements of a
// vector share a row unless the vector straddles a row boundary (then the second row's scale applies from the boundary on)
__global__ __launch_bounds__(256) void chan_scale_flat4_kernel(float4* __restrict__ out, const float4* __restrict__ x,
                                                               const float* __restrict__ s, int64_t n4, int64_t hw) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int64_t e = 4 * i, r = e / hw;
        const int left = (int)(hw - (e - r * hw));           // elements of row r from e on (>= 1)
        const float s0 = s[r], s1 = left < 4 ? s[r + 1] : s0; // hw >= 4 here: a vector touches at most two rows
        float4 v = x[i];
        v.x *= s0;
        v.y *= left > 1 ? s0 : s1;
        v.z *= left > 2 ? s0 : s1;
        v.w *= left > 3 ? s0 : s1;
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void chan_scale_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                         const float* __restrict__ s, int64_t n, int64_t hw, int64_t first) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = x[i] * s[i / hw];
}

// one block per row; rows with few pixels (hw <= 64): one wave per row, 4 rows per block
__global__ __launch_bounds__(256) void chan_dot_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                       const float* __restrict__ b, int64_t rows, int64_t hw, int vec) {
    __shared__ float part[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (hw <= 64) {
        const int64_t r = (int64_t)blockIdx.x * 4 + wid;
        float acc = 0.f;
        if (r < rows && lane < hw) acc = a[r * hw + lane] * b[r * hw + lane];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (r < rows && lane == 0) out[r] = acc;
        return;
    }
    const int64_t r = blockIdx.x;
    const float* ar = a + r * hw;
    const float* br = b + r * hw;
    float acc = 0.f;
    if (vec) {
        const float4* a4 = reinterpret_cast<const float4*>(ar);
        const float4* b4 = reinterpret_cast<const float4*>(br);
        const int64_t n4 = hw / 4;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;      // four independent chains: eight 16-byte loads in flight per lane
        int64_t j = tid;
        for (; j + 768 < n4; j += 1024) {
            const float4 u0 = a4[j], u1 = a4[j + 256], u2 = a4[j + 512], u3 = a4[j + 768];
            const float4 v0 = b4[j], v1 = b4[j + 256], v2 = b4[j + 512], v3 = b4[j + 768];
            c0 += u0.x * v0.x + u0.y * v0.y + u0.z * v0.z + u0.w * v0.w;
            c1 += u1.x * v1.x + u1.y * v1.y + u1.z * v1.z + u1.w * v1.w;
            c2 += u2.x * v2.x + u2.y * v2.y + u2.z * v2.z + u2.w * v2.w;
            c3 += u3.x * v3.x + u3.y * v3.y + u3.z * v3.z + u3.w * v3.w;
        }
        for (; j < n4; j += 256) {
            const float4 u = a4[j], v = b4[j];
            c0 += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
        }
        acc = (c0 + c1) + (c2 + c3);
    } else {
        float c0 = 0.f, c1 = 0.f;
        int64_t j = tid;
        for (; j + 256 < hw; j += 512) { c0 += ar[j] * br[j]; c1 += ar[j + 256] * br[j + 256]; }
        for (; j < hw; j += 256) c0 += ar[j] * br[j];
        acc = c0 + c1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) part[wid] = acc;
    __syncthreads();
    if (tid == 0) out[r] = (part[0] + part[1]) + (part[2] + part[3]);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int te_chan_scale_f32(float* out, const float* x, const float* s, int64_t rows, int64_t hw, te_stream_t stream_) {
    TE_REQUIRE(out && x && s, TE_ERR_NULL, "te_chan_scale_f32: NULL pointer");
    TE_REQUIRE(rows >= 0 && hw > 0, TE_ERR_SHAPE, "te_chan_scale_f32: bad dims");
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream_;
    const int64_t n = rows * hw;
    if (hw % 4 == 0 && aligned16(out) && aligned16(x) && hw / 4 < (int64_t)0xFFFFFFFF) {
        const int grid = (int)std::min<int64_t>(te::cdiv(n / 4, 256), te::kNumCU * 16);
        chan_scale_vec4_kernel<<<grid, 256, 0, st>>>((float4*)out, (const float4*)x, s, n / 4, (uint32_t)(hw / 4));
    } else if (hw >= 4 && n >= 1024 && aligned16(out) && aligned16(x)) {
        const int grid = (int)std::min<int64_t>(te::cdiv(n / 4, 256), te::kNumCU * 16);
        chan_scale_flat4_kernel<<<grid, 256, 0, st>>>((float4*)out, (const float4*)x, s, n / 4, hw);
        if (n % 4) chan_scale_kernel<<<1, 64, 0, st>>>(out, x, s, n, hw, n - n % 4);
    } else {
        const int grid = (int)std::min<int64_t>(te::cdiv(n, 256), te::kNumCU * 16);
        chan_scale_kernel<<<grid, 256, 0, st>>>(out, x, s, n, hw, 0);
    }
    return te::launch_status("te_chan_scale_f32");
}

extern "C" int te_chan_dot_f32(float* out, const float* a, const float* b, int64_t rows, int64_t hw, te_stream_t stream_) {
    TE_REQUIRE(out && a && b, TE_ERR_NULL, "te_chan_dot_f32: NULL pointer");
    TE_REQUIRE(rows >= 0 && hw > 0 && rows < (int64_t)0x7FFFFFFF, TE_ERR_SHAPE, "te_chan_dot_f32: bad dims");
    if (rows == 0) return 0;
    const int vec = (hw % 4 == 0 && aligned16(a) && aligned16(b)) ? 1 : 0;
    const int64_t blocks = hw <= 64 ? te::cdiv(rows, 4) : rows;
    chan_dot_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream_>>>(out, a, b, rows, hw, vec);
    return te::launch_status("te_chan_dot_f32");
}
