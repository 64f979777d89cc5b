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

__global__ __launch_bounds__(256) void chan_scale_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                         const float* __restrict__ s, int64_t n, int64_t hw) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = x[i] * s[i / hw];
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
        for (int64_t j = tid; j < hw / 4; j += 256) {
            const float4 u = a4[j], v = b4[j];
            acc += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
        }
    } else {
        for (int64_t j = tid; j < hw; j += 256) acc += ar[j] * br[j];
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
    } else {
        const int grid = (int)std::min<int64_t>(te::cdiv(n, 256), te::kNumCU * 16);
        chan_scale_kernel<<<grid, 256, 0, st>>>(out, x, s, n, hw);
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
