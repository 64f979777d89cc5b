// Shared by the convolution kernels: vector types, buffer-descriptor loads, packed-weight layout.
#pragma once
#include "te_common.h"
#include <stdlib.h>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector (HIP's float4 struct copies lower to memcpy and stay in scratch)

constexpr unsigned OOBH = 0x40000000u;   // "out of bounds" half-offset: any sum containing it exceeds num_records -> load returns 0

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

constexpr int KPAD = 16;    // packed weights Wp[tap][Kp][Mp]: K padded to a multiple of 16 (stages take 8, 16 or 32 channels)
constexpr int MPAD = 128;   // ... M padded to a multiple of 128 (block tiles cover 32, 64 or 128 rows)
constexpr int NTHREADS = 256;


// csrc/wino.hip: the 1-D Winograd F(2,3) form of the 3x3 / stride 1 / pad 1 convolution (kind TE_CONV_3X3W); weights packed
// TE_PACK_WFWD / TE_PACK_WDGRAD as U[K/8][ky][component][8][M]
int te_wino_launch(float* out, const float* in, const float* U, const float* isc, const float* osc, const float* bias, const float* res,
                   const float* mask_ref, float mask_gain, int act, int B, int K, int M, int H, int W, hipStream_t s);
// csrc/wino6.hip: the same form with its products on the bf16 matrix pipe (three-piece split, six products: fp32-equivalent);
// kind TE_CONV_3X3W6, weights packed TE_PACK_W6FWD / TE_PACK_W6DGRAD in MFMA fragment order
int te_wino6_launch(float* out, const float* in, const float* U, const float* isc, const float* osc, const float* bias, const float* res,
                    const float* mask_ref, float mask_gain, int act, int B, int K, int M, int H, int W, hipStream_t s);
// csrc/s2s6.hip: the 3x3 / stride 2 / pad 0 convolution on the bf16 matrix pipe (three-piece split, six products: fp32-equivalent);
// kind TE_CONV_S2S6, weights packed TE_PACK_S6FWD / TE_PACK_S6SWAP in MFMA fragment order
int te_s2s6_launch(float* out, const float* in, const float* U, const float* isc, const float* osc, const float* bias, const float* res,
                   const float* mask_ref, float mask_gain, int act, int B, int K, int M, int H, int W, hipStream_t s);
// csrc/t2s6.hip: body cells of the transposed 3x3 / stride 2 convolution on the bf16 matrix pipe; kind TE_CONV_T2S6 (conv.hip adds the
// last output row / column through the fp32 kernel), weights packed TE_PACK_T6FWD / TE_PACK_T6SWAP
int te_t2s6_launch(float* out, const float* in, const float* U, const float* isc, const float* osc, const float* bias, int act,
                   int B, int K, int M, int H, int W, hipStream_t s, float* colbuf = nullptr);
// ... and its last output row / column as one small vector-ALU launch (round 6; wplain = the Wp[tap][Kp][Mp] copy behind the split layout;
// colbuf: optional [B][K][H] floats in which the body kernel leaves the scaled last input column for the edge kernel)
int te_t2s6_edge_launch(float* out, const float* in, const float* wplain, const float* isc, const float* osc, const float* bias, int act,
                        int B, int K, int M, int H, int W, int Kp, int Mp, hipStream_t s, const float* colbuf = nullptr);
// csrc/p1s6.hip (round 6): the 1x1 convolution on the bf16 matrix pipe (three-piece split: fp32-equivalent), plain product + optional
// residual; kind TE_CONV_1X1S6, weights packed TE_PACK_P6FWD / TE_PACK_P6DGRAD in MFMA fragment order
int te_p1s6_launch(float* out, const float* in, const float* U, const float* res, int B, int K, int M, int H, int W, hipStream_t s);
