// EXPERIMENT (round 4, not part of the product library): 3x3 / stride 1 / pad 1 convolution with a 1-D Winograd F(2,3)
// transform along x on the fp32 matrix pipe.  Two adjacent outputs of a row come from FOUR multiplications per (input channel,
// tap row) instead of six, so the MFMA count of the implicit GEMM drops to 12/18 = 2/3 of the direct form; the price is an input
// transform on the vector ALU while the tile is staged (4 adds per pair), 4/3 of the weight bytes per stage and twice the
// accumulator registers (4 transform components per output pair).
//
//   d0..d3 = in[.., 2j-1 .. 2j+2]           t0 = d0 - d2   t1 = d1 + d2   t2 = d2 - d1   t3 = d1 - d3
//   U[ky][c] = G w[.., ky, :]  (host)        m_c = sum_{ci, ky} U[ky][c][ci] * t_c[ci][row + ky]
//   out[2j] = m0 + m1 + m2                   out[2j+1] = m1 - m2 - m3
//
// Block: 256 threads, output tile 128 (M) x 4 rows x 32 columns (= 64 pairs); wave (wm, wr): 64 output channels x rows
// {2 wr, 2 wr + 1} x 16 pairs x 4 components = 8 accumulator tiles of v_mfma_f32_32x32x2_f32.  8 input channels per stage;
// the loads of stage s + 1 are issued before the MFMAs of stage s (register prefetch), transformed and written to LDS after them.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 8, TH = 4, TW = 32, NP = TW / 2, BM = 128;
constexpr int T_FLOATS = KC * 4 * (TH + 2) * NP;          // 3072
constexpr int U_FLOATS = 3 * 4 * KC * BM;                 // 12288
constexpr int N_IN = (KC * (TH + 2) * NP + 255) / 256;    // input items (channel, row, pair) per thread: 3
constexpr int N_W4 = U_FLOATS / 4 / 256;                  // weight float4 per thread: 12

struct WinoArgs {
    float* out; const float* in; const float* U; const float* isc;
    int B, K, M, H, W;
};

__global__ __launch_bounds__(256, 2) void wino3x3_kernel(const WinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Tl = smem;                 // [KC][4][TH + 2][NP]
    float* Ul = smem + T_FLOATS;      // [3][4][KC][BM]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm = wid >> 1, wr = wid & 1;
    const int tiles_x = p.W / TW, tiles_y = p.H / TH, mblocks = p.M / BM;
    int t = blockIdx.x;
    const int mb = t % mblocks; t /= mblocks;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const float* inb = p.in + (size_t)b * p.K * p.H * p.W;
    const bool edge = (x0 == 0) || (x0 + TW == p.W) || (y0 == 0) || (y0 + TH == p.H);

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][c][r] = 0.f;

    // per-thread staging geometry
    int i_ch[N_IN], i_row[N_IN], i_pair[N_IN];
#pragma unroll
    for (int i = 0; i < N_IN; ++i) {
        const int e = tid + 256 * i;
        i_pair[i] = e % NP; i_row[i] = (e / NP) % (TH + 2); i_ch[i] = e / (NP * (TH + 2));
    }
    f32x4 rin[N_IN];
    f32x4 rw[N_W4];
    const int nstage = p.K / KC;
    auto issue = [&](int s) {
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            const int gy = y0 - 1 + i_row[i], gx = x0 + 2 * i_pair[i] - 1, ch = s * KC + i_ch[i];
            const float* src = inb + ((size_t)ch * p.H + gy) * p.W + gx;
            if (!edge) {
                rin[i] = *reinterpret_cast<const f32x4u*>(src);
            } else {
                const bool rowok = gy >= 0 && gy < p.H;
#pragma unroll
                for (int q = 0; q < 4; ++q) rin[i][q] = (rowok && gx + q >= 0 && gx + q < p.W) ? src[q] : 0.f;
            }
        }
        const float* us = p.U + (size_t)s * 3 * 4 * KC * p.M;
#pragma unroll
        for (int i = 0; i < N_W4; ++i) {
            const int idx = tid + 256 * i, row = idx >> 5, c4 = idx & 31;       // row = (ky * 4 + c) * KC + k
            rw[i] = *reinterpret_cast<const f32x4*>(us + (size_t)row * p.M + mb * BM + 4 * c4);
        }
    };
    auto commit = [&](int s) {
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            const float sc = p.isc ? p.isc[(size_t)b * p.K + s * KC + i_ch[i]] : 1.f;
            const float d0 = rin[i][0] * sc, d1 = rin[i][1] * sc, d2 = rin[i][2] * sc, d3 = rin[i][3] * sc;
            float* dst = Tl + ((i_ch[i] * 4) * (TH + 2) + i_row[i]) * NP + i_pair[i];
            dst[0] = d0 - d2;
            dst[(TH + 2) * NP] = d1 + d2;
            dst[2 * (TH + 2) * NP] = d2 - d1;
            dst[3 * (TH + 2) * NP] = d1 - d3;
        }
#pragma unroll
        for (int i = 0; i < N_W4; ++i) *reinterpret_cast<f32x4*>(Ul + 4 * (tid + 256 * i)) = rw[i];
    };

    issue(0);
    commit(0);
    __syncthreads();
    const int rr = l31 >> 4, jj = l31 & 15;
    for (int s = 0; s < nstage; ++s) {
        if (s + 1 < nstage) issue(s + 1);
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            const int ch = 2 * ks + half;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float bv = Tl[((ch * 4 + c) * (TH + 2) + 2 * wr + rr + ky) * NP + jj];
                    const float* ua = Ul + ((ky * 4 + c) * KC + ch) * BM + wm * 64 + l31;
                    acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[0], bv, acc[0][c], 0, 0, 0);
                    acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[32], bv, acc[1][c], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (s + 1 < nstage) commit(s + 1);
        __syncthreads();
    }
    // epilogue: output transform, two adjacent columns per accumulator element
    float* ob = p.out + ((size_t)b * p.M + mb * BM + wm * 64) * p.H * p.W;
    const int oy = y0 + 2 * wr + rr, ox = x0 + 2 * jj;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            f32x2 v;
            v[0] = acc[mt][0][r] + acc[mt][1][r] + acc[mt][2][r];
            v[1] = acc[mt][1][r] - acc[mt][2][r] - acc[mt][3][r];
            *reinterpret_cast<f32x2*>(ob + ((size_t)m * p.H + oy) * p.W + ox) = v;
        }
    }
}

extern "C" int wino3x3_f32(float* out, const float* in, const float* U, const float* isc, int B, int K, int M, int H, int W,
                           void* stream) {
    if (K % KC || M % BM || H % TH || W % TW) return -1;
    WinoArgs a{out, in, U, isc, B, K, M, H, W};
    const size_t lds = sizeof(float) * (T_FLOATS + U_FLOATS);
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)wino3x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int64_t blocks = (int64_t)B * (H / TH) * (W / TW) * (M / BM);
    wino3x3_kernel<<<dim3((unsigned)blocks), 256, lds, (hipStream_t)stream>>>(a);
    return (int)hipGetLastError();
}
