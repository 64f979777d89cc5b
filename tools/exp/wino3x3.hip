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
// {2 wr, 2 wr + 1} x 16 pairs x 4 components = 8 accumulator tiles of v_mfma_f32_32x32x2_f32.  WKC input channels per stage;
// the loads of stage s + 1 are issued before the MFMAs of stage s (register prefetch).  WDB = 0: one LDS image, transformed
// tile written after the MFMAs (two barriers per stage); WDB = 1: two images, the next tile is written between the two halves of
// the current stage's MFMAs (one barrier per stage).
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef WKC
#define WKC 8
#endif
#ifndef WDB
#define WDB 0
#endif
#ifndef WOCC
#define WOCC 2
#endif
#ifndef WNR          // waves along the rows of the tile: 2 -> 4 rows; 4 -> 8 rows
#define WNR 2
#endif
#ifndef WMT          // 32-row M tiles per wave: 2 (128 accumulator registers, 2 waves along M) or 1 (64 registers, 4 waves along M)
#define WMT 2
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NMW = 4 / WMT;          // waves along M
constexpr int NTHR = 64 * NMW * WNR;
constexpr int KC = WKC, TH = 2 * WNR, TW = 32, NP = TW / 2, BM = 128, RS = (TH + 2) * NP;     // RS: floats per (channel, component) plane
constexpr int T_FLOATS = KC * 4 * RS;
constexpr int U_FLOATS = 3 * 4 * KC * BM;
constexpr int IMG = T_FLOATS + U_FLOATS;
constexpr int N_IN = (KC * RS + NTHR - 1) / NTHR;               // input items (channel, row, pair) per thread
constexpr int N_W4 = U_FLOATS / 4 / NTHR;                  // weight float4 per thread
constexpr bool IN_EXACT = (KC * RS) % NTHR == 0;
static_assert(U_FLOATS % (4 * NTHR) == 0, "weight staging split");

struct WinoArgs {
    float* out; const float* in; const float* U; const float* isc;
    int B, K, M, H, W;
};

__global__ __launch_bounds__(NTHR, WOCC) void wino3x3_kernel(const WinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm = wid / WNR, wr = wid % WNR;
    const int tiles_x = p.W / TW, tiles_y = p.H / TH, mblocks = p.M / BM;
    int t = blockIdx.x;
    const int mb = t % mblocks; t /= mblocks;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const float* inb = p.in + (size_t)b * p.K * p.H * p.W;
    const float* iscb = p.isc ? p.isc + (size_t)b * p.K : nullptr;
    const bool edge = (x0 == 0) || (x0 + TW == p.W) || (y0 == 0) || (y0 + TH == p.H);      // block-uniform

    f32x16 acc[WMT][4];
#pragma unroll
    for (int mt = 0; mt < WMT; ++mt)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][c][r] = 0.f;

    // staging geometry: item e = tid + 256 i -> (channel, row, pair); one element offset and one LDS offset per item
    int g_off[N_IN], l_off[N_IN];
#pragma unroll
    for (int i = 0; i < N_IN; ++i) {
        const int e = tid + NTHR * i;
        const int pr = e % NP, row = (e / NP) % (TH + 2), ch = e / RS;
        g_off[i] = (ch * p.H + (y0 - 1 + row)) * p.W + x0 + 2 * pr - 1;
        l_off[i] = (IN_EXACT || e < KC * RS) ? ch * 4 * RS + row * NP + pr : -1;
    }
    const int w_off = (tid >> 5) * p.M + mb * BM + 4 * (tid & 31);          // float4 i: + i * (NTHR / 32) * M
    const int stage_in = KC * p.H * p.W, stage_w = 3 * 4 * KC * p.M;
    f32x4 rin[N_IN];
    float rsc[N_IN];
    f32x4 rw[N_W4];
    const int nstage = p.K / KC;
    auto issue = [&](int s) {
        const float* base = inb + (size_t)s * stage_in;
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            if (!IN_EXACT && l_off[i] < 0) continue;
            const int e = tid + NTHR * i, ch = e / RS;
            rsc[i] = iscb ? iscb[s * KC + ch] : 1.f;
            if (!edge) {
                rin[i] = *reinterpret_cast<const f32x4u*>(base + g_off[i]);
            } else {
                const int pr = e % NP, row = (e / NP) % (TH + 2);
                const int gy = y0 - 1 + row, gx = x0 + 2 * pr - 1;
                const bool rowok = gy >= 0 && gy < p.H;
#pragma unroll
                for (int q = 0; q < 4; ++q) rin[i][q] = (rowok && gx + q >= 0 && gx + q < p.W) ? base[g_off[i] + q] : 0.f;
            }
        }
        const float* us = p.U + (size_t)s * stage_w + w_off;
#pragma unroll
        for (int i = 0; i < N_W4; ++i) rw[i] = *reinterpret_cast<const f32x4*>(us + (size_t)i * (NTHR / 32) * p.M);
    };
    auto commit = [&](float* img) {
        float* Tl = img;
        float* Ul = img + T_FLOATS;
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            if (!IN_EXACT && l_off[i] < 0) continue;
            const float sc = rsc[i];
            const float d0 = rin[i][0] * sc, d1 = rin[i][1] * sc, d2 = rin[i][2] * sc, d3 = rin[i][3] * sc;
            float* dst = Tl + l_off[i];
            dst[0] = d0 - d2;
            dst[RS] = d1 + d2;
            dst[2 * RS] = d2 - d1;
            dst[3 * RS] = d1 - d3;
        }
#pragma unroll
        for (int i = 0; i < N_W4; ++i) *reinterpret_cast<f32x4*>(Ul + 4 * (tid + NTHR * i)) = rw[i];
    };
    const int rr = l31 >> 4, jj = l31 & 15;
    const int b_off = half * 4 * RS + (2 * wr + rr) * NP + jj;          // + (2 ks * 4 + c) * RS + ky * NP
    const int a_off = T_FLOATS + half * BM + wm * (32 * WMT) + l31;             // + ((ky * 4 + c) * KC + 2 ks) * BM
#ifndef WPIPE
#define WPIPE 1
#endif
    auto mfmas = [&](const float* img, int ks_from, int ks_to) {
#if WPIPE
        // operands of step i + WPIPE are read from LDS before the MFMAs of step i are issued (a step = one (ks, ky, c) triple)
        const int n = (ks_to - ks_from) * 12;
        float a0[WPIPE + 1], a1[WPIPE + 1], bb[WPIPE + 1];
        auto rd = [&](int st, int slot) {
            const int ks = ks_from + st / 12, ky = (st % 12) / 4, c = st % 4;
            bb[slot] = img[b_off + (2 * ks * 4 + c) * RS + ky * NP];
            const float* ua = img + a_off + ((ky * 4 + c) * KC + 2 * ks) * BM;
            a0[slot] = ua[0];
            if (WMT > 1) a1[slot] = ua[32];
        };
#pragma unroll
        for (int i = 0; i < WPIPE; ++i) rd(i, i);
#pragma unroll
        for (int st = 0; st < (KC / 2) * 12; ++st) {
            if (st >= n) break;
            if (st + WPIPE < n) rd(st + WPIPE, (st + WPIPE) % (WPIPE + 1));
            __builtin_amdgcn_sched_barrier(0);           // keep those reads in front of the MFMAs of step st
            const int c = st % 4, slot = st % (WPIPE + 1);
            acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[slot], bb[slot], acc[0][c], 0, 0, 0);
            if (WMT > 1) acc[WMT - 1][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[slot], bb[slot], acc[WMT - 1][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
#pragma unroll
        for (int ks = ks_from; ks < ks_to; ++ks) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float bv = img[b_off + (2 * ks * 4 + c) * RS + ky * NP];
                    const float* ua = img + a_off + ((ky * 4 + c) * KC + 2 * ks) * BM;
                    acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[0], bv, acc[0][c], 0, 0, 0);
                    if (WMT > 1) acc[WMT - 1][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[32], bv, acc[WMT - 1][c], 0, 0, 0);
                }
            }
        }
#endif
    };

    issue(0);
    commit(smem);
    if (WDB && nstage > 1) issue(1);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        if (WDB) {
            float* cur = smem + (s & 1) * IMG;
            float* nxt = smem + ((s & 1) ^ 1) * IMG;
            mfmas(cur, 0, KC / 4);
            if (s + 1 < nstage) commit(nxt);                 // stage s + 1: loaded while stage s - 1 was multiplied
            if (s + 2 < nstage) issue(s + 2);
            mfmas(cur, KC / 4, KC / 2);
            __syncthreads();
        } else {
            if (s + 1 < nstage) issue(s + 1);
            mfmas(smem, 0, KC / 2);
            __syncthreads();
            if (s + 1 < nstage) commit(smem);
            __syncthreads();
        }
    }
    // epilogue: output transform, two adjacent columns per accumulator element
    float* ob = p.out + ((size_t)b * p.M + mb * BM + wm * (32 * WMT)) * p.H * p.W;
    const int oy = y0 + 2 * wr + rr, ox = x0 + 2 * jj;
#pragma unroll
    for (int mt = 0; mt < WMT; ++mt) {
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
    const size_t lds = sizeof(float) * IMG * (WDB ? 2 : 1);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)wino3x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int64_t blocks = (int64_t)B * (H / TH) * (W / TW) * (M / BM);
    wino3x3_kernel<<<dim3((unsigned)blocks), NTHR, lds, (hipStream_t)stream>>>(a);
    return (int)hipGetLastError();
}

extern "C" int wino_kc(void) { return KC; }
