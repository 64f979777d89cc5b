// EXPERIMENT: the 1-D Winograd F(2,3) 3x3 kernel (see csrc/wino.hip) with PING-PONG wave groups.  512 threads = two groups of
// four waves; a group owns a 128 (M) x 4 rows x 32 columns output tile (the block: 8 rows), every SIMD holds one wave of each
// group.  Time is cut into half-stages: while group A issues the 96 MFMAs of its stage s, group B writes what it prefetched
// (its transformed input tile and one half of the NEXT stage's weights) to LDS, then the roles swap.  The matrix pipe of a SIMD
// always has exactly one wave streaming MFMAs, the other wave's staging work (global-load waits, transform, LDS writes) hides
// behind it; the 4-wave kernel left that to chance (two blocks per CU, whatever phase they happen to be in): 69 % utilisation.
//   LDS: U double-buffered (2 x 48 KB: stage s is read by both groups at different times), T per group (2 x 12 KB).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KC = 8, GH = 4, TH = 8, TW = 32, NP = TW / 2, BM = 128, RS = (GH + 2) * NP;   // RS: floats per (channel, component) plane of a group
constexpr int T_FLOATS = KC * 4 * RS;                     // 3072 per group
constexpr int U_FLOATS = 3 * 4 * KC * BM;                 // 12288 per stage
constexpr int N_IN = KC * RS / 256;                       // 3 items per thread of a group
constexpr int N_W4 = U_FLOATS / 2 / 4 / 256;              // 6 float4 per thread: a group stages HALF of a stage's weights

struct WinoArgs {
    float* out; const float* in; const float* U; const float* isc;
    int B, K, M, H, W;
};

__global__ __launch_bounds__(512, 1) void wino3x3_kernel(const WinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ubuf = smem;                                   // [2][U_FLOATS]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int grp = wid >> 2, gt = tid & 255, w4 = wid & 3;
    const int wm = w4 >> 1, wr = w4 & 1;
    float* Tl = smem + 2 * U_FLOATS + grp * T_FLOATS;     // this group's transformed input tile
    const int tiles_x = p.W / TW, tiles_y = p.H / TH, mblocks = p.M / BM;
    int t = blockIdx.x;
    const int mb = t % mblocks; t /= mblocks;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y; const int b = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH + grp * GH;      // the group's 4 rows
    const float* inb = p.in + (size_t)b * p.K * p.H * p.W;
    const float* iscb = p.isc ? p.isc + (size_t)b * p.K : nullptr;
    const bool edge = (x0 == 0) || (x0 + TW == p.W) || (y0 == 0) || (y0 + GH == p.H);      // group-uniform

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][c][r] = 0.f;

    int g_off[N_IN], l_off[N_IN];
#pragma unroll
    for (int i = 0; i < N_IN; ++i) {
        const int e = gt + 256 * i;
        const int pr = e % NP, row = (e / NP) % (GH + 2), ch = e / RS;
        g_off[i] = (ch * p.H + (y0 - 1 + row)) * p.W + x0 + 2 * pr - 1;
        l_off[i] = ch * 4 * RS + row * NP + pr;
    }
    // weights: the group stages float4 [grp * 1536, grp * 1536 + 1536) of the stage's 3072; float4 idx -> row idx >> 5 (of 96), col idx & 31
    const int w4base = grp * (U_FLOATS / 8) + gt;         // + 256 i
    const int stage_w = 3 * 4 * KC * p.M;
    const size_t stage_in = (size_t)KC * p.H * p.W;
    f32x4 rin[N_IN];
    float rsc[N_IN];
    f32x4 rw[N_W4];
    const int nstage = p.K / KC;
    auto issue_t = [&](int s) {
        const float* base = inb + s * stage_in;
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            const int e = gt + 256 * i, ch = e / RS;
            rsc[i] = iscb ? iscb[s * KC + ch] : 1.f;
            if (!edge) {
                rin[i] = *reinterpret_cast<const f32x4u*>(base + g_off[i]);
            } else {
                const int pr = e % NP, row = (e / NP) % (GH + 2);
                const int gy = y0 - 1 + row, gx = x0 + 2 * pr - 1;
                const bool rowok = gy >= 0 && gy < p.H;
#pragma unroll
                for (int q = 0; q < 4; ++q) rin[i][q] = (rowok && gx + q >= 0 && gx + q < p.W) ? base[g_off[i] + q] : 0.f;
            }
        }
    };
    auto issue_u = [&](int s) {
        const float* us = p.U + (size_t)s * stage_w + mb * BM;
#pragma unroll
        for (int i = 0; i < N_W4; ++i) {
            const int idx = w4base + 256 * i;
            rw[i] = *reinterpret_cast<const f32x4*>(us + (size_t)(idx >> 5) * p.M + 4 * (idx & 31));
        }
    };
    auto commit_t = [&]() {
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            const float sc = rsc[i];
            const float d0 = rin[i][0] * sc, d1 = rin[i][1] * sc, d2 = rin[i][2] * sc, d3 = rin[i][3] * sc;
            float* dst = Tl + l_off[i];
            dst[0] = d0 - d2;
            dst[RS] = d1 + d2;
            dst[2 * RS] = d2 - d1;
            dst[3 * RS] = d1 - d3;
        }
    };
    auto commit_u = [&](int s) {
        float* ul = Ubuf + (s & 1) * U_FLOATS;
#pragma unroll
        for (int i = 0; i < N_W4; ++i) *reinterpret_cast<f32x4*>(ul + 4 * (w4base + 256 * i)) = rw[i];
    };
    const int rr = l31 >> 4, jj = l31 & 15;
    const int b_off = half * 4 * RS + (2 * wr + rr) * NP + jj;
    const int a_off = half * BM + wm * 64 + l31;
#ifndef WAHEAD
#define WAHEAD 1          // LDS operand reads this many steps ahead of the MFMAs that consume them (a step = one (ks, ky, c): two MFMAs)
#endif
    auto mfmas = [&](int s) {
        const float* ul = Ubuf + (s & 1) * U_FLOATS + a_off;
        const float* tl = Tl + b_off;
        constexpr int NST = (KC / 2) * 12;
        float a0[WAHEAD + 1], a1[WAHEAD + 1], bb[WAHEAD + 1];
        auto rd = [&](int st, int slot) {
            const int ks = st / 12, ky = (st % 12) / 4, c = st % 4;
            bb[slot] = tl[(2 * ks * 4 + c) * RS + ky * NP];
            const float* ua = ul + ((ky * 4 + c) * KC + 2 * ks) * BM;
            a0[slot] = ua[0]; a1[slot] = ua[32];
        };
#pragma unroll
        for (int i = 0; i < WAHEAD; ++i) rd(i, i);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            if (st + WAHEAD < NST) rd(st + WAHEAD, (st + WAHEAD) % (WAHEAD + 1));
            __builtin_amdgcn_sched_barrier(0);           // keep those reads in front of the MFMAs of step st
            const int c = st % 4, slot = st % (WAHEAD + 1);
            acc[0][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[slot], bb[slot], acc[0][c], 0, 0, 0);
            acc[1][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[slot], bb[slot], acc[1][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // prologue: U_0 (each group its half), T^A_0; group B's first tile and U_1 halves follow inside the loop
    issue_u(0);
    issue_t(0);
    commit_u(0);
    if (grp == 0) commit_t();
    // group A will stage T^A_{s+1} + U_{s+1}[first half] in the second half of stage s; group B stages T^B_s + U_{s+1}[second half]
    // in the first half of stage s.  Registers: A prefetches (s+1) while it multiplies s; B already holds T^B_0 and prefetches U_1.
    if (grp == 1 && nstage > 1) issue_u(1);
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        // ---- first half: A multiplies stage s; B writes T^B_s and its half of U_{s+1}
        if (grp == 0) {
            if (s + 1 < nstage) { issue_t(s + 1); issue_u(s + 1); }
            mfmas(s);
        } else {
            commit_t();
            if (s + 1 < nstage) commit_u(s + 1);
        }
        __syncthreads();
        // ---- second half: B multiplies stage s; A writes T^A_{s+1} and its half of U_{s+1}
        if (grp == 1) {
            if (s + 1 < nstage) issue_t(s + 1);
            if (s + 2 < nstage) issue_u(s + 2);
            mfmas(s);
        } else {
            if (s + 1 < nstage) { commit_t(); commit_u(s + 1); }
        }
        __syncthreads();
    }
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
    const size_t lds = sizeof(float) * (2 * U_FLOATS + 2 * T_FLOATS);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)wino3x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int64_t blocks = (int64_t)B * (H / TH) * (W / TW) * (M / BM);
    wino3x3_kernel<<<dim3((unsigned)blocks), 512, lds, (hipStream_t)stream>>>(a);
    return (int)hipGetLastError();
}
