// F1w6: the 1-D Winograd F(2,3) form of the 3x3 / stride 1 / pad 1 convolution (wino.hip) with its multiply-adds on the BF16 matrix
// pipe and fp32-equivalent results (round 4, kind TE_CONV_3X3W6).
//
// Every fp32 operand x is split into three bf16 pieces  x = h + m + l  (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 8 + 8 + 8
// mantissa bits, exact to 2^-25 relative).  A product of two bf16 numbers is exact in fp32, so
//     a b  =  ah bh + (ah bm + am bh) + (ah bl + am bm + al bh)  +  O(2^-24 |a b|)
// is six MFMAs of v_mfma_f32_32x32x16_bf16 accumulated in fp32 - the three dropped cross terms (am bl, al bm, al bl) are below the
// rounding unit of the fp32 product.  Measured on the MI355X (tools/exp/bf16x_probe.hip, profiles/experiments/r04_bf16_split_probe.log):
// a 64 x 64 x 1152 product against double: 5.3e-7 relative (L2) for the six-product form, 6.2e-7 for the native fp32 MFMA chain - the
// split form is not "reduced precision", it is fp32 arithmetic on a different pipe - while the bf16 pipe sustains 1 490 - 1 530 TFLOP/s
// with its operands coming from LDS one 16-byte read per MFMA, i.e. 250 fp32-equivalent TFLOP/s against the 134 the fp32 matrix
// instructions reach at the clock the chip sustains.  The tests that pin the fp32 kernels (tests/test_gpu_winograd.py: 5e-6 against fp64)
// pin this one at the same bar.
//
// Data flow per stage of 16 input channels (one MFMA K step):
//   weights  : packed by te_conv_pack_weights (TE_PACK_W6FWD / TE_PACK_W6DGRAD) as U = G w, split, in MFMA FRAGMENT order
//              U6[K/16][piece][ky][component][M/32][64 lanes][8 bf16]  ->  a stage's slots are copied 16 bytes per lane to LDS
//   input    : d = style scale * in, t = B^T d (fp32, as in wino.hip), split, two channels packed per dword, written to
//              T[piece][component][row][k half][pair][8 bf16]: a wave's B operand is one conflict-free ds_read_b128 per piece
//   products : per (tap row, component): 3 + 3 operand reads, 6 MFMAs into the component's accumulator tile
// Block: 512 threads, tile 64 output channels x 8 rows x 32 columns; wave (wm, wr) = 32 channels x rows {2 wr, 2 wr + 1} x 16 pairs x
// 4 components (64 accumulator registers).  LDS: 72 KB of weights + 60 KB of transformed input per stage = one block per CU, two
// waves per SIMD.  Epilogue = wino.hip's (output transform, demodulation scale, bias, leaky ReLU, residual, mask).
// Measured (tools/wino6_check.py, batch 16): 225 / 248 / 261 / 265 TFLOP/s algorithmic at 128 -> 128 @256^2, 256 -> 256 @128^2,
// 512 -> 512 @64^2 / @32^2 against 167 / 173 / 177 / 176 of wino.hip and 134 - 140 of the direct kernel on the same box; with the
// staging compiled out the MFMA loop alone runs at 368: a stage's MFMAs and its staging do not overlap inside the one resident block
// (variants that tried - per-tap-row weight DMA, smaller double-buffered tiles on paper - lost to barriers or L2 bandwidth).
#include "conv_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifndef W6_CG
#define W6_CG 1          // components per MFMA group (1: a chain of six dependent MFMAs per group; 2: two accumulators alternate)
#endif
#ifndef W6_PIPE
#define W6_PIPE 1        // read the operands of the next group before issuing the MFMAs of the current one
#endif
constexpr int WT = 512, KC = 16, TW = 32, NP = TW / 2, BM = 64, TH = 8, R = TH + 2;
constexpr int U_CHUNKS = 36 * 2 * 64;                   // 16-byte chunks of a stage's weights: [piece 3][ky 3][c 4][mtile 2][64 lanes]
constexpr int N_UC = U_CHUNKS / WT;                     // 9 per thread
constexpr int T_DWORDS = 3 * 4 * R * 2 * NP * 4;        // [piece][c][row][k half][pair][4 dwords]
constexpr int N_ITEMS = R * NP * 8;                     // (row, pair, channel pair) items of a stage: 1280
constexpr int N_IN = (N_ITEMS + WT - 1) / WT;           // 3 per thread (the last one bounds-checked)

struct Wino6Args {
    float* out; const float* in; const u32x4* U; const float* isc; const float* osc; const float* bias; const float* res;
    const float* mref; float mgain; int act;
    int B, K, M, H, W, ntiles, mblocks, tiles_x, tiles_y, nt8;
    int lgpw;      // log2 of the column PAIRS one sample contributes to a tile row: 4 (W >= 32); 3 (W == 16: two samples side by side, round 6)
};

__device__ __forceinline__ unsigned bf16_rn(float x) {          // round to nearest even (finite inputs)
    unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// x -> (h, m, l) as 16-bit patterns
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_rn(x);
    const float r = x - __builtin_bit_cast(float, h << 16);
    m = bf16_rn(r);
    const float r2 = r - __builtin_bit_cast(float, m << 16);
    l = bf16_rn(r2);
}

__global__ __launch_bounds__(WT, 2) void wino6_kernel(const Wino6Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* ul = reinterpret_cast<u32x4*>(smem_raw);                                   // weights, 16-byte chunks
    unsigned* tl = reinterpret_cast<unsigned*>(smem_raw + U_CHUNKS * 16);             // transformed input, dwords
    const u32x4* tl4 = reinterpret_cast<const u32x4*>(smem_raw + U_CHUNKS * 16);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm = wid >> 2, wr = wid & 3;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int tq = jx / p.mblocks, mb = jx % p.mblocks;
    const int tile = p.nt8 ? (int)(((int64_t)xcd * p.ntiles) >> 3) + tq : tq * 8 + xcd;
    if (tile >= (p.nt8 ? (int)(((int64_t)(xcd + 1) * p.ntiles) >> 3) : p.ntiles)) return;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    const float* inb = p.in + (size_t)b * p.K * p.H * p.W;
    const float* iscb = p.isc ? p.isc + (size_t)b * p.K : nullptr;
    const bool edge = (x0 == 0) || (x0 + TW == p.W) || (y0 == 0) || (y0 + TH == p.H);      // block-uniform

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    // ---- staging geometry.  Input item e = tid + 512 i -> (pair jj = e % 16, channel pair q = (e / 16) % 8, row = e / 128)
    // Edge tiles read the same four floats per item from an address moved INSIDE the row / image (one float right at the left image
    // border, one left at the right border, the nearest valid row above / below) and patch the vector afterwards: no per-element
    // branches, same number of loads as the interior path.  e_flag: bit 0 left, bit 1 right, bit 2 row outside.
    int g_off[N_IN], l_off[N_IN], e_flag[N_IN];
#pragma unroll
    for (int i = 0; i < N_IN; ++i) {
        const int e = tid + WT * i;
        const int jj = e & 15, q = (e >> 4) & 7, row = e >> 7;
        const int gy = y0 - 1 + row;
        const bool left = x0 == 0 && jj == 0, right = x0 + TW == p.W && jj == NP - 1, rowout = gy < 0 || gy >= p.H;
        e_flag[i] = (left ? 1 : 0) | (right ? 2 : 0) | (rowout ? 4 : 0);
        const int gyc = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy);
        g_off[i] = (2 * q * p.H + gyc) * p.W + x0 + 2 * jj - 1 + (left ? 1 : 0) - (right ? 1 : 0);
        l_off[i] = e < N_ITEMS ? ((row * 2 + (q >> 2)) * NP + jj) * 4 + (q & 3) : -1;     // + ((piece * 4 + c) * R) * 2 * NP * 4
    }
    const size_t plane = (size_t)p.H * p.W;
    const int MT = p.M >> 5;
    f32x4 rin[N_IN][2];
    float rsc[N_IN][2];
    const int nstage = p.K / KC;
    auto issue = [&](int s) {
        const float* base = inb + (size_t)s * KC * plane;
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            if (l_off[i] < 0) continue;
            const int e = tid + WT * i, q = (e >> 4) & 7;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                rsc[i][h2] = iscb ? iscb[s * KC + 2 * q + h2] : 1.f;
                const float* src = base + g_off[i] + h2 * plane;
                rin[i][h2] = *reinterpret_cast<const f32x4u*>(src);      // (edge tiles: patched in scale(), when the load has landed)
            }
        }
    };
    // weights: chunk idx = tid + 512 r -> slot idx / 64 = (piece, ky, c, mtile), lane idx % 64; copied global -> LDS by the LDS-DMA path
    // (global_load_lds_dwordx4: a wave's 64 lanes land on 64 consecutive 16-byte chunks = one fragment slot; no staging registers, no
    // ds_write pass).  Issued when the MFMAs of a stage are done, in flight under the transform / split of the input tile.
    auto issue_u = [&](int s) {
        const u32x4* us = p.U + (size_t)s * 36 * MT * 64;
#pragma unroll
        for (int r = 0; r < N_UC; ++r) {
            const int idx = tid + WT * r, slot = idx >> 6;
            const u32x4* g = us + ((size_t)(slot >> 1) * MT + 2 * mb + (slot & 1)) * 64 + (idx & 63);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(ul + (idx & ~63)), 16, 0, 0);
        }
    };
    // commit in two halves around the weight DMA: `scale` consumes the loaded registers (the only wait on the vector-memory counter,
    // taken while no DMA is in flight - with one in flight the compiler would wait for IT at the first use of a loaded value),
    // `commit` transforms, splits and writes the input tile while the DMA runs
    f32x4 dv[N_IN][2];
    auto scale = [&]() {
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            if (l_off[i] < 0) continue;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 v = rin[i][h2];
                if (edge) {
                    const int f = e_flag[i];
                    if (f & 1) { v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = 0.f; }        // loaded from column 0: element 0 is column -1
                    if (f & 2) { v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = 0.f; }        // loaded one column early: element 3 is column W
                    if (f & 4) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                }
                dv[i][h2] = v * rsc[i][h2];
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            if (l_off[i] < 0) continue;
            const f32x4 e = dv[i][0], o = dv[i][1];                               // even / odd channel of the pair
            const f32x2 t[4] = {{e[0] - e[2], o[0] - o[2]}, {e[1] + e[2], o[1] + o[2]}, {e[2] - e[1], o[2] - o[1]}, {e[1] - e[3], o[1] - o[3]}};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // three-piece split of the channel pair: v_cvt_pk_bf16_f32 packs (even, odd) into one dword = the LDS element
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t[c], bf16x2));
                const f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xFFFF0000u)};
                const f32x2 r1 = t[c] - hf;
                const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
                const f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xFFFF0000u)};
                const f32x2 r2 = r1 - mf;
                const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
                tl[l_off[i] + (0 * 4 + c) * (R * 2 * NP * 4)] = h;
                tl[l_off[i] + (1 * 4 + c) * (R * 2 * NP * 4)] = m;
                tl[l_off[i] + (2 * 4 + c) * (R * 2 * NP * 4)] = l;
            }
        }
    };
    const int rr = l31 >> 4, jj = l31 & 15;
    const int b_chunk = ((2 * wr + rr) * 2 + half) * NP + jj;          // + ((piece * 4 + c) * R + ky) * 2 * NP       (16-byte chunks)
    const int a_chunk = wm * 64 + lane;                                 // + ((piece * 3 + ky) * 4 + c) * 128

    issue(0);
    scale();
    __builtin_amdgcn_sched_barrier(0);
    issue_u(0);
    __builtin_amdgcn_sched_barrier(0);
    commit();
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        if (s + 1 < nstage) issue(s + 1);
#ifndef W6_SKIP_MFMA
        {
            // groups of CG components of one tap row: 3 + 3 operand reads and 6 MFMAs per component.  W6_PIPE: the operands of group
            // g + 1 are read before the MFMAs of group g are issued; CG = 2 alternates two accumulators (no back-to-back dependent MFMAs)
            constexpr int CG = W6_CG, NG = 12 / CG;
            bf16x8 av[2][CG][3], bv[2][CG][3];
            auto rd = [&](int g, int slot) {
                const int ky = g / (4 / CG), c0 = (g % (4 / CG)) * CG;
#pragma unroll
                for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) {
                        av[slot][cc][pc] = __builtin_bit_cast(bf16x8, ul[a_chunk + ((pc * 3 + ky) * 4 + c0 + cc) * 128]);
                        bv[slot][cc][pc] = __builtin_bit_cast(bf16x8, tl4[b_chunk + ((pc * 4 + c0 + cc) * R + ky) * 2 * NP]);
                    }
            };
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
            if (W6_PIPE) rd(0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int slot = W6_PIPE ? (g & 1) : 0, c0 = (g % (4 / CG)) * CG;
                if (W6_PIPE) { if (g + 1 < NG) rd(g + 1, (g + 1) & 1); }
                else rd(g, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int cc = 0; cc < CG; ++cc)
                        acc[c0 + cc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][cc][PA[q]], bv[slot][cc][PB[q]], acc[c0 + cc], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        __syncthreads();
#ifndef W6_SKIP_COMMIT
        if (s + 1 < nstage) {
            scale();
            __builtin_amdgcn_sched_barrier(0);
            issue_u(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            commit();
        }
#endif
        __syncthreads();
    }
    // epilogue: output transform (two adjacent columns per accumulator element), then the direct kernel's epilogue stages
    const int mbase = mb * BM + wm * 32;
    const size_t off0 = ((size_t)b * p.M + mbase) * plane + (size_t)(y0 + 2 * wr + rr) * p.W + x0 + 2 * jj;
    const float g_pos = p.act == 3 ? 1.4142135623730951f : 1.f;
    // (the 16 demodulation scales / biases of this lane are loaded up front: inside the store loop every load would wait behind the
    //  previous row's store - the compiler cannot prove `out` does not alias them - i.e. 16 serialized L2 round trips per block)
    float scv[16], biv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * half;
        scv[r] = p.osc ? p.osc[(size_t)b * p.M + m] : 1.f;
        biv[r] = p.bias ? p.bias[m] : 0.f;
    }
    f32x2 resv[16], mrefv[16];                      // likewise the residual / mask rows of the discriminator's launches
    if (p.res) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            resv[r] = *reinterpret_cast<const f32x2*>(p.res + off0 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * plane);
    }
    if (p.mref) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            mrefv[r] = *reinterpret_cast<const f32x2*>(p.mref + off0 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * plane);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
        float v0 = acc[0][r] + acc[1][r] + acc[2][r];
        float v1 = acc[1][r] - acc[2][r] - acc[3][r];
        const float sc = scv[r], bi = biv[r];
        v0 = v0 * sc + bi;
        v1 = v1 * sc + bi;
        if (p.act >= 3) {
            v0 = (v0 > 0.f ? v0 : v0 * 0.2f) * g_pos;
            v1 = (v1 > 0.f ? v1 : v1 * 0.2f) * g_pos;
        }
        const size_t o = off0 + (size_t)dm * plane;
        if (p.res) { v0 += resv[r][0]; v1 += resv[r][1]; }
        if (p.mref) {
            v0 *= mrefv[r][0] > 0.f ? p.mgain : 0.2f * p.mgain;
            v1 *= mrefv[r][1] > 0.f ? p.mgain : 0.2f * p.mgain;
        }
        f32x2 v; v[0] = v0; v[1] = v1;
        *reinterpret_cast<f32x2*>(p.out + o) = v;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the PING-PONG form of the same kernel (wino6p_kernel).  wino6_kernel above alternates "all eight waves multiply" and "all
// eight waves stage" between two block-wide barriers, so the matrix pipe idles while the next stage is transformed, split and written
// (measured: 36 % of the kernel).  Double-buffering the whole stage does not fit (2 x 132 KB), so the tile is cut the other way:
//
//   * the 8-row tile is two HALF tiles of 4 rows (6 input rows each, 36 KB of transformed pieces per half: T0, T1); the weights of a
//     stage (72 KB) stay ONE image shared by both halves: 144 KB of LDS;
//   * waves 0-3 (group 0) own half 0, waves 4-7 (group 1) own half 1.  Waves w and w + 4 share a SIMD, so every SIMD holds one wave
//     of each group.  The groups run half a stage apart:
//         phase X(s): group 0 multiplies  T0(s) x U(s)   |  group 1 transforms / splits / writes T1(s)
//         phase Y(s): group 1 multiplies  T1(s) x U(s)   |  group 0 transforms / splits / writes T0(s + 1)
//     i.e. on every SIMD one wave feeds the matrix pipe while its partner does the vector-ALU / LDS-write / DMA work of staging
//     (the arrangement MI355X_MICROARCH.md "Two waves per SIMD" describes for attention: matrix beside memory, never matrix beside
//     matrix).  A half tile is private to its group, so the only hand-off between the groups is the weight image;
//   * the weight image is renewed in two halves without a second buffer.  A multiplying wave walks the 12 (tap row, component)
//     groups in order; the first six read Ua, the last six Ub.  Ua(s) is dead once group 1 is past the middle of Y(s): group 0 (then
//     staging) issues the DMA of Ua(s + 1) right behind that mid-phase barrier and waits for it before the end-of-phase barrier.
//     Ub(s - 1) is dead at the end of Y(s - 1): group 1 (staging in X(s)) issues Ub(s) first thing and waits before the mid-phase
//     barrier of X(s), behind which group 0 starts to read it.  Four barriers per stage (mid-X, end-X, mid-Y, end-Y), each of them a
//     point where the multiplying wave has its operands for the next six MFMAs in registers already;
//   * the mid-phase barrier sits in front of group 5's MFMAs (whose operands were read before it) and in front of the first Ub read.
// Same arithmetic, same packed weights, same epilogue and the same supported shapes as wino6_kernel; TE_W6_FORM=0 selects the old form.
constexpr int PH = 4, PR = PH + 2;
constexpr int TP_PLANE = PR * 2 * NP * 4;                 // dwords of one (piece, component) plane of a half tile
constexpr int TP_DWORDS = 12 * TP_PLANE;                  // 9 216 dwords = 36 KB
constexpr int GT = WT / 2;                                // threads of a group
constexpr int P_IN = (PR * NP * 8) / GT;                  // (row, pair, channel pair) items per thread and stage: 768 / 256 = 3
static_assert(PR * NP * 8 == P_IN * GT, "items must divide over the group");

#ifdef W6P_PROF      // experimental builds: per-wave cycle counts of the phases (s_memtime), read back with te_debug_w6p_prof
__device__ unsigned long long te_w6p_prof_buf[2048 * 8 * 8];
#define W6P_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define W6P_ACC(i, a, b) pc[i] += (b) - (a)
#else
#define W6P_T(v)
#define W6P_ACC(i, a, b)
#endif
#ifndef W6P_SLOT0
#define W6P_SLOT0 21         // MFMA slot behind which the staging arithmetic starts (72 slots, the program has 51): the fetch it
#endif                       // consumes is issued half a phase earlier by group 1
#ifndef DMA_PRIO
#define DMA_PRIO 0         // experiment: the staging wave raises its priority while it issues the weight DMA
#endif
#ifndef W6P_PRIO
#define W6P_PRIO 1           // 1: a wave raises its priority while it multiplies (+1 - 1.5 % over 0); 3: static priority for group 1 (-3 %)
#endif
__device__ __forceinline__ void w6p_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void w6p_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ISC: the launch carries style scales.  A template parameter, and the fetch of the staging role is unconditional (clamped to the last
// stage): with `if (iscb)` / `if (fetch)` around the loads the compiler kept two copies of the 26 fetch registers and moved them twice per
// staging phase - ~40 vector-ALU instructions in the wave whose instructions cost 10 - 27 cycles each (found in s2s6.hip's phase
// profile, profiles/experiments/r05_s2s6_phase_profile.log; the ISA of the staging role now has no register move at all).
// (Round 6, tried and NOT taken: a NARROW form for M == 32 - the 32 -> 32 layer at 1024^2 of the FFHQ-1024 generator - in which the block's
//  second 32-channel tile is padding, its weight slots are not fetched and the waves that own it skip MFMAs and stores.  Correct at the
//  5e-6 bar, but 1 121 us against the 769 us of the fp32 Winograd kernel wino3x3_kernel<32>: with K = 32 a block lives for two stages, one
//  block per CU (144 KB of LDS), so the prologue's HBM latency and the epilogue are not overlapped by anything;
//  profiles/experiments/r06_g1024_kernel_stats_with_wino6p_narrow.txt, r06_wino6p_narrow.patch.)
template <bool ISC>
__global__ __launch_bounds__(WT, 2) void wino6p_kernel(const Wino6Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* ul = reinterpret_cast<u32x4*>(smem_raw);                                   // weights, 16-byte chunks
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform by construction: keep the role branches scalar
    const int grp = wid >> 2, wq = wid & 3, wm = wq >> 1, wrl = wq & 1, gt = tid & (GT - 1);
    unsigned* tl = reinterpret_cast<unsigned*>(smem_raw + U_CHUNKS * 16) + grp * TP_DWORDS;      // this group's half tile
    const u32x4* tl4 = reinterpret_cast<const u32x4*>(tl);
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int tq = jx / p.mblocks, mb = jx % p.mblocks;
    const int tile = p.nt8 ? (int)(((int64_t)xcd * p.ntiles) >> 3) + tq : tq * 8 + xcd;
    if (tile >= (p.nt8 ? (int)(((int64_t)(xcd + 1) * p.ntiles) >> 3) : p.ntiles)) return;
    // W == 16 (round 6): a tile row holds TWO samples side by side - pairs 0-7 = sample b, pairs 8-15 = sample b + 1 (lgpw = 3); every
    // column pair is transformed from its own four input columns, so only the staging geometry and the output addresses know about it
    const int pwm = (1 << p.lgpw) - 1;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = (tile / (p.tiles_x * p.tiles_y)) << (4 - p.lgpw);
    const int x0 = tx * TW, y0 = ty * TH, yh = y0 + PH * grp;
    const float* inb = p.in + (size_t)b * p.K * p.H * p.W;
    const float* iscb = ISC ? p.isc + (size_t)b * p.K : nullptr;
    // block-uniform edge flags: interior tiles skip the patches of arith() by scalar branches (an empty asm statement inside each
    // branch keeps the compiler from turning them back into 96 selects per phase that every tile pays: a vector-ALU instruction in
    // the MFMA stream costs ~4.5 cycles, profiles/experiments/r05_wgrad6.log)
    const bool has_left = x0 == 0, has_right = x0 + TW >= p.W, has_rowout = (yh == 0) || (yh + PH == p.H);

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    // staging geometry of this group's half: item e = gt + 256 i -> (pair jj = e % 16, channel pair q = (e / 16) % 8, row = e / 128)
    unsigned g_off[P_IN];          // (never negative: the left-border item is moved one column to the right)
    int l_off[P_IN], e_flag[P_IN];
#pragma unroll
    for (int i = 0; i < P_IN; ++i) {
        const int e = gt + GT * i;
        const int jj = e & 15, q = (e >> 4) & 7, row = e >> 7;
        const int gy = yh - 1 + row;
        const int sj = jj >> p.lgpw, jl = jj & pwm;                       // sample of the tile row, pair inside the sample's row
        const bool left = x0 == 0 && jl == 0, right = x0 + TW >= p.W && jl == pwm, rowout = gy < 0 || gy >= p.H;
        e_flag[i] = (left ? 1 : 0) | (right ? 2 : 0) | (rowout ? 4 : 0);
        const int gyc = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy);
        g_off[i] = (unsigned)(sj * p.K * p.H * p.W + (2 * q * p.H + gyc) * p.W + x0 + 2 * jl - 1 + (left ? 1 : 0) - (right ? 1 : 0));
        l_off[i] = ((row * 2 + (q >> 2)) * NP + jj) * 4 + (q & 3);                        // + (piece * 4 + c) * TP_PLANE
    }
    const unsigned q2 = 2u * ((gt >> 4) & 7) + (unsigned)(((gt & 15) >> p.lgpw) * p.K);      // (+ the style-scale row of this thread's sample:
                                                                                              //  a thread's items share their pair index)
    const size_t plane = (size_t)p.H * p.W;
    const int MT = p.M >> 5;
    f32x4 rin[P_IN][2];
    f32x2 rsc = {1.f, 1.f};        // style scales of this thread's channel pair (the same pair for its three items: 256 % 128 == 0)
    const int nstage = p.K / KC;
    // fetch of stage s, item by item (uniform base pointer + unsigned 32-bit per-thread offset: the scalar-base addressing form, no
    // 64-bit address registers).  Inside the loop the fetch is issued by the MULTIPLYING role, each item right behind the last slot of
    // the arithmetic that reads its registers: the fetch registers are then written and read in one role only.  Issued by the staging
    // role - as the first ping-pong versions did - they came out of the register allocator as two sets with 12 - 18 64-bit moves per
    // staging phase between them (and with them conditional on `fetch`, twice that), in the wave whose vector-ALU instructions cost
    // 10 - 27 cycles each; found in s2s6.hip's phase profile, profiles/experiments/r05_s2s6_phase_profile.log.
    auto fetch_scales = [&](int s) {
        if (ISC) rsc = *reinterpret_cast<const f32x2u*>(iscb + s * KC + q2);
    };
    auto fetch_item = [&](int i, int s) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
            rin[i][h2] = *reinterpret_cast<const f32x4u*>(inb + ((size_t)s * KC + h2) * plane + g_off[i]);
    };
    auto issue = [&](int s) {
        fetch_scales(s);
#pragma unroll
        for (int i = 0; i < P_IN; ++i) fetch_item(i, s);
    };
    // weight half `uh` of stage s: 36 fragment slots (3 pieces x 6 (tap row, component) groups x 2 M tiles), 9 per wave of the group
    auto issue_u = [&](int uh, int s) {
        const u32x4* us = p.U + (size_t)s * 36 * MT * 64;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int j = wq * 9 + r, piece = j / 12, rem = j % 12, kc = (rem >> 1) + 6 * uh, mt = rem & 1;
            const int pk = piece * 12 + kc;                                   // == (piece * 3 + ky) * 4 + c
            const u32x4* g = us + ((size_t)pk * MT + 2 * mb + mt) * 64 + (unsigned)lane;      // uniform base + 32-bit lane offset
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(ul + (pk * 2 + mt) * 64), 16, 0, 0);
        }
    };
    // ---- the arithmetic of staging (style scale, B^T d, three-piece split) as a PROGRAM OF 51 SLOTS that the multiplying wave runs
    // behind its own MFMAs (one slot per MFMA; all slot indices are compile-time after unrolling).  Measured with the first
    // ping-pong version (profiles/experiments/r05_w6p_phase_profile.log): the same ~50 vector-ALU instructions per item issued by the
    // PARTNER wave of a SIMD that streams MFMAs cost 10 (older wave) to 27 (younger wave) cycles apiece - more than the multiplying
    // phase lasts - while instructions of the multiplying wave itself issue in the shadow of its own MFMAs (32 cycles each).
    //   slots 0..2         : item k: edge patch, style scale (in place in rin)
    //   slots 3 + 4 u + j  : unit u = item * 4 + component, step j:  0: t = (B^T d)_c for the channel pair, h = bf16x2(t)
    //                        1: t -= h    2: m = bf16x2(t), unpack m    3: t -= m, l = bf16x2(t)
    // The results wait in `res` (36 registers) for the staging phase, which only moves them to LDS.
    unsigned res[P_IN][4][3];
    float te = 0.f, to = 0.f, fe = 0.f, fo = 0.f;
    constexpr int N_SLOT = 3 + 4 * 4 * P_IN;
    auto arith = [&](int k) {
        if (k < 0) {
        } else if (k < P_IN) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 v = rin[k][h2];
                const int f = e_flag[k];
                if (has_left) {
                    asm volatile("" ::: "memory");
                    if (f & 1) { v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = 0.f; }        // loaded from column 0: element 0 is column -1
                }
                if (has_right) {
                    asm volatile("" ::: "memory");
                    if (f & 2) { v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = 0.f; }        // loaded one column early: element 3 is column W
                }
                if (has_rowout) {
                    asm volatile("" ::: "memory");
                    if (f & 4) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                }
                rin[k][h2] = ISC ? v * rsc[h2] : v;
                asm volatile("" : "+v"(rin[k][h2]));
            }
        } else if (k < N_SLOT) {
            const int u = (k - P_IN) >> 2, j = (k - P_IN) & 3, i = u >> 2, c = u & 3;
            if (j == 0) {
                const f32x4 e = rin[i][0], o = rin[i][1];                         // even / odd channel of the pair
                te = c == 0 ? e[0] - e[2] : (c == 1 ? e[1] + e[2] : (c == 2 ? e[2] - e[1] : e[1] - e[3]));
                to = c == 0 ? o[0] - o[2] : (c == 1 ? o[1] + o[2] : (c == 2 ? o[2] - o[1] : o[1] - o[3]));
                const f32x2 t = {te, to};
                // v_cvt_pk_bf16_f32 packs (even, odd) into one dword = the LDS element
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                res[i][c][0] = h;
                fe = __builtin_bit_cast(float, h << 16);
                fo = __builtin_bit_cast(float, h & 0xFFFF0000u);
            } else if (j == 1) {
                te -= fe; to -= fo;
            } else if (j == 2) {
                const f32x2 t = {te, to};
                const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                res[i][c][1] = m;
                fe = __builtin_bit_cast(float, m << 16);
                fo = __builtin_bit_cast(float, m & 0xFFFF0000u);
            } else {
                te -= fe; to -= fo;
                const f32x2 t = {te, to};
                res[i][c][2] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                asm volatile("" : "+v"(res[i][c][2]));
            }
            // (pin the step HERE: the values have no use before the staging phase, and the compiler otherwise sinks the whole
            //  program behind the mid-phase barrier)
            asm volatile("" : "+v"(te), "+v"(to), "+v"(fe), "+v"(fo));
        }
    };
    auto write_res = [&]() {
#pragma unroll
        for (int i = 0; i < P_IN; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) tl[l_off[i] + (pc * 4 + c) * TP_PLANE] = res[i][c][pc];
    };
    const int rr = l31 >> 4, jj = l31 & 15;
    const int b_chunk = ((2 * wrl + rr) * 2 + half) * NP + jj;          // + ((piece * 4 + c) * PR + ky) * 2 * NP       (16-byte chunks)
    const int a_chunk = wm * 64 + lane;                                 // + ((piece * 3 + ky) * 4 + c) * 128

    // prologue: every group transforms and writes its half of stage 0 and fetches stage 1; group 0 brings in the whole weight image
    issue(0);
    if (grp == 0) { issue_u(0, 0); issue_u(1, 0); }
#pragma unroll
    for (int k = 0; k < N_SLOT; ++k) arith(k);
    write_res();
    issue(1);
    // (only the weight DMA has to have landed at the barrier: a COUNTED wait in group 0 leaves the fetch of stage 1 - 7 loads, 6 without
    //  style scales - in flight; waiting for it too cost every block one HBM latency, ~2 of the ~8 us a block spends outside its stages)
    if (grp == 0) {
        if (ISC) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    w6p_barrier();
    const int nphase = 2 * nstage;
    if (W6P_PRIO == 3 && grp == 1) __builtin_amdgcn_s_setprio(1);
#ifdef W6P_PROF
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long pstart = __builtin_readcyclecounter(), rstart = __builtin_amdgcn_s_memrealtime();
#endif
    for (int ph = 0; ph < nphase; ++ph) {
        const bool last = ph == nphase - 1;
        W6P_T(t0);
        if ((ph & 1) == grp) {
            // ---- multiply this group's half of stage ph / 2; behind the MFMAs: the arithmetic of stage ph / 2 + 1 (rin -> res)
            // (in a group's last multiplying phase the arithmetic runs on the stale registers of the last fetch and its results are never
            //  written: ONE copy of the MFMA stream instead of two with their own register allocation and 64 moves between them)
            bf16x8 av[2][3], bv[2][3];
            const int fs2 = min((ph >> 1) + 2, nstage - 1);
            auto rd1 = [&](int g, int slot, int q) {
                const int ky = g >> 2, c = g & 3;
                if (q < 3) av[slot][q] = __builtin_bit_cast(bf16x8, ul[a_chunk + ((q * 3 + ky) * 4 + c) * 128]);
                else bv[slot][q - 3] = __builtin_bit_cast(bf16x8, tl4[b_chunk + (((q - 3) * 4 + c) * PR + ky) * 2 * NP]);
            };
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
            for (int q = 0; q < 6; ++q) rd1(0, 0, q);
            if (W6P_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                const int slot = g & 1, c = g & 3;
#ifdef W6P_PROF
                if (g == 5) { W6P_T(ta); w6p_barrier(); W6P_T(tb); W6P_ACC(0, t0, ta); W6P_ACC(1, ta, tb); pc[2] -= tb; }
#else
                if (g == 5) w6p_barrier();             // mid-phase barrier (every phase, the last one too: no branch in the MFMA stream)
#endif
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
#ifndef W6_SKIP_MFMA
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][PA[q]], bv[slot][PB[q]], acc[c], 0, 0, 0);
#endif
                    if (g + 1 < 12 && q < 3) { rd1(g + 1, slot ^ 1, 2 * q); rd1(g + 1, slot ^ 1, 2 * q + 1); }
#ifndef W6_SKIP_COMMIT
                    arith(g * 6 + q - W6P_SLOT0);
#endif
                    {   // the fetch of the stage after next, item by item behind the last slot that reads the item's registers
                        constexpr int LASTQ = 71 - W6P_SLOT0;                      // slot of the last MFMA
                        const int k = g * 6 + q - W6P_SLOT0;
                        if (k == P_IN - 1) fetch_scales(fs2);
#pragma unroll
                        for (int i = 0; i < P_IN; ++i)
                            if (k == (P_IN + 16 * i + 12 < LASTQ ? P_IN + 16 * i + 12 : LASTQ)) fetch_item(i, fs2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (W6P_PRIO == 1) __builtin_amdgcn_s_setprio(0);
#ifdef W6P_PROF
            { asm volatile("s_nop 0" ::: "memory"); W6P_T(tc); pc[2] += tc; }
#endif
        } else {
            // ---- stage: move this group's half of stage cs = (ph + 1) / 2 to LDS, fetch stage cs + 1, renew a half of the weight image
            const int cs = (ph + 1) >> 1;
            const bool work = cs >= 1 && cs < nstage;
            // group 1 renews Ub in front of the mid-phase barrier (the partner reads it right behind): DMA first, the LDS writes of
            // this half tile in its shadow, then the wait for the DMA; group 0 renews Ua behind the barrier.  (The fetch of the next
            // stage is not issued here any more: see fetch_item.)  The LDS writes are unconditional: in group 1's first phase they
            // repeat what the prologue wrote, in group 0's last phase they put stale results in a tile nobody reads any more.
            if (DMA_PRIO) __builtin_amdgcn_s_setprio(DMA_PRIO);
            if (grp == 1 && work) issue_u(1, cs);
            if (DMA_PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#ifndef W6_SKIP_COMMIT
            write_res();
#endif
            if (grp == 1 && work) w6p_wait_vm();
            W6P_T(ta);
            w6p_barrier();
            W6P_T(tb);
            if (grp == 0 && work) {
                if (DMA_PRIO) __builtin_amdgcn_s_setprio(DMA_PRIO);
                issue_u(0, cs);
                if (DMA_PRIO) __builtin_amdgcn_s_setprio(0);
                w6p_wait_vm();
            }
            W6P_T(tc);
            W6P_ACC(3, t0, ta); W6P_ACC(4, ta, tb); W6P_ACC(5, tb, tc);
        }
        W6P_T(t8);
        if (!last) w6p_barrier();
        W6P_T(t9);
        W6P_ACC(6, t8, t9);
    }
#ifdef W6P_PROF
    if (lane == 0 && blockIdx.x < 2048) {
        unsigned long long* d = te_w6p_prof_buf + ((size_t)blockIdx.x * 8 + wid) * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = pc[i];
        d[6] = pc[6] | ((__builtin_amdgcn_s_memrealtime() - rstart) << 40);          // (100 MHz counter: the shader clock follows)
        d[7] = ((unsigned long long)nstage << 48) | ((__builtin_readcyclecounter() - pstart) & 0xFFFFFFFFFFFFull);
    }
#endif
    // epilogue: as wino6_kernel (output transform, demodulation scale, bias, leaky ReLU, residual, mask); group 0 is here one phase early
    const int wr = grp * 2 + wrl;
    const int mbase = mb * BM + wm * 32;
    const int bo = b + (jj >> p.lgpw);                                  // this lane's output sample
    const size_t off0 = ((size_t)bo * p.M + mbase) * plane + (size_t)(y0 + 2 * wr + rr) * p.W + x0 + 2 * (jj & pwm);
    const float g_pos = p.act == 3 ? 1.4142135623730951f : 1.f;
    float scv[16], biv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * half;
        scv[r] = p.osc ? p.osc[(size_t)bo * p.M + m] : 1.f;
        biv[r] = p.bias ? p.bias[m] : 0.f;
    }
    f32x2 resv[16], mrefv[16];
    if (p.res) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            resv[r] = *reinterpret_cast<const f32x2*>(p.res + off0 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * plane);
    }
    if (p.mref) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            mrefv[r] = *reinterpret_cast<const f32x2*>(p.mref + off0 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * plane);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
        float v0 = acc[0][r] + acc[1][r] + acc[2][r];
        float v1 = acc[1][r] - acc[2][r] - acc[3][r];
        const float sc = scv[r], bi = biv[r];
        v0 = v0 * sc + bi;
        v1 = v1 * sc + bi;
        if (p.act >= 3) {
            v0 = (v0 > 0.f ? v0 : v0 * 0.2f) * g_pos;
            v1 = (v1 > 0.f ? v1 : v1 * 0.2f) * g_pos;
        }
        const size_t o = off0 + (size_t)dm * plane;
        if (p.res) { v0 += resv[r][0]; v1 += resv[r][1]; }
        if (p.mref) {
            v0 *= mrefv[r][0] > 0.f ? p.mgain : 0.2f * p.mgain;
            v1 *= mrefv[r][1] > 0.f ? p.mgain : 0.2f * p.mgain;
        }
        f32x2 v; v[0] = v0; v[1] = v1;
        *reinterpret_cast<f32x2*>(p.out + o) = v;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: TWO 64-channel weight images per staged half tile (wino6q_kernel, form 2, taken when M % 128 == 0).
//
// What the round-6 power / clock telemetry says about wino6p_kernel (profiles/r06_power_clock_*.txt, 3-second loops, 1 400 W cap): the
// product kernel sits AT the cap (1 364 - 1 389 W, PPT residency 60 - 75 % of the samples) at 1.99 - 2.04 GHz; with the staging arithmetic
// and its LDS writes compiled out the same MFMA stream runs 21 - 26 % faster at the same power (0.96 - 1.10 pJ per executed FLOP against
// 1.23 - 1.40), and the staging alone (MFMAs compiled out) needs MORE shader cycles than the MFMAs alone (2.16 M against 1.92 M kilocycles
// at 128 -> 128 @256^2).  The block tile of wino6p is 64 output channels, so the style scale, B^T d and the three-piece split of every
// input element are re-done by every M block: twice at 128 channels, four times at 256, eight times at 512.  Here a block owns 128
// output channels: each half tile T_g(s) is staged ONCE and multiplied by the two weight images (s, m = 0) and (s, m = 1) in two
// multiplying phases with their own accumulators (2 x 64 registers) - half the staging instructions, LDS writes and fetches per MFMA, same LDS
// budget (one 72 KB weight image, two 36 KB half tiles), same weight-image hand-over protocol with the "stage" index replaced by the
// image index j = 2 s + m:
//     phase p (0 .. 4 nstage - 1): group p & 1 multiplies with image j = p >> 1; the other group is in its staging role
//     group 0:            M0(s)  SA  M1(s)  SB          group 1:   S  M0(s)  SA  M1(s)  SB   (one phase behind)
//     M0: 72 MFMAs + the fetch of stage s + 1 (consumed one multiplying phase later)    SA: weight DMA only
//     M1: 72 MFMAs + the staging arithmetic (rin -> res) behind them                       SB: weight DMA + res -> T_g(s + 1)
// Both groups run the SAME straight-line loop body (M0 SA M1 SB), group 1 one staging phase late: no role branch around the MFMA streams, two
// copies of the stream (one per accumulator set).  Same products in the same order per output element as wino6p / wino6: bit-identical.
template <bool ISC>
__global__ __launch_bounds__(WT, 2) void wino6q_kernel(const Wino6Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* ul = reinterpret_cast<u32x4*>(smem_raw);                                   // weights, 16-byte chunks
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wq = wid & 3, wm = wq >> 1, wrl = wq & 1, gt = tid & (GT - 1);
    unsigned* tl = reinterpret_cast<unsigned*>(smem_raw + U_CHUNKS * 16) + grp * TP_DWORDS;      // this group's half tile
    const u32x4* tl4 = reinterpret_cast<const u32x4*>(tl);
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int tq = jx / p.mblocks, mbq = jx % p.mblocks;               // (mblocks = M / 128 for this form)
    const int tile = p.nt8 ? (int)(((int64_t)xcd * p.ntiles) >> 3) + tq : tq * 8 + xcd;
    if (tile >= (p.nt8 ? (int)(((int64_t)(xcd + 1) * p.ntiles) >> 3) : p.ntiles)) return;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx * TW, y0 = ty * TH, yh = y0 + PH * grp;
    const float* inb = p.in + (size_t)b * p.K * p.H * p.W;
    const float* iscb = ISC ? p.isc + (size_t)b * p.K : nullptr;
    const bool has_left = x0 == 0, has_right = x0 + TW == p.W, has_rowout = (yh == 0) || (yh + PH == p.H);

    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][c][r] = 0.f;

    unsigned g_off[P_IN];
    int l_off[P_IN], e_flag[P_IN];
#pragma unroll
    for (int i = 0; i < P_IN; ++i) {
        const int e = gt + GT * i;
        const int jj = e & 15, q = (e >> 4) & 7, row = e >> 7;
        const int gy = yh - 1 + row;
        const bool left = x0 == 0 && jj == 0, right = x0 + TW == p.W && jj == NP - 1, rowout = gy < 0 || gy >= p.H;
        e_flag[i] = (left ? 1 : 0) | (right ? 2 : 0) | (rowout ? 4 : 0);
        const int gyc = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy);
        g_off[i] = (unsigned)((2 * q * p.H + gyc) * p.W + x0 + 2 * jj - 1 + (left ? 1 : 0) - (right ? 1 : 0));
        l_off[i] = ((row * 2 + (q >> 2)) * NP + jj) * 4 + (q & 3);                        // + (piece * 4 + c) * TP_PLANE
    }
    const unsigned q2 = 2u * ((gt >> 4) & 7);
    const size_t plane = (size_t)p.H * p.W;
    const int MT = p.M >> 5;
    f32x4 rin[P_IN][2];
    f32x2 rsc = {1.f, 1.f};
    const int nstage = p.K / KC, nimg = 2 * nstage;
    auto fetch_scales = [&](int s) {
        if (ISC) rsc = *reinterpret_cast<const f32x2u*>(iscb + s * KC + q2);
    };
    auto fetch_item = [&](int i, int s) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
            rin[i][h2] = *reinterpret_cast<const f32x4u*>(inb + ((size_t)s * KC + h2) * plane + g_off[i]);
    };
    // weight half `uh` of image j = 2 s + m: 36 fragment slots (3 pieces x 6 (tap row, component) groups x 2 M tiles), 9 per wave
    auto issue_u = [&](int uh, int j) {
        const u32x4* us = p.U + (size_t)(j >> 1) * 36 * MT * 64;
        const int mb = 2 * mbq + (j & 1);
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int jw = wq * 9 + r, piece = jw / 12, rem = jw % 12, kc = (rem >> 1) + 6 * uh, mt = rem & 1;
            const int pk = piece * 12 + kc;                                   // == (piece * 3 + ky) * 4 + c
            const u32x4* g = us + ((size_t)pk * MT + 2 * mb + mt) * 64 + (unsigned)lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(ul + (pk * 2 + mt) * 64), 16, 0, 0);
        }
    };
    // the staging arithmetic as a program of 51 slots (wino6p_kernel: arith), run behind the MFMAs of the m = 1 phase
    unsigned res[P_IN][4][3];
    float te = 0.f, to = 0.f, fe = 0.f, fo = 0.f;
    constexpr int N_SLOT = 3 + 4 * 4 * P_IN;
    auto arith = [&](int k) {
        if (k < 0) {
        } else if (k < P_IN) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 v = rin[k][h2];
                // (the flag is made opaque INSIDE each branch: otherwise the nine lane masks (f & bit) != 0 are hoisted out of the loop
                //  as 18 scalar registers, which - with two accumulator sets - spill to v_writelane / v_readlane pairs in the MFMA stream)
                if (has_left) {
                    int f = e_flag[k];
                    asm volatile("" : "+v"(f) :: "memory");
                    if (f & 1) { v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = 0.f; }
                }
                if (has_right) {
                    int f = e_flag[k];
                    asm volatile("" : "+v"(f) :: "memory");
                    if (f & 2) { v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = 0.f; }
                }
                if (has_rowout) {
                    int f = e_flag[k];
                    asm volatile("" : "+v"(f) :: "memory");
                    if (f & 4) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                }
                rin[k][h2] = ISC ? v * rsc[h2] : v;
                asm volatile("" : "+v"(rin[k][h2]));
            }
        } else if (k < N_SLOT) {
            const int u = (k - P_IN) >> 2, j = (k - P_IN) & 3, i = u >> 2, c = u & 3;
            if (j == 0) {
                const f32x4 e = rin[i][0], o = rin[i][1];
                te = c == 0 ? e[0] - e[2] : (c == 1 ? e[1] + e[2] : (c == 2 ? e[2] - e[1] : e[1] - e[3]));
                to = c == 0 ? o[0] - o[2] : (c == 1 ? o[1] + o[2] : (c == 2 ? o[2] - o[1] : o[1] - o[3]));
                const f32x2 t = {te, to};
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                res[i][c][0] = h;
                fe = __builtin_bit_cast(float, h << 16);
                fo = __builtin_bit_cast(float, h & 0xFFFF0000u);
            } else if (j == 1) {
                te -= fe; to -= fo;
            } else if (j == 2) {
                const f32x2 t = {te, to};
                const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                res[i][c][1] = m;
                fe = __builtin_bit_cast(float, m << 16);
                fo = __builtin_bit_cast(float, m & 0xFFFF0000u);
            } else {
                te -= fe; to -= fo;
                const f32x2 t = {te, to};
                res[i][c][2] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                asm volatile("" : "+v"(res[i][c][2]));
            }
            asm volatile("" : "+v"(te), "+v"(to), "+v"(fe), "+v"(fo));
        }
    };
    auto write_res = [&]() {
#pragma unroll
        for (int i = 0; i < P_IN; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) tl[l_off[i] + (pc * 4 + c) * TP_PLANE] = res[i][c][pc];
    };
    const int rr = l31 >> 4, jj = l31 & 15;
    const int b_chunk = ((2 * wrl + rr) * 2 + half) * NP + jj;
    const int a_chunk = wm * 64 + lane;

    // one multiplying phase with accumulator set MSET; behind the MFMAs: MSET 0 the fetch of stage `fs`, MSET 1 the staging arithmetic
    auto multiply = [&](auto mset_tag, int fs) {
        constexpr int MSET = decltype(mset_tag)::value;
        bf16x8 av[2][3], bv[2][3];
        auto rd1 = [&](int g, int slot, int q) {
            const int ky = g >> 2, c = g & 3;
            if (q < 3) av[slot][q] = __builtin_bit_cast(bf16x8, ul[a_chunk + ((q * 3 + ky) * 4 + c) * 128]);
            else bv[slot][q - 3] = __builtin_bit_cast(bf16x8, tl4[b_chunk + (((q - 3) * 4 + c) * PR + ky) * 2 * NP]);
        };
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
        for (int q = 0; q < 6; ++q) rd1(0, 0, q);
        if (W6P_PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const int slot = g & 1, c = g & 3;
            if (g == 5) w6p_barrier();             // mid-phase barrier
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
#ifndef W6_SKIP_MFMA
                acc[MSET][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][PA[q]], bv[slot][PB[q]], acc[MSET][c], 0, 0, 0);
#endif
                if (g + 1 < 12 && q < 3) { rd1(g + 1, slot ^ 1, 2 * q); rd1(g + 1, slot ^ 1, 2 * q + 1); }
                const int k = g * 6 + q;
                if (MSET == 0) {
                    // the fetch of the next stage, an item every twelve slots from the sixth on (rin is free: the m = 1 phase consumed it)
                    if (k == 5) fetch_scales(fs);
#pragma unroll
                    for (int i = 0; i < P_IN; ++i)
                        if (k == 6 + 12 * i) fetch_item(i, fs);
                } else {
#ifndef W6_SKIP_COMMIT
                    arith(k - W6P_SLOT0);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (W6P_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        w6p_barrier();                             // end of phase
    };
    // one phase in the staging role.  p = global phase index; image cs = (p + 1) >> 1 is the one whose half this phase renews:
    // group 1 (even p) renews Ub(cs) in FRONT of the mid-phase barrier (its partner reads it right behind), group 0 (odd p) renews
    // Ua(cs) BEHIND it (the partner is past the last read of Ua(cs - 1) there).  WRITE: this is the phase behind the group's m = 1
    // multiply: the parked results go to the half tile (nobody reads it until the group's next m = 0 phase).
    auto stage = [&](int ph, bool write) {
        const int cs = (ph + 1) >> 1;
        const bool work = cs >= 1 && cs < nimg;
        if (grp == 1 && work) issue_u(1, cs);
        __builtin_amdgcn_sched_barrier(0);
#ifndef W6_SKIP_COMMIT
        if (write) write_res();
#endif
        if (grp == 1 && work) w6p_wait_vm();
        w6p_barrier();                             // mid-phase
        if (grp == 0 && work) {
            issue_u(0, cs);
            w6p_wait_vm();
        }
        w6p_barrier();                             // end of phase
    };

    // prologue: every group transforms and writes its half of stage 0; group 0 brings in the whole weight image 0
    fetch_scales(0);
#pragma unroll
    for (int i = 0; i < P_IN; ++i) fetch_item(i, 0);
    if (grp == 0) { issue_u(0, 0); issue_u(1, 0); }
#pragma unroll
    for (int k = 0; k < N_SLOT; ++k) arith(k);
    write_res();
    w6p_wait_vm();
    w6p_barrier();
    int ph = 0;
    if (grp == 1) { stage(0, false); ph = 1; }
    for (int s = 0; s < nstage; ++s) {
        const int fs = min(s + 1, nstage - 1);
        multiply(std::integral_constant<int, 0>{}, fs);
        stage(ph + 1, false);
        multiply(std::integral_constant<int, 1>{}, fs);
        if (!(grp == 1 && s == nstage - 1)) stage(ph + 3, true);
        ph += 4;
    }

    // epilogue: as wino6_kernel, once per accumulator set
    const int wr = grp * 2 + wrl;
    const float g_pos = p.act == 3 ? 1.4142135623730951f : 1.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int mbase = (2 * mbq + m) * BM + wm * 32;
        const size_t off0 = ((size_t)b * p.M + mbase) * plane + (size_t)(y0 + 2 * wr + rr) * p.W + x0 + 2 * jj;
        float scv[16], biv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = mbase + (r & 3) + 8 * (r >> 2) + 4 * half;
            scv[r] = p.osc ? p.osc[(size_t)b * p.M + mm] : 1.f;
            biv[r] = p.bias ? p.bias[mm] : 0.f;
        }
        f32x2 resv[16], mrefv[16];
        if (p.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                resv[r] = *reinterpret_cast<const f32x2*>(p.res + off0 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * plane);
        }
        if (p.mref) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mrefv[r] = *reinterpret_cast<const f32x2*>(p.mref + off0 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * plane);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
            float v0 = acc[m][0][r] + acc[m][1][r] + acc[m][2][r];
            float v1 = acc[m][1][r] - acc[m][2][r] - acc[m][3][r];
            const float sc = scv[r], bi = biv[r];
            v0 = v0 * sc + bi;
            v1 = v1 * sc + bi;
            if (p.act >= 3) {
                v0 = (v0 > 0.f ? v0 : v0 * 0.2f) * g_pos;
                v1 = (v1 > 0.f ? v1 : v1 * 0.2f) * g_pos;
            }
            const size_t o = off0 + (size_t)dm * plane;
            if (p.res) { v0 += resv[r][0]; v1 += resv[r][1]; }
            if (p.mref) {
                v0 *= mrefv[r][0] > 0.f ? p.mgain : 0.2f * p.mgain;
                v1 *= mrefv[r][1] > 0.f ? p.mgain : 0.2f * p.mgain;
            }
            f32x2 v; v[0] = v0; v[1] = v1;
            *reinterpret_cast<f32x2*>(p.out + o) = v;
        }
    }
}

}  // namespace

// kernel form of TE_CONV_3X3W6: 1 = ping-pong (wino6p_kernel, round 5, default), 0 = block-phase (wino6_kernel, round 4); same results
// bit for bit (same products, same accumulation order per output element).  TE_W6_FORM in the environment sets the initial value.
// 2 (round 6, default) = the two-image form wino6q_kernel where M % 128 == 0 and the grid still gives every CU a block (it has half
// as many blocks as the ping-pong form: measured slower on small grids, e.g. 39 against 22 us at 3 x 64 -> 256 @16x64), the ping-pong
// form elsewhere; 3 = the two-image form wherever M % 128 == 0 (tests).
static std::atomic<int> g_w6_form{[] { const char* e = getenv("TE_W6_FORM"); return e ? atoi(e) : 2; }()};
extern "C" int te_conv_wino6_form(int form) {
    const int old = g_w6_form.load(std::memory_order_relaxed);
    if (form >= 0 && form <= 3) g_w6_form.store(form, std::memory_order_relaxed);
    return old;
}

#ifdef W6P_PROF
extern "C" int te_debug_w6p_prof(void* host_dst, int64_t bytes) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(te_w6p_prof_buf), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int te_conv_wino6_supported(int B, int K, int M, int H, int W) {
    if (!(B > 0 && K >= 32 && K % 32 == 0 && M >= BM && M % BM == 0 && H >= TH && H % TH == 0)) return 0;
    // W % 32 == 0, or (round 6) W == 16 with an even batch: two samples side by side in a 32-column tile row - the 16 x 16 layers of both
    // networks (512 channels: 32 stages per block) leave the fp32 pipe; 8 x 8 and 4 x 4 images give too few blocks and stay there
    const bool wide = W >= TW && W % TW == 0, pair16 = W == 16 && B % 2 == 0;
    if (!wide && !pair16) return 0;
    const int64_t tiles = wide ? (int64_t)B * (H / TH) * (W / TW) : (int64_t)(B / 2) * (H / TH);
    // (16-column images: only from half a block per CU up - measured at 512 -> 512 @16x16: 163 against 357 us at batch 32, 123 against 166 at
    //  batch 16, but 111 against 92 at batch 8, where the direct kernel's split over K fills the chip better than 64 blocks do)
    if (!wide && tiles * (M / BM) < te::kNumCU / 2) return 0;
    return ((int64_t)K * H * W * 4 * (wide ? 1 : 2) < 0x7FFFFFFF && tiles * (M / BM) < 0x7FFFFFF0) ? 1 : 0;
}

int te_wino6_launch(float* out, const float* in, const float* U, const float* isc, const float* osc, const float* bias, const float* res,
                    const float* mask_ref, float mask_gain, int act, int B, int K, int M, int H, int W, hipStream_t s) {
    TE_REQUIRE(te_conv_wino6_supported(B, K, M, H, W), TE_ERR_UNSUPPORTED,
               "te_conv_f32(TE_CONV_3X3W6): needs K %% 32 == 0, M %% 64 == 0, W %% 32 == 0 (or W == 16 and an even batch), H %% 8 == 0 (te_conv_wino6_supported)");
    TE_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(res) |
                 reinterpret_cast<uintptr_t>(mask_ref)) & 15) == 0 && (reinterpret_cast<uintptr_t>(in) & 3) == 0, TE_ERR_UNSUPPORTED,
               "te_conv_f32(TE_CONV_3X3W6): 16-byte aligned tensors required");
    Wino6Args a{};
    a.out = out; a.in = in; a.U = reinterpret_cast<const u32x4*>(U); a.isc = isc; a.osc = osc; a.bias = bias; a.res = res;
    a.mref = mask_ref; a.mgain = mask_gain; a.act = act;
    a.B = B; a.K = K; a.M = M; a.H = H; a.W = W;
    a.lgpw = W >= TW ? 4 : 3;
    a.tiles_x = W >= TW ? W / TW : 1; a.tiles_y = H / TH; a.mblocks = M / BM;
    a.ntiles = (W >= TW ? B : B / 2) * a.tiles_x * a.tiles_y;
    a.nt8 = te::xcd_banded() ? (int)te::cdiv(a.ntiles, 8) : 0;
    const int64_t blocks = te::cdiv(a.ntiles, 8) * 8 * a.mblocks;
    const int form = g_w6_form.load(std::memory_order_relaxed);
    const int64_t blocks_q = te::cdiv(a.ntiles, 8) * 8 * (M / (2 * BM));
    if (form >= 2 && M % (2 * BM) == 0 && W >= TW && (form == 3 || blocks_q >= te::kNumCU)) {     // (W == 16: the ping-pong kernel only)
        a.mblocks = M / (2 * BM);
        const int64_t blocks2 = blocks_q;
        const size_t lds = (size_t)U_CHUNKS * 16 + 2 * (size_t)TP_DWORDS * 4;
        static std::atomic<uint64_t> attr_done_q{0}, attr_done_qs{0};
        if (isc) {
            te::allow_big_lds(attr_done_qs, (const void*)wino6q_kernel<true>, 160 * 1024);
            wino6q_kernel<true><<<dim3((unsigned)blocks2), WT, lds, s>>>(a);
        } else {
            te::allow_big_lds(attr_done_q, (const void*)wino6q_kernel<false>, 160 * 1024);
            wino6q_kernel<false><<<dim3((unsigned)blocks2), WT, lds, s>>>(a);
        }
    } else if (form >= 1 || W < TW) {          // (the side-by-side form of 16-column images exists in the ping-pong / two-image kernels only)
        const size_t lds = (size_t)U_CHUNKS * 16 + 2 * (size_t)TP_DWORDS * 4;
        static std::atomic<uint64_t> attr_done_p{0}, attr_done_ps{0};
        if (isc) {
            te::allow_big_lds(attr_done_ps, (const void*)wino6p_kernel<true>, 160 * 1024);
            wino6p_kernel<true><<<dim3((unsigned)blocks), WT, lds, s>>>(a);
        } else {
            te::allow_big_lds(attr_done_p, (const void*)wino6p_kernel<false>, 160 * 1024);
            wino6p_kernel<false><<<dim3((unsigned)blocks), WT, lds, s>>>(a);
        }
    } else {
        const size_t lds = (size_t)U_CHUNKS * 16 + (size_t)T_DWORDS * 4;
        static std::atomic<uint64_t> attr_done{0};
        te::allow_big_lds(attr_done, (const void*)wino6_kernel, 160 * 1024);
        wino6_kernel<<<dim3((unsigned)blocks), WT, lds, s>>>(a);
    }
    return te::launch_status("te_conv_f32(TE_CONV_3X3W6)");
}
