// Weight-gradient correlation of the 3x3 / stride 1 / pad 1 convolutions on the bf16 matrix pipe (round 5): the same slabs as
// wgrad_mfma_kernel<TE_CONV_3X3, 4, true> of wgrad.hip -
//     slab[b][s][co][ci][ky][kx] = sum over the cells (y, x) of chunk s of sample b   g[b, co, y, x] * x[b, ci, y + ky - 1, x + kx - 1]
// (reference: the autograd of F.conv2d(groups = B) in ModulatedConv2d.forward, model_spatial_query.py:318-333; the per-sample scales are
// applied by te_wgrad_reduce_f32, which is unchanged) - with every fp32 operand split into three bf16 pieces and six exact piece
// products per multiply-add accumulated in fp32, as wino6.hip / s2s6.hip / t2s6.hip do for the forward and data-gradient passes
// (same split, same product order "small terms first": fp32-equivalent results, tests/test_gpu_wgrad6.py).
//
// Pair form F(3,2) along x, as the fp32 kernel: for the pair (g0, g1) of the output gradient at columns (2p, 2p + 1) and the four
// inputs e0..e3 at columns 2p - 1 .. 2p + 2 of one tap row,
//     u = [g0, g0 + g1, g1 - g0, -g1]     t = [e0 - e2, e1 + e2, e2 - e1, e1 - e3]     m_j = sum over pairs u_j t_j
//     dW[ky][0] = m0 + (m1 - m2) / 2      dW[ky][1] = (m1 + m2) / 2                   dW[ky][2] = (m1 - m2) / 2 + m3
// (u2 carries the opposite sign of wgrad.hip's so that u1, u2 and t1, t2 are the SAME expressions of the staged registers in both
// staging roles).  The reduction runs over PAIRS: one v_mfma_f32_32x32x16_bf16 consumes the 16 pairs of a 32-cell row segment, and a
// step (one row segment) is 4 components x 3 tap rows x 6 piece products = 72 MFMAs per wave into twelve 32 x 32 accumulators.
//
// Block = 256 threads = 4 waves, ONE PER SIMD (192 accumulator registers + operands need the 512-register budget), tile 64 output x 64
// input channels, wave (wco, wci) owns 32 x 32 of it.  A block walks the steps (sample of the group, 32-column tile, row) of its chunk
// in sweeps down a column tile; the transformed and split operands live in LDS as six 24 KB images
//     [piece 3][component 4][k half 2][channel 64][8 bf16]        (one conflict-free ds_read_b128 per MFMA operand)
// - two for the gradient rows (double buffer) and a ring of four for the input rows (a step reads rows y - 1, y, y + 1 and row y + 2 is
// being written), so an input row is transformed once per sweep, not once per tap row.  There is one role per wave and no idle
// phase: while the 72 MFMAs of step y run, the wave's own vector ALU transforms and splits its share of the NEXT rows behind them
// (waves 0-1: gradient row y + 1, waves 2-3: input row y + 2; a lane = 16 cells of one channel = whole 16-byte LDS elements) from
// registers loaded during step y - 1, and the loads of the rows after that are issued at the head of the step: every global load
// has a whole step (>= 2 300 cycles) to land.  One workgroup barrier per step.
#include "conv_common.h"
#include <atomic>

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int WT = 256;                       // 4 waves
constexpr int TC = 64;                        // channels per block, both sides
constexpr int IMG = 3 * 4 * 2 * TC;           // 16-byte elements of one operand image: 1 536 (24 KB)
constexpr int N_IMG = 6;                      // 2 gradient images + 4 input-row images
constexpr unsigned OOBW = 0x80000000u;        // buffer offset beyond num_records: the load returns 0
constexpr int NRAW = 18;                      // staged registers of a lane: columns c0 - 1 .. c0 + 16

struct Wg6Args {
    float* slabs; const float* g; const float* x;
    int B, Co, Ci, H, W, S, NB, tiles_x;
};

__device__ __forceinline__ f32x4 buf_load4w(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void w6g_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// Round 6 (operand traffic 2.49x / 3.65x the algorithmic bytes in the counters).  Two changes of ORDER, no change of work:
//  (i) the chunks of a sample group and their channel-tile siblings run on ONE XCD.  Blocks are dealt round-robin to the 8 XCDs in
//      linear order, and gridDim.x is a multiple of 8 at every model shape, so XCD = blockIdx.x % 8 whatever (y, z): virtual index
//      v = (x % 8) * (gridDim.x / 8) + x / 8 gives every XCD a CONTIGUOUS range of (group, chunk) pairs (the banded order conv.hip uses);
// (ii) the chunks that run side by side work on ADJACENT 32- / 16-column tiles at the same rows: chunk s takes the columns s, s + S,
//      s + 2 S ... (when S divides the column count) or the column s % U and the row range s / U (when a column is split), instead
//      of a contiguous run of columns.  A lane's halo loads - one cell left of its 16 cells, one cell right: two more 64-byte sectors per
//      row segment, 3 fetched for 1.1 used when the neighbouring tile is staged hundreds of steps later - and the channel-tile siblings' copies of the
//      same rows then meet in that XCD's L2 instead of going out to the fabric again.
struct Wg6Chunk { int bgrp, s, nseg, mode, U; };
__device__ __forceinline__ Wg6Chunk wg6_chunk(int S, int U) {
    const int total = (int)gridDim.x, x = (int)blockIdx.x;
    const int v = (total & 7) == 0 ? (x & 7) * (total >> 3) + (x >> 3) : x;
    Wg6Chunk c;
    c.bgrp = v / S; c.s = v - c.bgrp * S; c.U = U;
    c.mode = (U % S == 0) ? 1 : ((S % U == 0) ? 2 : 0);
    c.nseg = c.mode == 1 ? U / S : 1;
    return c;
}
// steps [t0, t1) of segment k of the chunk, in the flattened order t = column * H + row (column = sample-of-group * tiles_x + tile)
__device__ __forceinline__ void wg6_segment(const Wg6Chunk& c, int k, int S, int H, int n_steps, int& t0, int& t1) {
    if (c.mode == 1) { t0 = (c.s + k * S) * H; t1 = t0 + H; }
    else if (c.mode == 2) {
        const int u = c.s % c.U, r = c.s / c.U, R = S / c.U;
        t0 = u * H + (int)((int64_t)H * r / R); t1 = u * H + (int)((int64_t)H * (r + 1) / R);
    } else { t0 = (int)((int64_t)n_steps * c.s / S); t1 = (int)((int64_t)n_steps * (c.s + 1) / S); }
}

#ifndef WG6_ILV
#define WG6_ILV 1    // 1: consecutive MFMAs on different accumulators (see the MFMA loop); 0: six in a row per accumulator (A/B: same time)
#endif
#ifndef WG6_SLOT0
#define WG6_SLOT0 0  // MFMA behind which the 64-slot staging program starts (72 MFMAs per step)
#endif
#ifdef WG6_PROF      // experimental builds: cycle counts per wave (tools/wgrad6_check.py prof)
__device__ unsigned long long te_wgrad6_prof_buf[2048 * 4 * 4];
#endif

// PAIR (round 6; Co == Ci == 32, NB == 1, B even - the 32-channel 3x3 layer at 1024^2 of the FFHQ-1024 generator, reference channel table
// model_spatial_query.py:473-483): the 64 lanes of a staging wave are the 32 channels of sample 2 k (lanes 0-31) and the 32 channels of
// sample 2 k + 1 (lanes 32-63), so the block tile holds TWO 32 x 32 problems on its diagonal: wave (0, 0) accumulates sample 2 k, wave
// (1, 1) sample 2 k + 1, the two off-diagonal waves (products across the two samples) only stage.  Half of the block's matrix pipes
// idle, but every staged element is used and the launch leaves the fp32 pipe: 1 255 us -> see profiles/r06_*.
template <bool PAIR>
__global__ __launch_bounds__(WT, 1) void wgrad6_kernel(const Wg6Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wid >> 1, wci = wid & 1;                  // multiplying: the wave's 32 x 32 tile
    const bool role = (wid >> 1) != 0;                        // staging: false = gradient rows (waves 0, 1), true = input rows (2, 3)
    const int kh = wid & 1;                                   //          k half = cells 16 kh .. 16 kh + 15 of the row segment; lane = channel
    const bool active = !PAIR || wco == wci;                  // (wave-uniform) does this wave's tile belong to a problem?

    const Wg6Chunk ck = wg6_chunk(p.S, (PAIR ? 1 : p.NB) * p.tiles_x);
    const int s_chunk = ck.s, bgrp = ck.bgrp, b = PAIR ? 2 * bgrp : bgrp * p.NB;
    const int co0 = PAIR ? 0 : blockIdx.y * TC, ci0 = PAIR ? 0 : blockIdx.z * TC;
    const size_t plane = (size_t)p.H * p.W;
    const unsigned g_sample = (unsigned)(p.Co * plane * 4), x_sample = (unsigned)(p.Ci * plane * 4);      // bytes (host-checked < 2 GiB per group)
    const unsigned nb_span = PAIR ? 2u : (unsigned)p.NB;
    const __amdgpu_buffer_rsrc_t rs = role ? make_rsrc(p.x + (size_t)b * p.Ci * plane, x_sample * nb_span)
                                           : make_rsrc(p.g + (size_t)b * p.Co * plane, g_sample * nb_span);
    const unsigned my_sample = role ? x_sample : g_sample;
    const unsigned chan_off = PAIR ? (unsigned)(l31 * plane * 4) + (lane >> 5) * my_sample                 // channel l31 of sample b + (lane >> 5)
                                   : (unsigned)(((role ? ci0 : co0) + lane) * plane * 4);                  // bytes

    f32x16 acc[12];                                           // [ky * 4 + component]
#pragma unroll
    for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- staging: raw registers -> 12 LDS elements (4 components x 3 pieces) of this lane's (channel, k half)
    // rw[i] = column c0 - 1 + i of the row (c0 = first column of the lane's 16 cells); gradient rows: rw[0] = rw[17] = 0 (never used)
    auto load_raw = [&](float (&rw)[NRAW], int bb, int cx, int row) {
        const bool ok = (unsigned)row < (unsigned)p.H;
        const int c0 = cx * 32 + 16 * kh;
        const unsigned off = (unsigned)bb * my_sample + chan_off + (unsigned)((row * p.W + c0) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = buf_load4w(rs, ok ? off + 16u * q : OOBW);
#pragma unroll
            for (int e = 0; e < 4; ++e) rw[1 + 4 * q + e] = v[e];
        }
        rw[0] = buf_load(rs, (ok && role && c0 > 0) ? off - 4u : OOBW);
        rw[17] = buf_load(rs, (ok && role && c0 + 16 < p.W) ? off + 64u : OOBW);
    };
    // one step of the split of unit (component j, dword d = pairs 2d, 2d + 1): the four steps of wino6.hip's slot program
    unsigned pcs[3][4];                                       // [piece][dword] of the component in work
    // (One unit advances per slot.  Instructions of the wave's own stream are not free - a step costs ~4.5 cycles per vector-ALU
    //  instruction on top of the MFMAs' 35 - so the program is kept short: no register pins on the finished pieces (they cost a
    //  move each), selects only where the two roles differ.  Bundling two units per slot - 32 full slots, 40 empty - measured 3 660
    //  cycles per step against 3 430; the subtractions written as float2 came out scalar again: profiles/experiments/r05_wgrad6.log.)
    // (v_pk_add_f32 for the two subtractions of a step - inline assembly on 64-bit register pairs - measured 4 094 cycles per step
    //  against 3 430: packed fp32 is not faster here and drags hazard nops in)
    float v0 = 0.f, v1 = 0.f, f0 = 0.f, f1 = 0.f;
    auto unit_step = [&](const float (&rw)[NRAW], int j, int d, int step) {
        if (step == 0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int q = 2 * (2 * d + e);                // rw index of e0 of pair 2d + e
                float v;
                if (j == 0) v = (role ? rw[q] - rw[q + 2] : rw[q + 1]);
                else if (j == 1) v = rw[q + 1] + rw[q + 2];
                else if (j == 2) v = rw[q + 2] - rw[q + 1];
                else v = (role ? rw[q + 1] - rw[q + 3] : -rw[q + 2]);
                if (e == 0) v0 = v; else v1 = v;
            }
            const f32x2 t = {v0, v1};
            const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
            pcs[0][d] = h;
            f0 = __builtin_bit_cast(float, h << 16);
            f1 = __builtin_bit_cast(float, h & 0xFFFF0000u);
        } else if (step == 1) {
            v0 -= f0; v1 -= f1;
        } else if (step == 2) {
            const f32x2 t = {v0, v1};
            const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
            pcs[1][d] = m;
            f0 = __builtin_bit_cast(float, m << 16);
            f1 = __builtin_bit_cast(float, m & 0xFFFF0000u);
        } else {
            v0 -= f0; v1 -= f1;
            const f32x2 t = {v0, v1};
            pcs[2][d] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
        }
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(f0), "+v"(f1));          // (pin the step where it is written: wino6.hip)
    };
    // element index of this lane inside an image: ((piece * 4 + j) * 2 + kh) * 64 + lane
    const int w_elem = kh * TC + lane;
    auto write_comp = [&](int img, int j) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            u32x4 v; v[0] = pcs[pc][0]; v[1] = pcs[pc][1]; v[2] = pcs[pc][2]; v[3] = pcs[pc][3];
            lds[img * IMG + (pc * 4 + j) * 2 * TC + w_elem] = v;
        }
    };
    // slot k of the staging program (64 slots): unit k >> 2 = (component k >> 4, dword (k >> 2) & 3), step k & 3; a component's three
    // elements go to LDS behind its last step
    auto stage_slot = [&](const float (&rw)[NRAW], int k, int img) {
        unit_step(rw, k >> 4, (k >> 2) & 3, k & 3);
        if ((k & 15) == 15) write_comp(img, k >> 4);
    };
    auto stage_all = [&](const float (&rw)[NRAW], int img) {               // outside the MFMA stream (head of a sweep)
#pragma unroll
        for (int k = 0; k < 64; ++k) stage_slot(rw, k, img);
    };
    auto g_img = [&](int y) { return y & 1; };
    auto x_img = [&](int y) { return 2 + (y & 3); };

    const int sps = p.tiles_x * p.H;                          // steps of one sample
    const int n_steps = sps * (PAIR ? 1 : p.NB);
    const int a_elem = half * TC + wco * 32 + l31;            // + ((piece * 4 + j) * 2) * 64 + image * IMG
    const int b_elem = half * TC + wci * 32 + l31;
#ifdef WG6_PROF
    unsigned long long pc_mult = 0, pc_head = 0, pc_bar = 0;
    const unsigned long long pstart = __builtin_readcyclecounter();
    int nstep_done = 0;
#endif

    float rw[NRAW], rn[NRAW];
    for (int sg = 0; sg < ck.nseg; ++sg) {
    int t, t_end;
    wg6_segment(ck, sg, p.S, p.H, n_steps, t, t_end);
    while (t < t_end) {
        const int bb = t / sps, rem = t - bb * sps, cx = rem / p.H, ya = rem - cx * p.H;
        const int n = min(p.H - ya, t_end - t), yb = ya + n;
        t += n;
#ifdef WG6_PROF
        const unsigned long long th0 = __builtin_readcyclecounter();
#endif
        // ---- head of a sweep: input rows ya - 1, ya, ya + 1 and gradient row ya, then the registers of the first step
        {
            float r0[NRAW], r1[NRAW];
            load_raw(r0, bb, cx, role ? ya - 1 : -1);
            load_raw(r1, bb, cx, role ? ya : -1);
            load_raw(rn, bb, cx, role ? ya + 1 : ya);
            load_raw(rw, bb, cx, role ? ya + 2 : ya + 1);
            // (gradient role: the first two rounds write zeros to the image the third one fills - same lane, same elements, in order)
            stage_all(r0, role ? x_img(ya - 1) : g_img(ya));
            stage_all(r1, role ? x_img(ya) : g_img(ya));
            stage_all(rn, role ? x_img(ya + 1) : g_img(ya));
            // (a use of the first step's registers HERE: the compiler then waits for their loads at the end of the head; left pending
            //  they put an `s_waitcnt vmcnt(5)` - one load of the step itself - in front of the arithmetic of every step of the loop)
#pragma unroll
            for (int i = 0; i < NRAW; ++i) asm volatile("" :: "v"(rw[i]));
        }
        w6g_barrier();
#ifdef WG6_PROF
        pc_head += __builtin_readcyclecounter() - th0;
#endif
        for (int y = ya; y < yb; ++y) {
#ifdef WG6_PROF
            const unsigned long long tm0 = __builtin_readcyclecounter();
#endif
            // registers of the step after next: input row y + 3 / gradient row y + 2 (a whole step to land)
            load_raw(rn, bb, cx, role ? y + 3 : y + 2);
            const int w_img = role ? x_img(y + 2) : g_img(y + 1);           // image this lane's results of the step go to
            const int a_base = g_img(y) * IMG + a_elem;
            int b_base[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) b_base[ky] = x_img(y - 1 + ky) * IMG + b_elem;
#if WG6_ILV
            // MFMA order inside a component: product q of the three tap rows in turn, so that consecutive MFMAs write DIFFERENT
            // accumulators (an accumulator is touched every third instruction; each one still sees its six products in the order
            // mm, hl, lh, hm, mh, hh).  Measured against six in a row into one accumulator (WG6_ILV=0): 3 419 against 3 465 cycles per
            // step - the dependent accumulate chain is not what holds the stream at ~48 cycles per MFMA (the staging program is).
            bf16x8 av[2][3], bv[2][3][3];
            auto rd_a = [&](int j, int pc) { av[j & 1][pc] = __builtin_bit_cast(bf16x8, lds[a_base + (pc * 4 + j) * 2 * TC]); };
            auto rd_b = [&](int j, int ky, int pc) { bv[j & 1][ky][pc] = __builtin_bit_cast(bf16x8, lds[b_base[ky] + (pc * 4 + j) * 2 * TC]); };
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                rd_a(0, pc);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) rd_b(0, ky, pc);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 18; ++m) {
                    const int q = m / 3, ky = m % 3;
#ifndef WG6_SKIP_MFMA
                    if (active) acc[ky * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[j & 1][PA[q]], bv[j & 1][ky][PB[q]], acc[ky * 4 + j], 0, 0, 0);
#endif
                    if (j + 1 < 4 && m < 12) {                                            // operands of the next component: 12 reads
                        if (m < 3) rd_a(j + 1, m);
                        else rd_b(j + 1, (m - 3) / 3, (m - 3) % 3);
                    }
#ifndef WG6_SKIP_ARITH
                    {   // the staging program: slot k = unit (k >> 2), step (k & 3); a component's three elements go to LDS behind its last step
                        const int k = j * 18 + m - WG6_SLOT0;
                        if (k >= 0 && k < 64) stage_slot(rw, k, w_img);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#else
            bf16x8 av[2][3], bv[2][3];
            auto rd_a = [&](int j, int pc) { av[j & 1][pc] = __builtin_bit_cast(bf16x8, lds[a_base + (pc * 4 + j) * 2 * TC]); };
            auto rd_b = [&](int gi, int pc) {
                const int j = gi / 3, ky = gi % 3;
                bv[gi & 1][pc] = __builtin_bit_cast(bf16x8, lds[b_base[ky] + (pc * 4 + j) * 2 * TC]);
            };
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) { rd_a(0, pc); rd_b(0, pc); }
#pragma unroll
            for (int gi = 0; gi < 12; ++gi) {                                            // group = (component j, tap row ky)
                const int j = gi / 3, ky = gi % 3;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
#ifndef WG6_SKIP_MFMA
                    if (active) acc[ky * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[j & 1][PA[q]], bv[gi & 1][PB[q]], acc[ky * 4 + j], 0, 0, 0);
#endif
                    if (gi + 1 < 12 && q < 3) {                                           // operands of the next group
                        rd_b(gi + 1, q);
                        if ((gi + 1) % 3 == 0) rd_a((gi + 1) / 3, q);
                    }
#ifndef WG6_SKIP_ARITH
                    {   // the staging program: slot k = unit (k >> 2), step (k & 3); a component's three elements go to LDS behind its last step
                        const int k = gi * 6 + q - WG6_SLOT0;
                        if (k >= 0 && k < 64) stage_slot(rw, k, w_img);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#endif
#ifdef WG6_PROF
            const unsigned long long tm1 = __builtin_readcyclecounter();
#endif
#pragma unroll
            for (int i = 0; i < NRAW; ++i) rw[i] = rn[i];
            w6g_barrier();
#ifdef WG6_PROF
            { const unsigned long long tm2 = __builtin_readcyclecounter(); pc_mult += tm1 - tm0; pc_bar += tm2 - tm1; ++nstep_done; }
#endif
        }
    }
    }
#ifdef WG6_PROF
    {
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (lane == 0 && lin < 2048) {
            unsigned long long* d = te_wgrad6_prof_buf + ((size_t)lin * 4 + wid) * 4;
            d[0] = pc_mult; d[1] = pc_head; d[2] = pc_bar;
            d[3] = ((unsigned long long)nstep_done << 40) | ((__builtin_readcyclecounter() - pstart) & 0xFFFFFFFFFFull);
        }
    }
#endif

    // ---- fold the twelve accumulators to the nine taps and write the slab tile: slab[b][s][co][ci][tap]
    if (!active) return;
    float* sl = p.slabs + ((size_t)(PAIR ? 2 * bgrp + wco : bgrp) * p.S + s_chunk) * p.Co * p.Ci * 9;
    const int ci = PAIR ? l31 : ci0 + wci * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = (PAIR ? 0 : co0 + wco * 32) + (r & 3) + 8 * (r >> 2) + 4 * half;
        float* dst = sl + ((size_t)co * p.Ci + ci) * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float hs = 0.5f * (acc[4 * ky + 1][r] - acc[4 * ky + 2][r]);
            dst[3 * ky + 0] = acc[4 * ky + 0][r] + hs;
            dst[3 * ky + 1] = 0.5f * (acc[4 * ky + 1][r] + acc[4 * ky + 2][r]);
            dst[3 * ky + 2] = hs + acc[4 * ky + 3][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- transposed kind
// TE_CONV_T2 (the weight gradient of the generator's up-sampling and - with the two tensors swapped - of the discriminator's
// down-sampling convolutions, model_spatial_query.py:310-321 / :765-779):
//     slab[b][s][co][ci][ky][kx] = sum over the cells (i, j)   g[b, co, 2 i + ky, 2 j + kx] * x[b, ci, i, j]        g: (2H+1) x (2W+1), x: H x W
// Stride 2 decouples the taps (no Winograd form): the direct 9 taps x 6 piece products = 54 MFMAs per wave and step of 16 cells of a
// row (the reduction runs over cells: one MFMA = 16 cells).  The A operand of tap (ky, kx) is a stride-2 subsequence of g row 2 i + ky -
// even columns from j (kx = 0), odd columns (kx = 1), even columns from j + 1 (kx = 2) - so a g row lives in LDS as three COMPONENTS
// x three pieces, [piece 3][component 3][k half 2][channel 64][8 bf16] = 18 KB, in a ring of five rows (a step reads rows 2i, 2i+1,
// 2i+2 while rows 2i+3, 2i+4 are written: every g row is split once per sweep); the third component is the first one shifted by one
// value, built from its packed pieces with v_alignbit, not split again.  x rows: [piece 3][k half 2][channel 64], double buffer.
// Same block shape and pipeline as wgrad6_kernel: 4 waves, one per SIMD, 64 x 64 channels, the staging program of the next step
// behind the MFMAs of the current one, loads a whole step ahead.  All four waves have the same staging work: one g unit (row
// 2i + 3 + (wave >> 1), 8 cells = 17 columns, one channel per lane) and one x unit (waves 2, 3 repeat the x units of waves 0, 1).
constexpr int GI = 3 * 3 * 2 * TC;            // elements of one g-row image: 1 152 (18 KB)
constexpr int XI = 3 * 2 * TC;                // elements of one x-row image: 384 (6 KB)
constexpr int NGR = 5;

// NARROW (round 6): Co or Ci is an odd multiple of 32 (the 64 -> 32 up-sampling layer of the FFHQ-1024 generator): the last channel block
// of that side has 32 valid channels.  Lanes past them read the last valid channel again (no access outside the tensor; their LDS
// elements only feed wave tiles that are not computed), and the waves whose 32 x 32 tile lies in the padding skip MFMAs and stores.
template <bool NARROW>
__global__ __launch_bounds__(WT, 1) void wgrad6t_kernel(const Wg6Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wid >> 1, wci = wid & 1;                  // multiplying: the wave's 32 x 32 tile
    const int rsel = wid >> 1, q = wid & 1;                   // staging: g row 2i + 3 + rsel, cells 8 q .. 8 q + 7 of the step; lane = channel

    const Wg6Chunk ck = wg6_chunk(p.S, p.NB * (p.W / 16));
    const int s_chunk = ck.s, bgrp = ck.bgrp, b = bgrp * p.NB;
    const int co0 = blockIdx.y * TC, ci0 = blockIdx.z * TC;
    const int Hg = 2 * p.H + 1, Wg = 2 * p.W + 1;
    const size_t gplane = (size_t)Hg * Wg, xplane = (size_t)p.H * p.W;
    // addresses = uniform 64-bit base (scalar registers) + this lane's 32-bit byte offset (its channel): the scalar-base form of the
    // global loads, no 64-bit vector arithmetic in the step
    const float* gblk = p.g + ((size_t)b * p.Co + co0) * gplane;             // channel block of the group's first sample
    const float* xblk = p.x + ((size_t)b * p.Ci + ci0) * xplane;
    const int g_ch = NARROW ? min(lane, p.Co - co0 - 1) : lane, x_ch = NARROW ? min(lane, p.Ci - ci0 - 1) : lane;
    const unsigned g_lane = (unsigned)(g_ch * gplane * 4), x_lane = (unsigned)(x_ch * xplane * 4);       // bytes (host-checked < 2 GiB)
    const bool active = !NARROW || (co0 + wco * 32 < p.Co && ci0 + wci * 32 < p.Ci);                     // wave-uniform

    f32x16 acc[9];                                            // [ky * 3 + kx]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // rows past the end are clamped to the last row (their registers are never used: no element of g or x is padding)
    auto load_g = [&](float (&rg)[17], int bb, int cx, int grow) {
        const char* src = reinterpret_cast<const char*>(gblk + (size_t)bb * p.Co * gplane + (size_t)min(grow, Hg - 1) * Wg + 2 * (16 * cx + 8 * q));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4u*>(src + 16 * k + g_lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) rg[4 * k + e] = v[e];
        }
        rg[16] = *reinterpret_cast<const float*>(src + 64 + g_lane);
    };
    auto load_x = [&](float (&rx)[8], int bb, int cx, int xrow) {
        const char* src = reinterpret_cast<const char*>(xblk + (size_t)bb * p.Ci * xplane + (size_t)min(xrow, p.H - 1) * p.W + 16 * cx + 8 * q);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4u*>(src + 16 * k + x_lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) rx[4 * k + e] = v[e];
        }
    };
    // split of one packed pair of values in three steps; units 0..4: even columns of the g unit (pair d = columns 4d, 4d + 2; the
    // last one holds column 16 alone), 5..8: odd columns (4d + 1, 4d + 3), 9..12: the x unit
    unsigned pe[3][5], po[3][4], px[3][4];
    float v0 = 0.f, v1 = 0.f, f0 = 0.f, f1 = 0.f;
    auto unit_step = [&](const float (&rg)[17], const float (&rx)[8], int u, int step) {
        unsigned& d0 = u < 5 ? pe[0][u] : (u < 9 ? po[0][u - 5] : px[0][u - 9]);
        unsigned& d1 = u < 5 ? pe[1][u] : (u < 9 ? po[1][u - 5] : px[1][u - 9]);
        unsigned& d2 = u < 5 ? pe[2][u] : (u < 9 ? po[2][u - 5] : px[2][u - 9]);
        if (step == 0) {
            if (u < 5) { v0 = rg[4 * u]; v1 = u < 4 ? rg[4 * u + 2] : 0.f; }
            else if (u < 9) { v0 = rg[4 * (u - 5) + 1]; v1 = rg[4 * (u - 5) + 3]; }
            else { v0 = rx[2 * (u - 9)]; v1 = rx[2 * (u - 9) + 1]; }
            const f32x2 t = {v0, v1};
            const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
            d0 = h;
            f0 = __builtin_bit_cast(float, h << 16);
            f1 = __builtin_bit_cast(float, h & 0xFFFF0000u);
        } else if (step == 1) {
            v0 -= f0; v1 -= f1;
            const f32x2 t = {v0, v1};
            const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
            d1 = m;
            f0 = __builtin_bit_cast(float, m << 16);
            f1 = __builtin_bit_cast(float, m & 0xFFFF0000u);
        } else {
            v0 -= f0; v1 -= f1;
            const f32x2 t = {v0, v1};
            d2 = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
        }
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(f0), "+v"(f1));
    };
    const int w_elem = q * TC + lane;             // + ((piece * 3 + component) * 2) * 64 (g) / (piece * 2) * 64 (x)
    auto write_g = [&](int gimg, int comp, int pc) {          // component 0: even columns, 1: odd columns, 2: even columns from the second on
        u32x4 v;
        if (comp == 0) { v[0] = pe[pc][0]; v[1] = pe[pc][1]; v[2] = pe[pc][2]; v[3] = pe[pc][3]; }
        else if (comp == 1) { v[0] = po[pc][0]; v[1] = po[pc][1]; v[2] = po[pc][2]; v[3] = po[pc][3]; }
        else {
#pragma unroll
            for (int d = 0; d < 4; ++d) v[d] = __builtin_amdgcn_alignbit(pe[pc][d + 1], pe[pc][d], 16);
        }
        lds[gimg * GI + (pc * 3 + comp) * 2 * TC + w_elem] = v;
    };
    auto write_x = [&](int ximg, int pc) {
        u32x4 v; v[0] = px[pc][0]; v[1] = px[pc][1]; v[2] = px[pc][2]; v[3] = px[pc][3];
        lds[NGR * GI + ximg * XI + pc * 2 * TC + w_elem] = v;
    };
    // the staging program: slots 0..38 = 13 units x 3 steps, the writes of a finished group in the slot of its last step (the
    // shifted component in slots 39..41)
    auto stage_slot = [&](const float (&rg)[17], const float (&rx)[8], int k, int gimg, int ximg) {
        if (k < 39) unit_step(rg, rx, k / 3, k % 3);
        if (k == 14) { write_g(gimg, 0, 0); write_g(gimg, 0, 1); write_g(gimg, 0, 2); }
        if (k == 26) { write_g(gimg, 1, 0); write_g(gimg, 1, 1); write_g(gimg, 1, 2); }
        if (k == 38) { write_x(ximg, 0); write_x(ximg, 1); write_x(ximg, 2); }
        if (k >= 39 && k < 42) write_g(gimg, 2, k - 39);
    };

    const int tiles_x = p.W / 16;
    const int sps = tiles_x * p.H, n_steps = sps * p.NB;
    const int a_elem = half * TC + wco * 32 + l31;
    const int b_elem = NGR * GI + half * TC + wci * 32 + l31;

#ifdef WG6_PROF
    unsigned long long pc_mult = 0, pc_head = 0, pc_bar = 0;
    const unsigned long long pstart = __builtin_readcyclecounter();
    int nstep_done = 0;
#endif
    float rg[17], rx[8], ng[17], nx[8];
    for (int sg = 0; sg < ck.nseg; ++sg) {
    int t, t_end;
    wg6_segment(ck, sg, p.S, p.H, n_steps, t, t_end);
    while (t < t_end) {
        const int bb = t / sps, rem = t - bb * sps, cx = rem / p.H, ya = rem - cx * p.H;
        const int n = min(p.H - ya, t_end - t), yb = ya + n;
        t += n;
#ifdef WG6_PROF
        const unsigned long long th0 = __builtin_readcyclecounter();
#endif
        // ---- head of a sweep: g rows 2 ya .. 2 ya + 3 into ring slots 0..3, x row ya into buffer ya & 1, registers of the first step
        {
            float g0[17], g1[17];
            load_g(g0, bb, cx, 2 * ya + rsel);
            load_g(g1, bb, cx, 2 * ya + 2 + rsel);
            load_x(nx, bb, cx, ya);
            load_g(rg, bb, cx, 2 * ya + 3 + rsel);
            load_x(rx, bb, cx, ya + 1);
#pragma unroll
            for (int k = 0; k < 42; ++k) stage_slot(g0, nx, k, rsel, ya & 1);
#pragma unroll
            for (int k = 0; k < 42; ++k) stage_slot(g1, nx, k, 2 + rsel, ya & 1);
#pragma unroll
            for (int i = 0; i < 17; ++i) asm volatile("" :: "v"(rg[i]));          // (wait for the first step's registers here: wgrad6_kernel)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(rx[i]));
        }
        w6g_barrier();
#ifdef WG6_PROF
        pc_head += __builtin_readcyclecounter() - th0;
#endif
        int s0 = 0;                                           // ring slot of g row 2 y
        for (int y = ya; y < yb; ++y) {
#ifdef WG6_PROF
            const unsigned long long tm0 = __builtin_readcyclecounter();
#endif
            load_g(ng, bb, cx, 2 * y + 5 + rsel);             // registers of the step after next
            load_x(nx, bb, cx, y + 2);
            int wslot = s0 + 3 + rsel; wslot -= wslot >= NGR ? NGR : 0;
            int a_base[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) { int sl = s0 + ky; sl -= sl >= NGR ? NGR : 0; a_base[ky] = sl * GI + a_elem; }
            const int b_base = b_elem + (y & 1) * XI;
            bf16x8 av[2][3], bx[3];
            auto rd_a = [&](int tp, int pc) {
                const int ky = tp / 3, kx = tp % 3;
                av[tp & 1][pc] = __builtin_bit_cast(bf16x8, lds[a_base[ky] + (pc * 3 + kx) * 2 * TC]);
            };
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) { bx[pc] = __builtin_bit_cast(bf16x8, lds[b_base + pc * 2 * TC]); rd_a(0, pc); }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int qq = 0; qq < 6; ++qq) {
#ifndef WG6_SKIP_MFMA
                    if (active) acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[tp & 1][PA[qq]], bx[PB[qq]], acc[tp], 0, 0, 0);
#endif
                    if (tp + 1 < 9 && qq < 3) rd_a(tp + 1, qq);
#ifndef WG6_SKIP_ARITH
                    stage_slot(rg, rx, tp * 6 + qq, wslot, (y + 1) & 1);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#ifdef WG6_PROF
            const unsigned long long tm1 = __builtin_readcyclecounter();
#endif
#pragma unroll
            for (int i = 0; i < 17; ++i) rg[i] = ng[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) rx[i] = nx[i];
            s0 += 2; s0 -= s0 >= NGR ? NGR : 0;
            w6g_barrier();
#ifdef WG6_PROF
            { const unsigned long long tm2 = __builtin_readcyclecounter(); pc_mult += tm1 - tm0; pc_bar += tm2 - tm1; ++nstep_done; }
#endif
        }
    }
    }
#ifdef WG6_PROF
    {
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (lane == 0 && lin < 2048) {
            unsigned long long* d = te_wgrad6_prof_buf + ((size_t)lin * 4 + wid) * 4;
            d[0] = pc_mult; d[1] = pc_head; d[2] = pc_bar;
            d[3] = ((unsigned long long)nstep_done << 40) | ((__builtin_readcyclecounter() - pstart) & 0xFFFFFFFFFFull);
        }
    }
#endif

    if (!active) return;
    float* sl = p.slabs + ((size_t)bgrp * p.S + s_chunk) * p.Co * p.Ci * 9;
    const int ci = ci0 + wci * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float* dst = sl + ((size_t)co * p.Ci + ci) * 9;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) dst[tp] = acc[tp][r];
    }
}

// WIDE form of the transposed kind (round 6, Ci % 128 == 0 - every transposed-kind layer of the FFHQ-256 model): the block tile is 64
// channels of g x 128 channels of x.  The counters place wgrad6t_kernel BELOW the power cap (45.5 % MFMA busy at 2.2 GHz, HBM reads 2.1x the
// algorithmic bytes): its vector-ALU staging - which a wave's own MFMAs do not hide (r05_wgrad6.log) - is the g unit (17 columns, 3
// components) plus an x unit per wave and step of 54 MFMAs, and waves 2, 3 repeated the x units of waves 0, 1.  Here a wave multiplies
// its 32 rows of g by 32 channels of BOTH x tiles (108 MFMAs per step, 2 x 9 accumulators = 288 registers of the 512 a lone wave per SIMD
// owns), stages the same g unit and - waves 2, 3 - an x unit of the second tile: half the staging instructions, LDS writes and g loads
// per MFMA, and g is read Ci / 128 instead of Ci / 64 times.  Same products in the same order per slab element: bit-identical slabs.
__global__ __launch_bounds__(WT, 1) void wgrad6tw_kernel(const Wg6Args p) {
    constexpr bool NARROW = false;
    constexpr int XI2 = 2 * XI;                   // x-row image of this form: [x tile 2][piece 3][k half 2][channel 64]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wid >> 1, wci = wid & 1;                  // multiplying: the wave's 32 x 32 tile
    const int rsel = wid >> 1, q = wid & 1;                   // staging: g row 2i + 3 + rsel and x tile rsel, cells 8 q .. 8 q + 7 of the step; lane = channel

    const Wg6Chunk ck = wg6_chunk(p.S, p.NB * (p.W / 16));
    const int s_chunk = ck.s, bgrp = ck.bgrp, b = bgrp * p.NB;
    const int co0 = blockIdx.y * TC, ci0 = blockIdx.z * 2 * TC;
    const int Hg = 2 * p.H + 1, Wg = 2 * p.W + 1;
    const size_t gplane = (size_t)Hg * Wg, xplane = (size_t)p.H * p.W;
    // addresses = uniform 64-bit base (scalar registers) + this lane's 32-bit byte offset (its channel): the scalar-base form of the
    // global loads, no 64-bit vector arithmetic in the step
    const float* gblk = p.g + ((size_t)b * p.Co + co0) * gplane;             // channel block of the group's first sample
    const float* xblk = p.x + ((size_t)b * p.Ci + ci0) * xplane;
    const int g_ch = lane, x_ch = rsel * TC + lane;           // (waves 2, 3 stage the second x tile: no wave repeats another's unit)
    const unsigned g_lane = (unsigned)(g_ch * gplane * 4), x_lane = (unsigned)(x_ch * xplane * 4);       // bytes (host-checked < 2 GiB)

    f32x16 acc[2][9];                                         // [x tile][ky * 3 + kx]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

    // rows past the end are clamped to the last row (their registers are never used: no element of g or x is padding)
    auto load_g = [&](float (&rg)[17], int bb, int cx, int grow) {
        const char* src = reinterpret_cast<const char*>(gblk + (size_t)bb * p.Co * gplane + (size_t)min(grow, Hg - 1) * Wg + 2 * (16 * cx + 8 * q));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4u*>(src + 16 * k + g_lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) rg[4 * k + e] = v[e];
        }
        rg[16] = *reinterpret_cast<const float*>(src + 64 + g_lane);
    };
    auto load_x = [&](float (&rx)[8], int bb, int cx, int xrow) {
        const char* src = reinterpret_cast<const char*>(xblk + (size_t)bb * p.Ci * xplane + (size_t)min(xrow, p.H - 1) * p.W + 16 * cx + 8 * q);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4u*>(src + 16 * k + x_lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) rx[4 * k + e] = v[e];
        }
    };
    // split of one packed pair of values in three steps; units 0..4: even columns of the g unit (pair d = columns 4d, 4d + 2; the
    // last one holds column 16 alone), 5..8: odd columns (4d + 1, 4d + 3), 9..12: the x unit
    unsigned pe[3][5], po[3][4], px[3][4];
    float v0 = 0.f, v1 = 0.f, f0 = 0.f, f1 = 0.f;
    auto unit_step = [&](const float (&rg)[17], const float (&rx)[8], int u, int step) {
        unsigned& d0 = u < 5 ? pe[0][u] : (u < 9 ? po[0][u - 5] : px[0][u - 9]);
        unsigned& d1 = u < 5 ? pe[1][u] : (u < 9 ? po[1][u - 5] : px[1][u - 9]);
        unsigned& d2 = u < 5 ? pe[2][u] : (u < 9 ? po[2][u - 5] : px[2][u - 9]);
        if (step == 0) {
            if (u < 5) { v0 = rg[4 * u]; v1 = u < 4 ? rg[4 * u + 2] : 0.f; }
            else if (u < 9) { v0 = rg[4 * (u - 5) + 1]; v1 = rg[4 * (u - 5) + 3]; }
            else { v0 = rx[2 * (u - 9)]; v1 = rx[2 * (u - 9) + 1]; }
            const f32x2 t = {v0, v1};
            const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
            d0 = h;
            f0 = __builtin_bit_cast(float, h << 16);
            f1 = __builtin_bit_cast(float, h & 0xFFFF0000u);
        } else if (step == 1) {
            v0 -= f0; v1 -= f1;
            const f32x2 t = {v0, v1};
            const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
            d1 = m;
            f0 = __builtin_bit_cast(float, m << 16);
            f1 = __builtin_bit_cast(float, m & 0xFFFF0000u);
        } else {
            v0 -= f0; v1 -= f1;
            const f32x2 t = {v0, v1};
            d2 = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
        }
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(f0), "+v"(f1));
    };
    const int w_elem = q * TC + lane;             // + ((piece * 3 + component) * 2) * 64 (g) / (piece * 2) * 64 (x)
    auto write_g = [&](int gimg, int comp, int pc) {          // component 0: even columns, 1: odd columns, 2: even columns from the second on
        u32x4 v;
        if (comp == 0) { v[0] = pe[pc][0]; v[1] = pe[pc][1]; v[2] = pe[pc][2]; v[3] = pe[pc][3]; }
        else if (comp == 1) { v[0] = po[pc][0]; v[1] = po[pc][1]; v[2] = po[pc][2]; v[3] = po[pc][3]; }
        else {
#pragma unroll
            for (int d = 0; d < 4; ++d) v[d] = __builtin_amdgcn_alignbit(pe[pc][d + 1], pe[pc][d], 16);
        }
        lds[gimg * GI + (pc * 3 + comp) * 2 * TC + w_elem] = v;
    };
    auto write_x = [&](int ximg, int pc) {
        u32x4 v; v[0] = px[pc][0]; v[1] = px[pc][1]; v[2] = px[pc][2]; v[3] = px[pc][3];
        lds[NGR * GI + ximg * XI2 + rsel * XI + pc * 2 * TC + w_elem] = v;
    };
    // the staging program: slots 0..38 = 13 units x 3 steps, the writes of a finished group in the slot of its last step (the
    // shifted component in slots 39..41)
    auto stage_slot = [&](const float (&rg)[17], const float (&rx)[8], int k, int gimg, int ximg) {
        if (k < 39) unit_step(rg, rx, k / 3, k % 3);
        if (k == 14) { write_g(gimg, 0, 0); write_g(gimg, 0, 1); write_g(gimg, 0, 2); }
        if (k == 26) { write_g(gimg, 1, 0); write_g(gimg, 1, 1); write_g(gimg, 1, 2); }
        if (k == 38) { write_x(ximg, 0); write_x(ximg, 1); write_x(ximg, 2); }
        if (k >= 39 && k < 42) write_g(gimg, 2, k - 39);
    };

    const int tiles_x = p.W / 16;
    const int sps = tiles_x * p.H, n_steps = sps * p.NB;
    const int a_elem = half * TC + wco * 32 + l31;
    const int b_elem = NGR * GI + half * TC + wci * 32 + l31;

#ifdef WG6_PROF
    unsigned long long pc_mult = 0, pc_head = 0, pc_bar = 0;
    const unsigned long long pstart = __builtin_readcyclecounter();
    int nstep_done = 0;
#endif
    float rg[17], rx[8], ng[17], nx[8];
    for (int sg = 0; sg < ck.nseg; ++sg) {
    int t, t_end;
    wg6_segment(ck, sg, p.S, p.H, n_steps, t, t_end);
    while (t < t_end) {
        const int bb = t / sps, rem = t - bb * sps, cx = rem / p.H, ya = rem - cx * p.H;
        const int n = min(p.H - ya, t_end - t), yb = ya + n;
        t += n;
#ifdef WG6_PROF
        const unsigned long long th0 = __builtin_readcyclecounter();
#endif
        // ---- head of a sweep: g rows 2 ya .. 2 ya + 3 into ring slots 0..3, x row ya into buffer ya & 1, registers of the first step
        {
            float g0[17], g1[17];
            load_g(g0, bb, cx, 2 * ya + rsel);
            load_g(g1, bb, cx, 2 * ya + 2 + rsel);
            load_x(nx, bb, cx, ya);
            load_g(rg, bb, cx, 2 * ya + 3 + rsel);
            load_x(rx, bb, cx, ya + 1);
#pragma unroll
            for (int k = 0; k < 42; ++k) stage_slot(g0, nx, k, rsel, ya & 1);
#pragma unroll
            for (int k = 0; k < 42; ++k) stage_slot(g1, nx, k, 2 + rsel, ya & 1);
#pragma unroll
            for (int i = 0; i < 17; ++i) asm volatile("" :: "v"(rg[i]));          // (wait for the first step's registers here: wgrad6_kernel)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(rx[i]));
        }
        w6g_barrier();
#ifdef WG6_PROF
        pc_head += __builtin_readcyclecounter() - th0;
#endif
        int s0 = 0;                                           // ring slot of g row 2 y
        for (int y = ya; y < yb; ++y) {
#ifdef WG6_PROF
            const unsigned long long tm0 = __builtin_readcyclecounter();
#endif
            load_g(ng, bb, cx, 2 * y + 5 + rsel);             // registers of the step after next
            load_x(nx, bb, cx, y + 2);
            int wslot = s0 + 3 + rsel; wslot -= wslot >= NGR ? NGR : 0;
            int a_base[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) { int sl = s0 + ky; sl -= sl >= NGR ? NGR : 0; a_base[ky] = sl * GI + a_elem; }
            const int b_base = b_elem + (y & 1) * XI2;
            bf16x8 av[2][3], bx[2][3];
            auto rd_a = [&](int tp, int pc) {
                const int ky = tp / 3, kx = tp % 3;
                av[tp & 1][pc] = __builtin_bit_cast(bf16x8, lds[a_base[ky] + (pc * 3 + kx) * 2 * TC]);
            };
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                bx[0][pc] = __builtin_bit_cast(bf16x8, lds[b_base + pc * 2 * TC]);
                bx[1][pc] = __builtin_bit_cast(bf16x8, lds[b_base + XI + pc * 2 * TC]);
                rd_a(0, pc);
            }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int qq = 0; qq < 6; ++qq) {
                        acc[j][tp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[tp & 1][PA[qq]], bx[j][PB[qq]], acc[j][tp], 0, 0, 0);
                        if (tp + 1 < 9 && j == 0 && qq < 3) rd_a(tp + 1, qq);          // (the other operand slot: last used by tap tp - 1)
                        // the 42-slot staging program behind every other MFMA of the first 84 (108 per step)
                        const int k = tp * 12 + j * 6 + qq;
                        if ((k & 1) == 0) stage_slot(rg, rx, k >> 1, wslot, (y + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
#ifdef WG6_PROF
            const unsigned long long tm1 = __builtin_readcyclecounter();
#endif
#pragma unroll
            for (int i = 0; i < 17; ++i) rg[i] = ng[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) rx[i] = nx[i];
            s0 += 2; s0 -= s0 >= NGR ? NGR : 0;
            w6g_barrier();
#ifdef WG6_PROF
            { const unsigned long long tm2 = __builtin_readcyclecounter(); pc_mult += tm1 - tm0; pc_bar += tm2 - tm1; ++nstep_done; }
#endif
        }
    }
    }
#ifdef WG6_PROF
    {
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (lane == 0 && lin < 2048) {
            unsigned long long* d = te_wgrad6_prof_buf + ((size_t)lin * 4 + wid) * 4;
            d[0] = pc_mult; d[1] = pc_head; d[2] = pc_bar;
            d[3] = ((unsigned long long)nstep_done << 40) | ((__builtin_readcyclecounter() - pstart) & 0xFFFFFFFFFFull);
        }
    }
#endif

    float* sl = p.slabs + ((size_t)bgrp * p.S + s_chunk) * p.Co * p.Ci * 9;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = ci0 + j * TC + wci * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float* dst = sl + ((size_t)co * p.Ci + ci) * 9;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) dst[tp] = acc[j][tp][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- 1x1 kind (round 6)
// TE_CONV_1X1 (the weight gradient of the discriminator's ResBlock skip convolutions, model_spatial_query.py:173-181 / :780-798):
//     slab[b][s][co][ci] = sum over the cells of chunk s of sample b   g[b, co, cell] * x[b, ci, cell]
// One tap: six piece products per 16 cells and 32 x 32 channels, so every staged element is used for 1/9 of the MFMAs it feeds in the
// 3x3 kinds, and the staging decides the shape: block tile 128 x 128 channels, 4 waves (one per SIMD, as the other kernels of this
// file), wave (wco, wci) owns 64 x 64 = 2 x 2 sub-tiles (64 accumulator registers): 24 MFMAs per step of 16 cells, fed by 6 + 6
// operand reads, while the wave splits one g unit and one x unit (64 channels x 8 cells each = 4 packed pairs x 3 split steps: 24
// slots, one behind every MFMA).  Both tensors are plain H x W planes: no components, no ring - g and x rows are
// [piece 3][k half 2][channel 64] images (6 KB) per 64-channel group, double-buffered (48 KB of LDS in all).  Same step order, chunk
// mapping, prefetch distance (the loads of step y + 2 at the head of step y) and slab layout as wgrad6t_kernel; the fp32 kernel
// (wgrad_mfma_kernel<TE_CONV_1X1>) ran these launches at 70 - 77 TFLOP/s, the HBM floor (both tensors read once) is a third of its time.
constexpr int PI = 3 * 2 * TC;                // elements of one 64-channel row image: 384 (6 KB)

__global__ __launch_bounds__(WT, 1) void wgrad6p_kernel(const Wg6Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem_raw);          // [side: g, x][buffer 2][group 2][PI]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wid >> 1, wci = wid & 1;                  // multiplying: the wave's 64 x 64 quarter of the block tile
    const int grp = wid >> 1, q = wid & 1;                    // staging: channel group `grp` of g AND of x, cells 8 q .. 8 q + 7 of the step; lane = channel

    const Wg6Chunk ck = wg6_chunk(p.S, p.NB * (p.W / 16));
    const int s_chunk = ck.s, bgrp = ck.bgrp, b = bgrp * p.NB;
    const int co0 = blockIdx.y * 2 * TC, ci0 = blockIdx.z * 2 * TC;
    const size_t plane = (size_t)p.H * p.W;
    const float* gblk = p.g + ((size_t)b * p.Co + co0) * plane;
    const float* xblk = p.x + ((size_t)b * p.Ci + ci0) * plane;
    const unsigned ch_lane = (unsigned)((grp * TC + lane) * plane * 4);        // bytes (host-checked < 2 GiB)

    f32x16 acc[2][2];                                         // [co sub-tile][ci sub-tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // rows past the end are clamped to the last row (their registers are never used)
    auto load8 = [&](float (&r8)[8], const float* blk, int C, int bb, int cx, int row) {
        const char* src = reinterpret_cast<const char*>(blk + (size_t)bb * C * plane + (size_t)min(row, p.H - 1) * p.W + 16 * cx + 8 * q);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4u*>(src + 16 * k + ch_lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) r8[4 * k + e] = v[e];
        }
    };
    // split of one packed pair in three steps (wgrad6t_kernel: unit_step); units 0..3: the g unit, 4..7: the x unit
    unsigned pg[3][4], px[3][4];
    float v0 = 0.f, v1 = 0.f, f0 = 0.f, f1 = 0.f;
    auto unit_step = [&](const float (&rg)[8], const float (&rx)[8], int u, int step) {
        unsigned& d0 = u < 4 ? pg[0][u] : px[0][u - 4];
        unsigned& d1 = u < 4 ? pg[1][u] : px[1][u - 4];
        unsigned& d2 = u < 4 ? pg[2][u] : px[2][u - 4];
        if (step == 0) {
            if (u < 4) { v0 = rg[2 * u]; v1 = rg[2 * u + 1]; }
            else { v0 = rx[2 * (u - 4)]; v1 = rx[2 * (u - 4) + 1]; }
            const f32x2 t = {v0, v1};
            const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
            d0 = h;
            f0 = __builtin_bit_cast(float, h << 16);
            f1 = __builtin_bit_cast(float, h & 0xFFFF0000u);
        } else if (step == 1) {
            v0 -= f0; v1 -= f1;
            const f32x2 t = {v0, v1};
            const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
            d1 = m;
            f0 = __builtin_bit_cast(float, m << 16);
            f1 = __builtin_bit_cast(float, m & 0xFFFF0000u);
        } else {
            v0 -= f0; v1 -= f1;
            const f32x2 t = {v0, v1};
            d2 = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
        }
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(f0), "+v"(f1));
    };
    const int w_elem = grp * PI + q * TC + lane;              // + side * 4 PI + buffer * 2 PI + piece * 2 TC
    auto write_g = [&](int buf, int pc) {
        u32x4 v; v[0] = pg[pc][0]; v[1] = pg[pc][1]; v[2] = pg[pc][2]; v[3] = pg[pc][3];
        lds[buf * 2 * PI + pc * 2 * TC + w_elem] = v;
    };
    auto write_x = [&](int buf, int pc) {
        u32x4 v; v[0] = px[pc][0]; v[1] = px[pc][1]; v[2] = px[pc][2]; v[3] = px[pc][3];
        lds[4 * PI + buf * 2 * PI + pc * 2 * TC + w_elem] = v;
    };
    // the staging program: slots 0..23 = 8 units x 3 steps; the writes of a side in the slot of its last step
    auto stage_slot = [&](const float (&rg)[8], const float (&rx)[8], int k, int buf) {
        if (k < 24) unit_step(rg, rx, k / 3, k % 3);
        if (k == 11) { write_g(buf, 0); write_g(buf, 1); write_g(buf, 2); }
        if (k == 23) { write_x(buf, 0); write_x(buf, 1); write_x(buf, 2); }
    };

    const int tiles_x = p.W / 16;
    const int sps = tiles_x * p.H, n_steps = sps * p.NB;
    const int a_elem = wco * PI + half * TC + l31;                            // + sub-tile * 32 + buffer * 2 PI + piece * 2 TC
    const int b_elem = 4 * PI + wci * PI + half * TC + l31;

    float rg[8], rx[8], ng[8], nx[8];
    for (int sg = 0; sg < ck.nseg; ++sg) {
    int t, t_end;
    wg6_segment(ck, sg, p.S, p.H, n_steps, t, t_end);
    while (t < t_end) {
        const int bb = t / sps, rem = t - bb * sps, cx = rem / p.H, ya = rem - cx * p.H;
        const int n = min(p.H - ya, t_end - t), yb = ya + n;
        t += n;
        // ---- head of a sweep: row ya into buffer ya & 1, registers of the next step
        {
            float g0[8], x0[8];
            load8(g0, gblk, p.Co, bb, cx, ya);
            load8(x0, xblk, p.Ci, bb, cx, ya);
            load8(rg, gblk, p.Co, bb, cx, ya + 1);
            load8(rx, xblk, p.Ci, bb, cx, ya + 1);
#pragma unroll
            for (int k = 0; k < 24; ++k) stage_slot(g0, x0, k, ya & 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) { asm volatile("" :: "v"(rg[i])); asm volatile("" :: "v"(rx[i])); }       // (wait for the first step's registers here)
        }
        w6g_barrier();
        for (int y = ya; y < yb; ++y) {
            load8(ng, gblk, p.Co, bb, cx, y + 2);             // registers of the step after next
            load8(nx, xblk, p.Ci, bb, cx, y + 2);
            const int cur = (y & 1) * 2 * PI, nxt = (y + 1) & 1;
            bf16x8 av[2][3], bv[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    av[i][pc] = __builtin_bit_cast(bf16x8, lds[cur + a_elem + i * 32 + pc * 2 * TC]);
                    bv[i][pc] = __builtin_bit_cast(bf16x8, lds[cur + b_elem + i * 32 + pc * 2 * TC]);
                }
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int qq = 0; qq < 6; ++qq) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][PA[qq]], bv[j][PB[qq]], acc[i][j], 0, 0, 0);
                        stage_slot(rg, rx, (i * 2 + j) * 6 + qq, nxt);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
            for (int i = 0; i < 8; ++i) { rg[i] = ng[i]; rx[i] = nx[i]; }
            w6g_barrier();
        }
    }
    }

    float* sl = p.slabs + ((size_t)bgrp * p.S + s_chunk) * p.Co * p.Ci;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wci * TC + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * TC + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                sl[(size_t)co * p.Ci + ci] = acc[i][j][r];
            }
        }
}

// on by default; TE_SPLIT_WGRAD=0 (or TE_SPLIT_BF16=0, the switch of all split kernels) keeps the fp32 kernel - A/B measurements
std::atomic<int> g_wg6_on{[] {
    const char* e = getenv("TE_SPLIT_WGRAD");
    if (e) return atoi(e) ? 1 : 0;
    const char* a = getenv("TE_SPLIT_BF16");
    return (a && atoi(a) == 0) ? 0 : 1;
}()};

// TE_SPLIT_WGRAD_T2=0: the transposed kind alone stays on the fp32 kernel (A/B measurements)
std::atomic<int> g_wg6t_on{[] { const char* e = getenv("TE_SPLIT_WGRAD_T2"); return (e && atoi(e) == 0) ? 0 : 1; }()};
// TE_SPLIT_WGRAD_1X1=0: the 1x1 kind alone stays on the fp32 kernel (A/B measurements)
std::atomic<int> g_wg6p_on{[] { const char* e = getenv("TE_SPLIT_WGRAD_1X1"); return (e && atoi(e) == 0) ? 0 : 1; }()};

}  // namespace

#ifdef WG6_PROF
extern "C" int te_debug_wgrad6_prof(void* host_dst, int64_t bytes) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(te_wgrad6_prof_buf), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
#endif

// process-wide switch (0 / 1; anything else only queries): does te_wgrad_f32 / te_wgrad_group_f32 take this kernel where it applies?
extern "C" int te_wgrad_split_bf16(int on) {
    const int old = g_wg6_on.load(std::memory_order_relaxed);
    if (on == 0 || on == 1) g_wg6_on.store(on, std::memory_order_relaxed);
    return old;
}

// form of the transposed-kind kernel: 1 (default) = the wide form (64 x 128 channels per block) where Ci % 128 == 0, 0 = 64 x 64 everywhere;
// bit-identical slabs.  A process-wide A/B switch (te_hip.h); TE_WGRAD_T2_WIDE in the environment sets the initial value.
static std::atomic<int> g_wg6t_wide{[] { const char* e = getenv("TE_WGRAD_T2_WIDE"); return e ? atoi(e) : 1; }()};
extern "C" int te_wgrad_t2_wide(int on) {
    const int old = g_wg6t_wide.load(std::memory_order_relaxed);
    if (on == 0 || on == 1) g_wg6t_wide.store(on, std::memory_order_relaxed);
    return old;
}

extern "C" int te_wgrad_split_supported(int kind, int Co, int Ci, int H, int W) {
    if (!(Co > 0 && Ci > 0 && H > 0)) return 0;
    if (kind == TE_CONV_3X3) {
        if (!(W >= 32 && W % 32 == 0)) return 0;
        if (Co % TC == 0 && Ci % TC == 0) return 1;
        return (Co == 32 && Ci == 32) ? 2 : 0;              // 2: the sample-pair form (needs an even batch and per-sample slabs)
    }
    if (kind == TE_CONV_T2) {
        if (!(W >= 16 && W % 16 == 0)) return 0;
        return (Co % 32 == 0 && Ci % 32 == 0 && Co + Ci >= 96) ? 1 : 0;      // (32 x 32: a quarter of the tile - stays on the fp32 kernel)
    }
    if (kind == TE_CONV_1X1)                                                  // round 6: wgrad6p_kernel, 128 x 128 channels per block
        return (W >= 16 && W % 16 == 0 && Co % (2 * TC) == 0 && Ci % (2 * TC) == 0) ? 1 : 0;
    return 0;
}

// returns 1 when the launch was taken, 0 when the caller has to use the fp32 kernel, < 0 on error
int te_wgrad6_launch(float* slabs, const float* g, const float* x, int kind, int B, int Co, int Ci, int H, int W, int S, int NB, hipStream_t s) {
    const int sup = te_wgrad_split_supported(kind, Co, Ci, H, W);
    if (!g_wg6_on.load(std::memory_order_relaxed) || !sup) return 0;
    if (((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(x)) & 15) != 0) return 0;
    Wg6Args a{};
    a.slabs = slabs; a.g = g; a.x = x; a.B = B; a.Co = Co; a.Ci = Ci; a.H = H; a.W = W; a.S = S; a.NB = NB; a.tiles_x = W / 32;
    dim3 grid((unsigned)(B / NB * S), (unsigned)te::cdiv(Co, TC), (unsigned)te::cdiv(Ci, TC));
    if (kind == TE_CONV_1X1) {
        if (!g_wg6p_on.load(std::memory_order_relaxed)) return 0;
        if ((int64_t)NB * std::max(Co, Ci) * H * W * 4 >= (int64_t)OOBW) return 0;           // 32-bit byte offsets inside a sample group
        static std::atomic<uint64_t> attr_done_p1{0};
        te::allow_big_lds(attr_done_p1, (const void*)wgrad6p_kernel, 160 * 1024);
        wgrad6p_kernel<<<dim3(grid.x, (unsigned)(Co / (2 * TC)), (unsigned)(Ci / (2 * TC))), WT, (size_t)8 * PI * 16, s>>>(a);
        return 1;
    }
    if (kind == TE_CONV_T2) {
        if (!g_wg6t_on.load(std::memory_order_relaxed)) return 0;
        // 32-bit byte offsets inside a sample group: lane * plane * 4 over the 64 channels of a tile, for the (2H+1) x (2W+1) tensor and
        // for the H x W one (the fp32 kernel takes the launch otherwise, as for the 3x3 kind below)
        if ((int64_t)NB * std::max(Co, Ci) * (2 * (int64_t)H + 1) * (2 * (int64_t)W + 1) * 4 >= (int64_t)OOBW) return 0;
        const size_t lds = (size_t)(NGR * GI + 2 * XI) * 16;
        if (Co % TC == 0 && Ci % (2 * TC) == 0 && g_wg6t_wide.load(std::memory_order_relaxed)) {      // wide form: 64 x 128 channels per block
            static std::atomic<uint64_t> attr_done_tw{0};
            te::allow_big_lds(attr_done_tw, (const void*)wgrad6tw_kernel, 160 * 1024);
            wgrad6tw_kernel<<<dim3(grid.x, grid.y, (unsigned)(Ci / (2 * TC))), WT, (size_t)(NGR * GI + 4 * XI) * 16, s>>>(a);
            return 1;
        }
        if (Co % TC == 0 && Ci % TC == 0) {
            static std::atomic<uint64_t> attr_done_t{0};
            te::allow_big_lds(attr_done_t, (const void*)wgrad6t_kernel<false>, 160 * 1024);
            wgrad6t_kernel<false><<<grid, WT, lds, s>>>(a);
        } else {
            static std::atomic<uint64_t> attr_done_tn{0};
            te::allow_big_lds(attr_done_tn, (const void*)wgrad6t_kernel<true>, 160 * 1024);
            wgrad6t_kernel<true><<<grid, WT, lds, s>>>(a);
        }
        return 1;
    }
    if (sup == 2) {                                       // sample-pair form
        if (NB != 1 || B % 2 != 0) return 0;
        if ((int64_t)2 * std::max(Co, Ci) * H * W * 4 >= (int64_t)OOBW) return 0;
        static std::atomic<uint64_t> attr_done_p{0};
        te::allow_big_lds(attr_done_p, (const void*)wgrad6_kernel<true>, 160 * 1024);
        wgrad6_kernel<true><<<dim3((unsigned)(B / 2 * S), 1, 1), WT, (size_t)N_IMG * IMG * 16, s>>>(a);
        return 1;
    }
    if ((int64_t)NB * std::max(Co, Ci) * H * W * 4 >= (int64_t)OOBW) return 0;
    static std::atomic<uint64_t> attr_done{0};
    te::allow_big_lds(attr_done, (const void*)wgrad6_kernel<false>, 160 * 1024);
    wgrad6_kernel<false><<<grid, WT, (size_t)N_IMG * IMG * 16, s>>>(a);
    return 1;
}
