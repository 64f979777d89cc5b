// S2 on the bf16 matrix pipe (round 5, kind TE_CONV_S2S6): the 3x3 / stride 2 / pad 0 convolution  in [B,K,2H+1,2W+1] -> out [B,M,H,W]
// - the discriminator's down-sampling convolutions (model_spatial_query.py:765-779, ConvLayer(downsample=True) after its blur) and the
// data gradient of the generator's up-sampling layers (the adjoint of conv_transpose2d(stride 2), :318) - with every fp32 operand split
// into three bf16 pieces and six exact piece products per multiply-add accumulated in fp32, exactly as wino6.hip does for the stride-1
// layers (same split, same product order "small terms first", fp32-equivalent results: tests/test_gpu_s2s6.py pins it at the 5e-6 bar
// of the fp32 kernels against fp64).  No Winograd form here (stride 2): 9 taps x 6 = 54 MFMAs of v_mfma_f32_32x32x16_bf16 per 16
// input channels and 32 x 32 output tile, against 72 fp32 MFMAs of four times the duration each in conv_mfma_kernel<TE_CONV_S2>.
//
// Structure = the ping-pong form of wino6.hip (read its header first):
//   * block 512 threads, tile 64 output channels x 8 rows x 16 columns; two HALF tiles of 4 output rows (9 input rows x 33 columns
//     each), waves 0-3 own half 0, waves 4-7 half 1, wave (wm, wrl) of a group = 32 channels x rows {2 wrl, 2 wrl + 1} x 16 columns
//     (ONE accumulator tile); the groups run half a stage apart, so every SIMD has one wave feeding the matrix pipe while its
//     partner moves the next half tile to LDS, renews half of the weight image by LDS-DMA and fetches the stage after;
//   * the staging arithmetic (style scale, three-piece split of 2.5 items x 4 positions x 2 channels per thread) is a 51-slot
//     program behind the multiplying wave's own MFMAs (54 per stage); results wait in 36 registers for the staging phase;
//   * weights packed once per optimiser step in MFMA fragment order  S6[K/16][piece][tap][M/32][64 lanes][8 bf16]
//     (TE_PACK_S6FWD: M = Co, K = Ci; TE_PACK_S6SWAP: M = Ci, K = Co - the launch as data gradient of the transposed kind): a stage is
//     54 KB = 54 fragment slots; taps 0-4 form the half that is renewed behind the partner's mid-phase barrier, taps 5-8 the other;
//   * input tile in LDS as T[piece][row 9][k half][column parity][17 (+1 pad)][8 bf16]: a tap (ky, kx) reads row 2 r + ky, parity
//     kx & 1, index c + (kx >> 1) - one conflict-free ds_read_b128 per piece (the pad makes two input rows a multiple of 16 chunks).
// Every input element of the tile is inside the image (2 (H - 1) + 2 = 2 H: no padding, no edge cases); the epilogue is the direct
// kernel's (demodulation scale, bias, leaky ReLU, residual, mask).
#include "conv_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int WT = 512, GT = 256, KC = 16, BM = 64;
constexpr int TH = 8, TWO = 16, PH = 4;                       // output tile 8 x 16, half tile 4 rows
constexpr int IR = 2 * PH + 1, IC = 2 * TWO + 1;              // input rows / columns of a half tile: 9 x 33
constexpr int CI = 18;                                        // chunks per (row, k half, parity): 17 used + 1 pad
constexpr int NTAP = 9;
constexpr int U_SLOTS = 3 * NTAP * 2;                         // 54 fragment slots of 1 KB
constexpr int TP_PLANE = IR * 2 * 2 * CI * 4;                 // dwords of one piece of a half tile: 2 592
constexpr int TP_DWORDS = 3 * TP_PLANE;                       // 7 776 dwords = 30.4 KB
constexpr int NG = (IC + 3) / 4;                              // 4-column groups per input row: 9 (the last one holds column 32 only)
constexpr int N_ITEMS = IR * NG * 8;                          // (row, column group, channel pair) items of a half: 648
constexpr int P_IN = (N_ITEMS + GT - 1) / GT;                 // 3 per thread (the last one bounds-checked)
constexpr int N_SLOT = P_IN + 16 * P_IN;                      // arithmetic slots: 51 (<= 54 MFMAs of a phase)
constexpr int SLOT0 = NTAP * 6 - N_SLOT;                      // the program starts behind MFMA 3
constexpr int UA_TAPS = 5;                                    // taps 0-4: weight half a (30 slots), taps 5-8: half b (24 slots)

#ifdef S2_PROF       // experimental builds: per-wave cycle counts of the phases, read back with te_debug_s2s6_prof (tools/s2s6_phase_prof.py)
__device__ unsigned long long te_s2s6_prof_buf[2048 * 8 * 8];
#define S2_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define S2_ACC(i, a, b) pc[i] += (b) - (a)
#else
#define S2_T(v)
#define S2_ACC(i, a, b)
#endif

#ifndef DMA_PRIO
#define DMA_PRIO 0         // experiment: the staging wave raises its priority while it issues the weight DMA (wino6.hip)
#endif

struct S2Args {
    float* out; const float* in; const u32x4* U; const float* isc; const float* osc; const float* bias; const float* res;
    const float* mref; float mgain; int act;
    int B, K, M, H, W, Hi, Wi, ntiles, mblocks, tiles_x, tiles_y, nt8;
};

__device__ __forceinline__ void s2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void s2_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ISC: the launch carries style scales (a template parameter: `if (p.isc)` around the scale loads was one of the conditional updates of
// the fetch registers that made the compiler keep two copies of them - see fetch_item and profiles/experiments/r05_s2s6_phase_profile.log)
template <bool ISC>
__global__ __launch_bounds__(WT, 2) void s2s6_kernel(const S2Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* ul = reinterpret_cast<u32x4*>(smem_raw);                                   // weights, 16-byte chunks
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform by construction: keep the role branches scalar
    const int grp = wid >> 2, wq = wid & 3, wm = wq >> 1, wrl = wq & 1, gt = tid & (GT - 1);
    unsigned* tl = reinterpret_cast<unsigned*>(smem_raw + U_SLOTS * 1024) + grp * TP_DWORDS;      // this group's half tile
    const u32x4* tl4 = reinterpret_cast<const u32x4*>(tl);
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int tq = jx / p.mblocks, mb = jx % p.mblocks;
    const int tile = p.nt8 ? (int)(((int64_t)xcd * p.ntiles) >> 3) + tq : tq * 8 + xcd;
    if (tile >= (p.nt8 ? (int)(((int64_t)(xcd + 1) * p.ntiles) >> 3) : p.ntiles)) return;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx * TWO, y0 = ty * TH, yh = y0 + PH * grp;
    const size_t iplane = (size_t)p.Hi * p.Wi, oplane = (size_t)p.H * p.W;
    const float* inb = p.in + (size_t)b * p.K * iplane;
    const float* iscb = ISC ? p.isc + (size_t)b * p.K : nullptr;

    // two accumulators: `acc` takes the h x h products (the bulk of every multiply-add), `accl` the five small ones (together 2^-8
    // of it).  All 54 MFMAs of a stage into one register would round the running sum 54 times per stage where the fp32 kernel rounds
    // it 72 times at K = 2 per instruction with the same error per rounding - measured 3.3x the fp32 kernel's deviation from fp64 at
    // 512 channels; with the small terms kept apart the big sum is rounded 9 times per stage (and the small one at 2^-8 of the scale)
    f32x16 acc, accl;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accl[r] = 0.f; }

    // staging geometry of this group's half: item e = gt + 256 i -> (column group cg = e % 9, channel pair q = (e / 9) % 8, row = e / 72);
    // the item covers input columns 2 x0 + 4 cg .. + 3 of input row 2 yh + row, channels 2 q and 2 q + 1 of the stage
    unsigned g_off[P_IN];
    int l_off[P_IN];
    bool last_col[P_IN], live[P_IN];
    unsigned q2[P_IN];
#pragma unroll
    for (int i = 0; i < P_IN; ++i) {
        const int e = gt + GT * i;
        live[i] = e < N_ITEMS;
        const int ee = live[i] ? e : 0;
        // (column group fastest over the lanes: up to three lanes of an LDS write group share a bank, but the fetch reads 144-byte runs;
        //  with the pair-in-chunk fastest - conflict-light writes - the kernel measured 4 - 5 % SLOWER: 140 / 150 / 158 against 146 / 156 / 166)
        const int cg = ee % NG, q = (ee / NG) & 7, row = ee / (NG * 8);
        last_col[i] = cg == NG - 1;
        // (the last group holds column 2 x0 + 32 only, and the three behind it may lie behind the end of the tensor: its four floats are
        //  loaded three columns EARLY - always inside the row - and element 3 becomes position 0 when the item is scaled; an item behind
        //  the end of the list repeats item 0's loads and is never written: no divergent loads, one fixed instruction count per wave)
        g_off[i] = (unsigned)((2 * q * p.Hi + 2 * yh + row) * p.Wi + 2 * x0 + 4 * cg - (last_col[i] ? 3 : 0));
        // position j of the item: parity j & 1, index 2 cg + (j >> 1); dword = channel pair within its k half
        l_off[i] = (((row * 2 + (q >> 2)) * 2) * CI + 2 * cg) * 4 + (q & 3);          // + (j & 1) * CI * 4 + (j >> 1) * 4 + piece * TP_PLANE
        q2[i] = 2u * q;
    }
    const int MT = p.M >> 5;
    f32x4 rin[P_IN][2];
    f32x2 rsc[P_IN];               // style scales of the item's channel pair
#pragma unroll
    for (int i = 0; i < P_IN; ++i) rsc[i] = f32x2{1.f, 1.f};
    const int nstage = p.K / KC;
    // fetch of stage s, item by item.  Inside the loop it is issued by the MULTIPLYING role, each item right behind the last slot of the
    // arithmetic that reads its registers, so that the fetch registers are written and read in one role only (wino6.hip, fetch_item)
    auto fetch_scales = [&](int s) {
        if (ISC) {
#pragma unroll
            for (int i = 0; i < P_IN; ++i) rsc[i] = *reinterpret_cast<const f32x2u*>(iscb + s * KC + q2[i]);
        }
    };
    auto fetch_item = [&](int i, int s) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
            rin[i][h2] = *reinterpret_cast<const f32x4u*>(inb + ((size_t)s * KC + h2) * iplane + g_off[i]);
    };
    auto issue = [&](int s) {
        fetch_scales(s);
#pragma unroll
        for (int i = 0; i < P_IN; ++i) fetch_item(i, s);
    };
    // weight half `uh` of stage s: uh = 0: taps 0-4 (30 slots: 8 / 8 / 7 / 7 per wave), uh = 1: taps 5-8 (24 slots: 6 per wave)
    auto issue_u = [&](int uh, int s) {
        const u32x4* us = p.U + (size_t)s * 27 * MT * 64;
        const int ntap = uh ? NTAP - UA_TAPS : UA_TAPS, tap0 = uh ? UA_TAPS : 0, n = ntap * 6;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int j = wq + 4 * r;                                          // slot of this half: (piece, tap, M tile)
            if (j >= n) break;
            const int piece = j / (ntap * 2), rem = j % (ntap * 2), tap = tap0 + (rem >> 1), mt = rem & 1;
            const int pt = piece * NTAP + tap;
            const u32x4* g = us + ((size_t)pt * MT + 2 * mb + mt) * 64 + (unsigned)lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(ul + (pt * 2 + mt) * 64), 16, 0, 0);
        }
    };
    // ---- the staging arithmetic as a program of 51 slots behind the MFMAs of the multiplying phase (wino6.hip):
    //   slots 0..2: style scale of item k;  slots 3 + 4 u + j: unit u = item * 4 + position, the four steps of the three-piece split
    unsigned res[P_IN][4][3];
    float te = 0.f, to = 0.f, fe = 0.f, fo = 0.f;
    auto arith = [&](int k) {
        if (k < 0) {
        } else if (k < P_IN) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 v = rin[k][h2];
                if (last_col[k]) v[0] = v[3];                 // (loaded three columns early; positions 1..3 are never written)
                rin[k][h2] = ISC ? v * rsc[k][h2] : v;
                asm volatile("" : "+v"(rin[k][h2]));
            }
        } else if (k < N_SLOT) {
            const int u = (k - P_IN) >> 2, j = (k - P_IN) & 3, i = u >> 2, c = u & 3;
            if (j == 0) {
                te = rin[i][0][c]; to = rin[i][1][c];                              // even / odd channel of the pair at position c
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
            asm volatile("" : "+v"(te), "+v"(to), "+v"(fe), "+v"(fo));        // (pin the step here: see wino6.hip)
        }
    };
    auto write_res = [&]() {
#pragma unroll
        for (int i = 0; i < P_IN; ++i) {
            if (!live[i]) continue;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c > 0 && last_col[i]) continue;                      // (positions 33..35 of the row do not exist in the tile)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) tl[l_off[i] + (c & 1) * CI * 4 + (c >> 1) * 4 + pc * TP_PLANE] = res[i][c][pc];
            }
        }
    };
    const int rr = l31 >> 4, jj = l31 & 15;
    // B operand of tap (ky, kx), piece pc: chunk ((2 (2 wrl + rr) + ky) * 2 + half) * 2 + (kx & 1)) * CI + jj + (kx >> 1) + pc * TP_PLANE / 4
    const int b_chunk = ((2 * (2 * wrl + rr) * 2 + half) * 2) * CI + jj;
    const int a_chunk = wm * 64 + lane;                                 // + (piece * 9 + tap) * 128

    // prologue: every group splits and writes its half of stage 0 and fetches stage 1; group 0 brings in the whole weight image
    issue(0);
    if (grp == 0) { issue_u(0, 0); issue_u(1, 0); }
#pragma unroll
    for (int k = 0; k < N_SLOT; ++k) arith(k);
    write_res();
    issue(1);
    if (grp == 0) {          // (counted: only the weight DMA must have landed; the fetch of stage 1 - 9 loads, 6 without style scales - stays in flight)
        if (ISC) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    s2_barrier();
    const int nphase = 2 * nstage;
#ifdef S2_PROF
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long pstart = __builtin_readcyclecounter(), rstart = __builtin_amdgcn_s_memrealtime();
#endif
    for (int ph = 0; ph < nphase; ++ph) {
        const bool last = ph == nphase - 1;
        S2_T(t0);
        if ((ph & 1) == grp) {
            // ---- multiply this group's half of stage ph / 2; behind the MFMAs: the arithmetic of the stage after (rin -> res)
            bf16x8 av[2][3], bv[2][3];
            const int fs2 = min((ph >> 1) + 2, nstage - 1);
            auto rd1 = [&](int t, int slot, int q) {
                const int ky = t / 3, kx = t % 3;
                if (q < 3) av[slot][q] = __builtin_bit_cast(bf16x8, ul[a_chunk + (q * NTAP + t) * 128]);
                else bv[slot][q - 3] = __builtin_bit_cast(bf16x8, tl4[b_chunk + ((ky * 2) * 2 + (kx & 1)) * CI + (kx >> 1) + (q - 3) * (TP_PLANE / 4)]);
            };
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
            for (int q = 0; q < 6; ++q) rd1(0, 0, q);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int slot = t & 1;
#ifdef S2_PROF
                if (t == UA_TAPS - 1) { S2_T(ta); s2_barrier(); S2_T(tb); S2_ACC(0, t0, ta); S2_ACC(1, ta, tb); pc[2] -= tb; }
#else
                if (t == UA_TAPS - 1) s2_barrier();    // mid-phase barrier: in front of tap 4's MFMAs (operands read) and of the first read of half b
#endif
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
#ifndef ST_SKIP_MFMA     // (experiment switches ST_*: timing decomposition only, results are wrong)
                    if (q < 5) accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][PA[q]], bv[slot][PB[q]], accl, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][PA[q]], bv[slot][PB[q]], acc, 0, 0, 0);
#endif
                    if (t + 1 < NTAP && q < 3) { rd1(t + 1, slot ^ 1, 2 * q); rd1(t + 1, slot ^ 1, 2 * q + 1); }
#ifndef ST_NO_ARITH
                    arith(t * 6 + q - SLOT0);
#endif
#ifndef ST_NO_FETCH
                    {   // the fetch of the stage after next: an item's loads right behind the last slot that reads its registers
                        const int k = t * 6 + q - SLOT0;
                        if (k == P_IN - 1) fetch_scales(fs2);
#pragma unroll
                        for (int i = 0; i < P_IN; ++i)
                            if (k == P_IN + 16 * i + 12) fetch_item(i, fs2);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
#ifdef S2_PROF
            { asm volatile("s_nop 0" ::: "memory"); S2_T(tc); pc[2] += tc; }
#endif
        } else {
            // ---- stage: move this group's half of stage cs = (ph + 1) / 2 to LDS, fetch stage cs + 1, renew a half of the weight image
            const int cs = (ph + 1) >> 1;
            const bool work = cs >= 1 && cs < nstage;
            // group 1 renews weight half b in front of the mid-phase barrier (the partner reads it right behind): DMA first, the LDS
            // writes of this half tile in its shadow, then the wait for the DMA; group 0 renews half a behind the barrier.  The LDS
            // writes are unconditional: in group 1's first phase they repeat what the prologue wrote, in group 0's last phase they
            // put stale results in a tile nobody reads any more.
#ifndef ST_NO_DMA
            if (DMA_PRIO) __builtin_amdgcn_s_setprio(DMA_PRIO);
            if (work && grp == 1) issue_u(1, cs);
            if (DMA_PRIO) __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifndef ST_NO_DSW
            write_res();
#endif
            if (work && grp == 1) s2_wait_vm();
            S2_T(ta);
            s2_barrier();
            S2_T(tb);
#ifndef ST_NO_DMA
            if (work && grp == 0) {
                if (DMA_PRIO) __builtin_amdgcn_s_setprio(DMA_PRIO);
                issue_u(0, cs);
                if (DMA_PRIO) __builtin_amdgcn_s_setprio(0);
                s2_wait_vm();
            }
#endif
            S2_T(tc);
            S2_ACC(3, t0, ta); S2_ACC(4, ta, tb); S2_ACC(5, tb, tc);
        }
        S2_T(t8);
        if (!last) s2_barrier();
        S2_T(t9);
        S2_ACC(6, t8, t9);
    }
#ifdef S2_PROF
    if (lane == 0 && blockIdx.x < 2048) {
        unsigned long long* d = te_s2s6_prof_buf + ((size_t)blockIdx.x * 8 + wid) * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = pc[i];
        d[6] = pc[6] | ((__builtin_amdgcn_s_memrealtime() - rstart) << 40);
        d[7] = ((unsigned long long)nstage << 48) | ((__builtin_readcyclecounter() - pstart) & 0xFFFFFFFFFFFFull);
    }
#endif
    // epilogue: the direct kernel's stages (demodulation scale, bias, leaky ReLU, residual, mask)
    const int mbase = mb * BM + wm * 32;
    const size_t off0 = ((size_t)b * p.M + mbase) * oplane + (size_t)(yh + 2 * wrl + rr) * p.W + x0 + jj;
    const float g_pos = p.act == 3 ? 1.4142135623730951f : 1.f;
    // (scales, biases, residual and mask rows of this lane are loaded up front: inside the store loop every load would wait behind the
    //  previous row's store - the compiler cannot prove `out` does not alias them)
    float scv[16], biv[16], resv[16], mrefv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
        scv[r] = p.osc ? p.osc[(size_t)b * p.M + mbase + dm] : 1.f;
        biv[r] = p.bias ? p.bias[mbase + dm] : 0.f;
    }
    if (p.res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) resv[r] = p.res[off0 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * oplane];
    }
    if (p.mref) {
#pragma unroll
        for (int r = 0; r < 16; ++r) mrefv[r] = p.mref[off0 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * oplane];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = (acc[r] + accl[r]) * scv[r] + biv[r];
        if (p.act >= 3) v = (v > 0.f ? v : v * 0.2f) * g_pos;
        if (p.res) v += resv[r];
        if (p.mref) v *= mrefv[r] > 0.f ? p.mgain : 0.2f * p.mgain;
        p.out[off0 + (size_t)dm * oplane] = v;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: TWO 64-channel weight images per staged half tile (s2s6q_kernel, form 1, taken when M % 128 == 0) - wino6q_kernel's
// transformation (read its header in wino6.hip) applied to the strided kind.  The phase profile of s2s6_kernel
// (profiles/experiments/r05_s2s6_phase_profile.log) shows its staging role (split results to LDS + the weight DMA wait: 3 100 cycles
// per stage) LONGER than its multiplying role (2 240), at 39 % matrix-pipe utilisation and 2.1 GHz - below the power cap, unlike the
// stride-1 kernel: the tile of 64 output channels makes every M block re-fetch, re-scale and re-split the same input (4 x at 256
// channels, 8 x at 512).  Here a block owns 128 output channels: each half tile T_g(s) is staged ONCE and multiplied by the weight
// images (s, m = 0) and (s, m = 1) in two multiplying phases with their own accumulator pairs - half the fetches, staging
// instructions and LDS writes per MFMA, same LDS budget (one 54 KB weight image, two 30.4 KB half tiles), same weight hand-over with
// the stage index replaced by the image index j = 2 s + m:
//     phase p (0 .. 4 nstage - 1): group p & 1 multiplies with image j = p >> 1; the other group is in its staging role
//     group 0:            M0(s)  SA  M1(s)  SB          group 1:   S  M0(s)  SA  M1(s)  SB   (one phase behind)
//     M0: 54 MFMAs + the fetch of stage s + 1       SA: weight DMA only
//     M1: 54 MFMAs + the staging arithmetic         SB: weight DMA + res -> T_g(s + 1)
// Same products in the same order per output element as s2s6_kernel: bit-identical results (tests/test_gpu_s2s6.py).
// (FOUR images per staged half tile - 256 output channels, 4 x 32 accumulator registers - does not fit: 256 VGPRs with 192 spilled.)
template <bool ISC>
__global__ __launch_bounds__(WT, 2) void s2s6q_kernel(const S2Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* ul = reinterpret_cast<u32x4*>(smem_raw);                                   // weights, 16-byte chunks
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wq = wid & 3, wm = wq >> 1, wrl = wq & 1, gt = tid & (GT - 1);
    unsigned* tl = reinterpret_cast<unsigned*>(smem_raw + U_SLOTS * 1024) + grp * TP_DWORDS;      // this group's half tile
    const u32x4* tl4 = reinterpret_cast<const u32x4*>(tl);
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int tq = jx / p.mblocks, mbq = jx % p.mblocks;               // (mblocks = M / 128 for this form)
    const int tile = p.nt8 ? (int)(((int64_t)xcd * p.ntiles) >> 3) + tq : tq * 8 + xcd;
    if (tile >= (p.nt8 ? (int)(((int64_t)(xcd + 1) * p.ntiles) >> 3) : p.ntiles)) return;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx * TWO, y0 = ty * TH, yh = y0 + PH * grp;
    const size_t iplane = (size_t)p.Hi * p.Wi, oplane = (size_t)p.H * p.W;
    const float* inb = p.in + (size_t)b * p.K * iplane;
    const float* iscb = ISC ? p.isc + (size_t)b * p.K : nullptr;

    f32x16 acc[2], accl[2];            // per weight image: the h x h products / the five small ones (s2s6_kernel)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[m][r] = 0.f; accl[m][r] = 0.f; }

    unsigned g_off[P_IN];
    int l_off[P_IN];
    bool last_col[P_IN], live[P_IN];
    unsigned q2[P_IN];
#pragma unroll
    for (int i = 0; i < P_IN; ++i) {          // staging geometry: as s2s6_kernel
        const int e = gt + GT * i;
        live[i] = e < N_ITEMS;
        const int ee = live[i] ? e : 0;
        const int cg = ee % NG, q = (ee / NG) & 7, row = ee / (NG * 8);
        last_col[i] = cg == NG - 1;
        g_off[i] = (unsigned)((2 * q * p.Hi + 2 * yh + row) * p.Wi + 2 * x0 + 4 * cg - (last_col[i] ? 3 : 0));
        l_off[i] = (((row * 2 + (q >> 2)) * 2) * CI + 2 * cg) * 4 + (q & 3);
        q2[i] = 2u * q;
    }
    const int MT = p.M >> 5;
    f32x4 rin[P_IN][2];
    f32x2 rsc[P_IN];
#pragma unroll
    for (int i = 0; i < P_IN; ++i) rsc[i] = f32x2{1.f, 1.f};
    const int nstage = p.K / KC, nimg = 2 * nstage;
    auto fetch_scales = [&](int s) {
        if (ISC) {
#pragma unroll
            for (int i = 0; i < P_IN; ++i) rsc[i] = *reinterpret_cast<const f32x2u*>(iscb + s * KC + q2[i]);
        }
    };
    auto fetch_item = [&](int i, int s) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
            rin[i][h2] = *reinterpret_cast<const f32x4u*>(inb + ((size_t)s * KC + h2) * iplane + g_off[i]);
    };
    // weight half `uh` of image j = 2 s + m: uh = 0: taps 0-4 (30 slots: 8 / 8 / 7 / 7 per wave), uh = 1: taps 5-8 (24 slots: 6 per wave)
    auto issue_u = [&](int uh, int j) {
        const u32x4* us = p.U + (size_t)(j >> 1) * 27 * MT * 64;
        const int mb = 2 * mbq + (j & 1);
        const int ntap = uh ? NTAP - UA_TAPS : UA_TAPS, tap0 = uh ? UA_TAPS : 0, n = ntap * 6;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int jw = wq + 4 * r;                                         // slot of this half: (piece, tap, M tile)
            if (jw >= n) break;
            const int piece = jw / (ntap * 2), rem = jw % (ntap * 2), tap = tap0 + (rem >> 1), mt = rem & 1;
            const int pt = piece * NTAP + tap;
            const u32x4* g = us + ((size_t)pt * MT + 2 * mb + mt) * 64 + (unsigned)lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(ul + (pt * 2 + mt) * 64), 16, 0, 0);
        }
    };
    // the staging arithmetic: s2s6_kernel's 51-slot program, run behind the MFMAs of the m = 1 phase
    unsigned res[P_IN][4][3];
    float te = 0.f, to = 0.f, fe = 0.f, fo = 0.f;
    auto arith = [&](int k) {
        if (k < 0) {
        } else if (k < P_IN) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 v = rin[k][h2];
                if (last_col[k]) v[0] = v[3];
                rin[k][h2] = ISC ? v * rsc[k][h2] : v;
                asm volatile("" : "+v"(rin[k][h2]));
            }
        } else if (k < N_SLOT) {
            const int u = (k - P_IN) >> 2, j = (k - P_IN) & 3, i = u >> 2, c = u & 3;
            if (j == 0) {
                te = rin[i][0][c]; to = rin[i][1][c];
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
        for (int i = 0; i < P_IN; ++i) {
            if (!live[i]) continue;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c > 0 && last_col[i]) continue;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) tl[l_off[i] + (c & 1) * CI * 4 + (c >> 1) * 4 + pc * TP_PLANE] = res[i][c][pc];
            }
        }
    };
    const int rr = l31 >> 4, jj = l31 & 15;
    const int b_chunk = ((2 * (2 * wrl + rr) * 2 + half) * 2) * CI + jj;
    const int a_chunk = wm * 64 + lane;

    // one multiplying phase with accumulator pair MSET; behind the MFMAs: MSET 0 the fetch of stage `fs`, MSET 1 the staging arithmetic
    auto multiply = [&](auto mset_tag, int fs) {
        constexpr int MSET = decltype(mset_tag)::value;
        bf16x8 av[2][3], bv[2][3];
        auto rd1 = [&](int t, int slot, int q) {
            const int ky = t / 3, kx = t % 3;
            if (q < 3) av[slot][q] = __builtin_bit_cast(bf16x8, ul[a_chunk + (q * NTAP + t) * 128]);
            else bv[slot][q - 3] = __builtin_bit_cast(bf16x8, tl4[b_chunk + ((ky * 2) * 2 + (kx & 1)) * CI + (kx >> 1) + (q - 3) * (TP_PLANE / 4)]);
        };
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
        for (int q = 0; q < 6; ++q) rd1(0, 0, q);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int slot = t & 1;
            if (t == UA_TAPS - 1) s2_barrier();        // mid-phase barrier
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                if (q < 5) accl[MSET] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][PA[q]], bv[slot][PB[q]], accl[MSET], 0, 0, 0);
                else acc[MSET] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][PA[q]], bv[slot][PB[q]], acc[MSET], 0, 0, 0);
                if (t + 1 < NTAP && q < 3) { rd1(t + 1, slot ^ 1, 2 * q); rd1(t + 1, slot ^ 1, 2 * q + 1); }
                const int k = t * 6 + q;
                if (MSET == 0) {
                    // the fetch of the next stage, an item every twelve slots from the sixth on (rin is free: the m = 1 phase consumed it)
                    if (k == 5) fetch_scales(fs);
#pragma unroll
                    for (int i = 0; i < P_IN; ++i)
                        if (k == 6 + 12 * i) fetch_item(i, fs);
                } else {
                    arith(k - SLOT0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        s2_barrier();                                  // end of phase
    };
    // one phase in the staging role (wino6q_kernel: stage): image cs = (ph + 1) >> 1 is the one whose half this phase renews - group 1
    // (even ph) half b in FRONT of the mid-phase barrier, group 0 (odd ph) half a BEHIND it; WRITE: the parked split results go to LDS
    auto stage = [&](int ph, bool write) {
        const int cs = (ph + 1) >> 1;
        const bool work = cs >= 1 && cs < nimg;
        if (grp == 1 && work) issue_u(1, cs);
        __builtin_amdgcn_sched_barrier(0);
        if (write) write_res();
        if (grp == 1 && work) s2_wait_vm();
        s2_barrier();                                  // mid-phase
        if (grp == 0 && work) {
            issue_u(0, cs);
            s2_wait_vm();
        }
        s2_barrier();                                  // end of phase
    };

    // prologue: every group splits and writes its half of stage 0; group 0 brings in the whole weight image 0
    fetch_scales(0);
#pragma unroll
    for (int i = 0; i < P_IN; ++i) fetch_item(i, 0);
    if (grp == 0) { issue_u(0, 0); issue_u(1, 0); }
#pragma unroll
    for (int k = 0; k < N_SLOT; ++k) arith(k);
    write_res();
    s2_wait_vm();
    s2_barrier();
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

    // epilogue: s2s6_kernel's, once per accumulator pair
    const float g_pos = p.act == 3 ? 1.4142135623730951f : 1.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int mbase = (2 * mbq + m) * BM + wm * 32;
        const size_t off0 = ((size_t)b * p.M + mbase) * oplane + (size_t)(yh + 2 * wrl + rr) * p.W + x0 + jj;
        float scv[16], biv[16], resv[16], mrefv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
            scv[r] = p.osc ? p.osc[(size_t)b * p.M + mbase + dm] : 1.f;
            biv[r] = p.bias ? p.bias[mbase + dm] : 0.f;
        }
        if (p.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) resv[r] = p.res[off0 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * oplane];
        }
        if (p.mref) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mrefv[r] = p.mref[off0 + (size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * oplane];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = (acc[m][r] + accl[m][r]) * scv[r] + biv[r];
            if (p.act >= 3) v = (v > 0.f ? v : v * 0.2f) * g_pos;
            if (p.res) v += resv[r];
            if (p.mref) v *= mrefv[r] > 0.f ? p.mgain : 0.2f * p.mgain;
            p.out[off0 + (size_t)dm * oplane] = v;
        }
    }
}

}  // namespace

#ifdef S2_PROF
extern "C" int te_debug_s2s6_prof(void* host_dst, int64_t bytes) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(te_s2s6_prof_buf), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int te_conv_s2s6_supported(int B, int K, int M, int H, int W) {
    if (!(B > 0 && K >= 32 && K % KC == 0 && M >= BM && M % BM == 0 && H >= TH && H % TH == 0 && W >= TWO && W % TWO == 0)) return 0;
    return ((int64_t)K * (2 * H + 1) * (2 * W + 1) * 4 < 0x7FFFFFFF && (int64_t)B * (H / TH) * (W / TWO) * (M / BM) < 0x7FFFFFF0) ? 1 : 0;
}

// kernel form of TE_CONV_S2S6: 0 = ping-pong (s2s6_kernel, round 5); 1 (round 6, default) = the two-image form s2s6q_kernel where
// M % 128 == 0 and the grid still gives every CU a block (it has half as many blocks), the ping-pong form elsewhere; 2 = the two-image
// form wherever M % 128 == 0 (tests).  Same results bit for bit.  A process-wide A/B switch like te_conv_wino6_form (te_hip.h);
// TE_S2S6_FORM in the environment sets the initial value.
static std::atomic<int> g_s2_form{[] { const char* e = getenv("TE_S2S6_FORM"); return e ? atoi(e) : 1; }()};
extern "C" int te_conv_s2s6_form(int form) {
    const int old = g_s2_form.load(std::memory_order_relaxed);
    if (form >= 0 && form <= 2) g_s2_form.store(form, std::memory_order_relaxed);
    return old;
}

int te_s2s6_launch(float* out, const float* in, const float* U, const float* isc, const float* osc, const float* bias, const float* res,
                   const float* mask_ref, float mask_gain, int act, int B, int K, int M, int H, int W, hipStream_t s) {
    TE_REQUIRE(te_conv_s2s6_supported(B, K, M, H, W), TE_ERR_UNSUPPORTED,
               "te_conv_f32(TE_CONV_S2S6): needs K %% 16 == 0 (>= 32), M %% 64 == 0, H %% 8 == 0, W %% 16 == 0 (te_conv_s2s6_supported)");
    TE_REQUIRE((reinterpret_cast<uintptr_t>(U) & 15) == 0 && (reinterpret_cast<uintptr_t>(in) & 3) == 0, TE_ERR_UNSUPPORTED,
               "te_conv_f32(TE_CONV_S2S6): 16-byte aligned packed weights required");
    S2Args a{};
    a.out = out; a.in = in; a.U = reinterpret_cast<const u32x4*>(U); a.isc = isc; a.osc = osc; a.bias = bias; a.res = res;
    a.mref = mask_ref; a.mgain = mask_gain; a.act = act;
    a.B = B; a.K = K; a.M = M; a.H = H; a.W = W; a.Hi = 2 * H + 1; a.Wi = 2 * W + 1;
    a.tiles_x = W / TWO; a.tiles_y = H / TH; a.mblocks = M / BM;
    a.ntiles = B * a.tiles_x * a.tiles_y;
    a.nt8 = te::xcd_banded() ? (int)te::cdiv(a.ntiles, 8) : 0;
    const int64_t blocks = te::cdiv(a.ntiles, 8) * 8 * a.mblocks;
    const size_t lds = (size_t)U_SLOTS * 1024 + 2 * (size_t)TP_DWORDS * 4;
    const int form = g_s2_form.load(std::memory_order_relaxed);
    const int64_t blocks_q = te::cdiv(a.ntiles, 8) * 8 * (M / (2 * BM));
    static std::atomic<uint64_t> attr_done{0}, attr_done_sc{0}, attr_done_q{0}, attr_done_qs{0};
    if (form >= 1 && M % (2 * BM) == 0 && (form == 2 || blocks_q >= te::kNumCU)) {
        a.mblocks = M / (2 * BM);
        if (isc) {
            te::allow_big_lds(attr_done_qs, (const void*)s2s6q_kernel<true>, 160 * 1024);
            s2s6q_kernel<true><<<dim3((unsigned)blocks_q), WT, lds, s>>>(a);
        } else {
            te::allow_big_lds(attr_done_q, (const void*)s2s6q_kernel<false>, 160 * 1024);
            s2s6q_kernel<false><<<dim3((unsigned)blocks_q), WT, lds, s>>>(a);
        }
    } else if (isc) {
        te::allow_big_lds(attr_done_sc, (const void*)s2s6_kernel<true>, 160 * 1024);
        s2s6_kernel<true><<<dim3((unsigned)blocks), WT, lds, s>>>(a);
    } else {
        te::allow_big_lds(attr_done, (const void*)s2s6_kernel<false>, 160 * 1024);
        s2s6_kernel<false><<<dim3((unsigned)blocks), WT, lds, s>>>(a);
    }
    return te::launch_status("te_conv_f32(TE_CONV_S2S6)");
}
