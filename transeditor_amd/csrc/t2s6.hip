// T2 on the bf16 matrix pipe (round 5, kind TE_CONV_T2S6): the transposed 3x3 / stride 2 convolution  in [B,K,H,W] -> out [B,M,2H+1,2W+1]
// - the generator's up-sampling layers (conv_transpose2d(stride 2, pad 0) of ModulatedConv2d.forward, model_spatial_query.py:310-321)
// and the data gradient of the discriminator's down-sampling convolutions (:765-779) - with every fp32 operand split into three bf16
// pieces and six exact piece products per multiply-add accumulated in fp32 (wino6.hip / s2s6.hip: fp32-equivalent results).
//
// Polyphase form, as conv_mfma_kernel<TE_CONV_T2>: output pixel (2 i + a, 2 j + b) of cell (i, j) receives the taps with ky = a, kx = b
// (mod 2) - 4 + 2 + 2 + 1 = 9 tap products per input pixel, no zero insertion - into FOUR accumulator tiles (one per phase); a tap with
// ky = 2 (kx = 2) reads the input one row up (one column left).  This kernel covers the BODY cells [0, H) x [0, W), i.e. output rows
// 0 .. 2H - 1 and columns 0 .. 2W - 1; the last output row and column (cells i = H / j = W, which only see taps with ky = 2 / kx = 2)
// come from t2_edge_kernel below (round 6; until then two thin regions of the fp32 kernel), launched behind it from the fp32 copy of
// the weights that the packed buffer carries along (TE_PACK_T6FWD / TE_PACK_T6SWAP = split fragment-order layout of s2s6.hip +
// TE_PACK_FWD / TE_PACK_SWAP layout; conv.hip).
//
// Structure = s2s6.hip (ping-pong form, read wino6.hip's header): block 512 threads, cell tile 64 output channels x 8 rows x 16 columns,
// two half tiles of 4 cell rows (5 input rows x 17 columns each with the halo), wave (wm, wrl) of a group = 32 channels x cell rows
// {2 wrl, 2 wrl + 1} x 16 columns x 4 phases (64 accumulator registers); 54 MFMAs per wave and stage of 16 input channels; the staging
// arithmetic of a half (200 items: one per thread) rides behind the multiplying wave's own MFMAs; weights 54 KB per stage by LDS-DMA
// in two parts (taps 0-4 / 5-8).  Input tile in LDS: T[piece][row 5][k half][24 columns (17 used)][8 bf16] - two rows are 48 chunks,
// a multiple of 16, so the two cell rows of a B fragment never collide (conflict-free ds_read_b128).
//
// Round 6: the weight image is DOUBLE-BUFFERED (2 x 54 KB + 2 x 11.25 KB of half tiles = 130.5 KB).  The round-5 kernel renewed the one image
// half by half inside the phases that still read it (wino6.hip's protocol): every LDS-DMA had half a phase (~1 200 cycles) for ~1 500 - 1 750
// cycles of issue + L2 latency, and the phase profile showed the difference on the critical path of BOTH phases (the multiplying group waits
// ~770 cycles at the mid-phase barrier of X, the staging group's DMA outlasts the partner's MFMAs by ~340 in Y: ~15 % of a stage,
// profiles/experiments/r06_st_phase_profile_before.log).  Now stage s + 1 goes to the buffer stage s - 1 was read from: part 0 issued by
// group 1 at the start of phase 2 s, part 1 by group 0 at the start of phase 2 s + 1, each waited for at the end of the issuing group's
// staging phase - a whole phase per transfer, and ONE barrier per phase (the mid-phase barrier is gone).  Measured (same tool,
// profiles/experiments/r06_st_phase_profile_t2s6_double_buffered.log): 6 553 -> 5 350 cycles per stage at 512 -> 512 @32^2, 6 962 -> 5 633 at
// 128 -> 128 @128^2 under the phase profiler - and the un-profiled launch times did not move (158 - 166 TFLOP/s before and after,
// r06_t2s6_check_double_buffered.log): the kernel sits at the part's power limit, the saved cycles came back as a lower clock.  Kept for
// the simpler protocol.  (s2s6.hip keeps the half-by-half protocol: two of its 54 KB images do not fit beside its 61 KB of half tiles.)
#include "conv_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int WT = 512, GT = 256, KC = 16, BM = 64;
constexpr int TH = 8, TWC = 16, PH = 4;                       // cell tile 8 x 16, half tile 4 cell rows
constexpr int IR = PH + 1, IC = TWC + 1;                      // input rows / columns of a half tile (with the halo): 5 x 17
constexpr int CW = 24;                                        // chunks per (row, k half): 17 used
constexpr int NTAP = 9;
constexpr int U_SLOTS = 3 * NTAP * 2;                         // 54 fragment slots of 1 KB
constexpr int TP_PLANE = IR * 2 * CW * 4;                     // dwords of one piece of a half tile: 960
constexpr int TP_DWORDS = 3 * TP_PLANE;                       // 2 880 dwords = 11.25 KB
constexpr int NG = (IC + 3) / 4;                              // 4-column groups per input row: 5 (the last one holds column 16 only)
constexpr int N_ITEMS = IR * NG * 8;                          // (row, column group, channel pair) items of a half: 200 (one per thread)
constexpr int N_SLOT = 1 + 16;                                // arithmetic slots: scale + 4 positions x 4 steps
constexpr int SLOT0 = 24;                                     // MFMA behind which the program starts (54 per phase)
constexpr int UA_TAPS = 5;                                    // taps 0-4: weight half a (30 slots), taps 5-8: half b (24 slots)

#ifdef T2_PROF       // experimental builds: per-wave cycle counts of the phases, read back with te_debug_t2s6_prof (tools/s2s6_phase_prof.py)
__device__ unsigned long long te_t2s6_prof_buf[2048 * 8 * 8];
#define T2_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define T2_ACC(i, a, b) pc[i] += (b) - (a)
#else
#define T2_T(v)
#define T2_ACC(i, a, b)
#endif

#ifndef DMA_PRIO
#define DMA_PRIO 0         // experiment: the staging wave raises its priority while it issues the weight DMA (wino6.hip)
#endif

struct T2Args {
    float* out; const float* in; const u32x4* U; const float* isc; const float* osc; const float* bias; int act;
    float* colbuf;           // optional [B][K][H]: the (style-scaled) last input column, written while it is staged (t2_edge_kernel reads it)
    int B, K, M, H, W, Ho, Wo, ntiles, mblocks, tiles_x, tiles_y, nt8;
};

__device__ __forceinline__ void t2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void t2_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ISC: the launch carries style scales (template parameter; see s2s6.hip)
template <bool ISC>
__global__ __launch_bounds__(WT, 2) void t2s6_kernel(const T2Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* ul = reinterpret_cast<u32x4*>(smem_raw);                                   // weights, 16-byte chunks
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform by construction: keep the role branches scalar
    const int grp = wid >> 2, wq = wid & 3, wm = wq >> 1, wrl = wq & 1, gt = tid & (GT - 1);
    unsigned* tl = reinterpret_cast<unsigned*>(smem_raw + 2 * U_SLOTS * 1024) + grp * TP_DWORDS;  // this group's half tile (behind the two weight images)
    const u32x4* tl4 = reinterpret_cast<const u32x4*>(tl);
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int tq = jx / p.mblocks, mb = jx % p.mblocks;
    const int tile = p.nt8 ? (int)(((int64_t)xcd * p.ntiles) >> 3) + tq : tq * 8 + xcd;
    if (tile >= (p.nt8 ? (int)(((int64_t)(xcd + 1) * p.ntiles) >> 3) : p.ntiles)) return;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx * TWC, y0 = ty * TH, yh = y0 + PH * grp;
    const size_t iplane = (size_t)p.H * p.W, oplane = (size_t)p.Ho * p.Wo;
    const float* inb = p.in + (size_t)b * p.K * iplane;
    const float* iscb = ISC ? p.isc + (size_t)b * p.K : nullptr;
    const bool store_col = p.colbuf != nullptr && tx == p.tiles_x - 1 && mb == 0;      // block-uniform

    f32x16 acc[4];                                     // one tile per output phase (a, b) = (ky & 1, kx & 1)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    // staging geometry of this group's half: ONE item per thread, e = gt < 200 -> (pair-in-chunk e % 4, column group cg = (e / 4) % 5, k half (e / 20) % 2,
    // row = e / 40); the item covers input columns x0 - 1 + 4 cg .. + 3 of input row yh - 1 + row, channels 2 q and 2 q + 1 of the stage.
    // Branch-free edges (wino6.hip): the four floats are always loaded from inside the row - one column later at the left image
    // border (column -1 does not exist), three columns earlier in the last group (only its first column, x0 + 15, belongs to the
    // tile) - and the vector is patched when the item is scaled; the row above the image (yh = 0) is loaded from row 0 and zeroed.
    const bool live = gt < N_ITEMS;
    const int ee = live ? gt : 0;
    // (pair-in-chunk fastest over the lanes: four lanes fill one 16-byte chunk - at most two lanes of a write group share a bank)
    const int dq = ee & 3, rest = ee >> 2, cg = rest % NG, q = ((rest / NG) & 1) * 4 + dq, row = rest / (NG * 2);
    const bool left = x0 == 0 && cg == 0, last_col = cg == NG - 1, rowout = yh - 1 + row < 0;
    const int gy = rowout ? 0 : yh - 1 + row;
    const unsigned g_off = (unsigned)((2 * q * p.H + gy) * p.W + x0 - 1 + 4 * cg + (left ? 1 : 0) - (last_col ? 3 : 0));
    const int l_off = ((row * 2 + (q >> 2)) * CW + 4 * cg) * 4 + (q & 3);          // + position * 4 + piece * TP_PLANE
    const unsigned q2 = 2u * q;
    const int MT = p.M >> 5;
    f32x4 rin[2];
    f32x2 rsc = {1.f, 1.f};
    const int nstage = p.K / KC;
    // fetch of stage s.  Inside the loop it is issued by the MULTIPLYING role, right behind the last slot of the arithmetic that reads
    // the registers, so that the fetch registers are written and read in one role only (wino6.hip, fetch_item)
    auto fetch_scales = [&](int s) {
        if (ISC) rsc = *reinterpret_cast<const f32x2u*>(iscb + s * KC + q2);
    };
    auto fetch_item = [&](int s) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) rin[h2] = *reinterpret_cast<const f32x4u*>(inb + ((size_t)s * KC + h2) * iplane + g_off);
    };
    auto issue = [&](int s) { fetch_scales(s); fetch_item(s); };
    // weight part `uh` of stage s into buffer s & 1: uh = 0: taps 0-4 (30 slots: 8 / 8 / 7 / 7 per wave), uh = 1: taps 5-8 (24 slots: 6 per wave)
    auto issue_u = [&](int uh, int s) {
        const u32x4* us = p.U + (size_t)s * 27 * MT * 64;
        u32x4* ub = ul + (s & 1) * (U_SLOTS * 64);
        const int ntap = uh ? NTAP - UA_TAPS : UA_TAPS, tap0 = uh ? UA_TAPS : 0, n = ntap * 6;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int j = wq + 4 * r;                                          // slot of this half: (piece, tap, M tile)
            if (j >= n) break;
            const int piece = j / (ntap * 2), rem = j % (ntap * 2), tap = tap0 + (rem >> 1), mt = rem & 1;
            const int pt = piece * NTAP + tap;
            const u32x4* g = us + ((size_t)pt * MT + 2 * mb + mt) * 64 + (unsigned)lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(ub + (pt * 2 + mt) * 64), 16, 0, 0);
        }
    };
    // ---- the staging arithmetic as a program of 17 slots behind the MFMAs of the multiplying phase (wino6.hip):
    //   slot 0: edge patch + style scale;  slots 1 + 4 c + j: position c of the item, the four steps of the three-piece split
    unsigned res[4][3];
    float te = 0.f, to = 0.f, fe = 0.f, fo = 0.f;
    // (sa = the stage whose registers the program works on)
    auto arith = [&](int k, int sa) {
        if (k < 0) {
        } else if (k == 0) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 v = rin[h2];
                if (left) { v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = 0.f; }        // loaded from column 0: position 0 is column -1
                if (last_col) v[0] = v[3];                                               // loaded three columns early
                if (rowout) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                rin[h2] = ISC ? v * rsc[h2] : v;
                asm volatile("" : "+v"(rin[h2]));
            }
            // Round 6: the blocks of the last tile column (first M block only) leave the scaled last input column W - 1 in a compact
            // buffer [B][K][H] for t2_edge_kernel - which otherwise gathers it with one cache line per element (B K H lines: 524 k at the
            // two large FFHQ-256 shapes, 90 - 150 us of miss latency on the few CUs that run the column blocks).  Rows shared by two half
            // tiles (the halo) are written twice with the same value.
            if (store_col) {
                if (last_col && live && !rowout) {
                    float* cb = p.colbuf + ((size_t)b * p.K + (size_t)sa * KC + q2) * p.H + (yh - 1 + row);
                    cb[0] = rin[0][0];
                    cb[p.H] = rin[1][0];
                }
            }
        } else if (k < N_SLOT) {
            const int c = (k - 1) >> 2, j = (k - 1) & 3;
            if (j == 0) {
                te = rin[0][c]; to = rin[1][c];                              // even / odd channel of the pair at position c
                const f32x2 t = {te, to};
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                res[c][0] = h;
                fe = __builtin_bit_cast(float, h << 16);
                fo = __builtin_bit_cast(float, h & 0xFFFF0000u);
            } else if (j == 1) {
                te -= fe; to -= fo;
            } else if (j == 2) {
                const f32x2 t = {te, to};
                const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                res[c][1] = m;
                fe = __builtin_bit_cast(float, m << 16);
                fo = __builtin_bit_cast(float, m & 0xFFFF0000u);
            } else {
                te -= fe; to -= fo;
                const f32x2 t = {te, to};
                res[c][2] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                asm volatile("" : "+v"(res[c][2]));
            }
            asm volatile("" : "+v"(te), "+v"(to), "+v"(fe), "+v"(fo));        // (pin the step here: see wino6.hip)
        }
    };
    auto write_res = [&]() {
        if (!live) return;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c > 0 && last_col) continue;                             // (columns 17..19 do not exist in the tile)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) tl[l_off + c * 4 + pc * TP_PLANE] = res[c][pc];
        }
    };
    const int rr = l31 >> 4, jj = l31 & 15;
    // B operand of tap (ky, kx), piece pc: chunk ((2 wrl + rr + 1 - (ky == 2)) * 2 + half) * CW + jj + 1 - (kx == 2) + pc * TP_PLANE / 4
    const int b_chunk = ((2 * wrl + rr + 1) * 2 + half) * CW + jj + 1;
    const int a_chunk = wm * 64 + lane;                                 // + (piece * 9 + tap) * 128

    // prologue: every group splits and writes its half of stage 0 and fetches stage 1; group 0 brings in the whole weight image
    issue(0);
    if (grp == 0) { issue_u(0, 0); issue_u(1, 0); }
#pragma unroll
    for (int k = 0; k < N_SLOT; ++k) arith(k, 0);
    write_res();
    issue(1);
    if (grp == 0) {          // (counted: only the weight DMA must have landed; the fetch of stage 1 - 3 loads, 2 without style scales - stays in flight)
        if (ISC) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    t2_barrier();
    const int nphase = 2 * nstage;
#ifdef T2_PROF
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long pstart = __builtin_readcyclecounter(), rstart = __builtin_amdgcn_s_memrealtime();
#endif
    for (int ph = 0; ph < nphase; ++ph) {
        const bool last = ph == nphase - 1;
        T2_T(t0);
        if ((ph & 1) == grp) {
            // ---- multiply this group's half of stage ph / 2; behind the MFMAs: the arithmetic of the stage after (rin -> res)
            bf16x8 av[2][3], bv[2][3];
            const int fs2 = min((ph >> 1) + 2, nstage - 1), sa = min((ph >> 1) + 1, nstage - 1);
            const u32x4* ua = ul + ((ph >> 1) & 1) * (U_SLOTS * 64) + a_chunk;          // this stage's weight image
            auto rd1 = [&](int t, int slot, int qq) {
                const int ky = t / 3, kx = t % 3;
                if (qq < 3) av[slot][qq] = __builtin_bit_cast(bf16x8, ua[(qq * NTAP + t) * 128]);
                else bv[slot][qq - 3] = __builtin_bit_cast(bf16x8, tl4[b_chunk - (ky == 2 ? 2 * CW : 0) - (kx == 2 ? 1 : 0) + (qq - 3) * (TP_PLANE / 4)]);
            };
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
            for (int qq = 0; qq < 6; ++qq) rd1(0, 0, qq);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int slot = t & 1, c = ((t / 3) & 1) * 2 + ((t % 3) & 1);          // output phase of the tap
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int qq = 0; qq < 6; ++qq) {
#ifndef ST_SKIP_MFMA     // (experiment switches ST_*: timing decomposition only, results are wrong)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][PA[qq]], bv[slot][PB[qq]], acc[c], 0, 0, 0);
#endif
                    if (t + 1 < NTAP && qq < 3) { rd1(t + 1, slot ^ 1, 2 * qq); rd1(t + 1, slot ^ 1, 2 * qq + 1); }
#ifndef ST_NO_ARITH
                    arith(t * 6 + qq - SLOT0, sa);
#endif
#ifndef ST_NO_FETCH
                    if (t * 6 + qq - SLOT0 == 0) fetch_scales(fs2);          // the fetch of the stage after next (see fetch_item)
                    if (t * 6 + qq - SLOT0 == 13) fetch_item(fs2);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
#ifdef T2_PROF
            { asm volatile("s_nop 0" ::: "memory"); T2_T(tc); T2_ACC(0, t0, tc); }
#endif
        } else {
            // ---- stage: move this group's half of stage (ph + 1) / 2 to LDS; bring in this group's part (ph & 1: group 1 part 0 in even
            // phases, group 0 part 1 in odd ones) of the weights of stage (ph >> 1) + 1, into the buffer stage (ph >> 1) - 1 was read
            // from (last read in phase ph - 1 / ph - 2); waited for at the end of THIS phase.  The LDS writes are unconditional (s2s6.hip).
            const int cw = (ph >> 1) + 1;
#ifndef ST_NO_DMA
            if (cw < nstage) issue_u(ph & 1, cw);
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifndef ST_NO_DSW
            write_res();
#endif
            T2_T(ta);
            t2_wait_vm();
            T2_T(tb);
            T2_ACC(3, t0, ta); T2_ACC(4, ta, tb);
        }
        T2_T(t8);
        if (!last) t2_barrier();
        T2_T(t9);
        T2_ACC(6, t8, t9);
    }
#ifdef T2_PROF
    if (lane == 0 && blockIdx.x < 2048) {
        unsigned long long* d = te_t2s6_prof_buf + ((size_t)blockIdx.x * 8 + wid) * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = pc[i];
        d[6] = pc[6] | ((__builtin_amdgcn_s_memrealtime() - rstart) << 40);
        d[7] = ((unsigned long long)nstage << 48) | ((__builtin_readcyclecounter() - pstart) & 0xFFFFFFFFFFFFull);
    }
#endif
    // epilogue: demodulation scale, bias, leaky ReLU; phases (a, 0) and (a, 1) of a cell are adjacent output columns: one 8-byte store
    const int mbase = mb * BM + wm * 32;
    const int ci = yh + 2 * wrl + rr, cj = x0 + jj;
    const size_t off0 = ((size_t)b * p.M + mbase) * oplane + (size_t)(2 * ci) * p.Wo + 2 * cj;
    const float g_pos = p.act == 3 ? 1.4142135623730951f : 1.f;
    float scv[16], biv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
        scv[r] = p.osc ? p.osc[(size_t)b * p.M + mbase + dm] : 1.f;
        biv[r] = p.bias ? p.bias[mbase + dm] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float v0 = acc[a * 2][r] * scv[r] + biv[r], v1 = acc[a * 2 + 1][r] * scv[r] + biv[r];
            if (p.act >= 3) {
                v0 = (v0 > 0.f ? v0 : v0 * 0.2f) * g_pos;
                v1 = (v1 > 0.f ? v1 : v1 * 0.2f) * g_pos;
            }
            f32x2 v; v[0] = v0; v[1] = v1;
            *reinterpret_cast<f32x2u*>(p.out + off0 + (size_t)dm * oplane + (size_t)a * p.Wo) = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: TWO 64-channel weight images per staged half tile (t2s6q_kernel, taken when M % 128 == 0) - wino6q_kernel's / s2s6q_kernel's
// transformation on this kernel's double-buffered weight protocol.  A block owns 128 output channels: each half tile T_g(s) is staged
// once and multiplied by the weight images (s, m = 0) and (s, m = 1) in two multiplying phases with their own accumulators (2 x 64
// registers); image j = 2 s + m lives in buffer j & 1 and is renewed exactly as a stage was: part 0 of image j + 1 by group 1 at the
// start of phase 2 j, part 1 by group 0 at the start of phase 2 j + 1, each waited for at the end of the issuing group's staging phase.
//     group 0:   M0(s)  S  M1(s)  SW          group 1:   S  M0(s)  S  M1(s)  SW     (one phase behind)
//     M0: 54 MFMAs + the fetch of stage s + 1     M1: 54 MFMAs + the staging arithmetic     S: weight DMA     SW: weight DMA + res -> T_g(s + 1)
// Same products in the same order per output element as t2s6_kernel: bit-identical results (tests/test_gpu_t2s6.py).
template <bool ISC>
__global__ __launch_bounds__(WT, 2) void t2s6q_kernel(const T2Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* ul = reinterpret_cast<u32x4*>(smem_raw);                                   // weights, 16-byte chunks
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, wq = wid & 3, wm = wq >> 1, wrl = wq & 1, gt = tid & (GT - 1);
    unsigned* tl = reinterpret_cast<unsigned*>(smem_raw + 2 * U_SLOTS * 1024) + grp * TP_DWORDS;
    const u32x4* tl4 = reinterpret_cast<const u32x4*>(tl);
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int tq = jx / p.mblocks, mbq = jx % p.mblocks;               // (mblocks = M / 128 for this form)
    const int tile = p.nt8 ? (int)(((int64_t)xcd * p.ntiles) >> 3) + tq : tq * 8 + xcd;
    if (tile >= (p.nt8 ? (int)(((int64_t)(xcd + 1) * p.ntiles) >> 3) : p.ntiles)) return;
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, b = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx * TWC, y0 = ty * TH, yh = y0 + PH * grp;
    const size_t iplane = (size_t)p.H * p.W, oplane = (size_t)p.Ho * p.Wo;
    const float* inb = p.in + (size_t)b * p.K * iplane;
    const float* iscb = ISC ? p.isc + (size_t)b * p.K : nullptr;
    const bool store_col = p.colbuf != nullptr && tx == p.tiles_x - 1 && mbq == 0;      // block-uniform (t2s6_kernel)

    f32x16 acc[2][4];                                  // per weight image: one tile per output phase
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][c][r] = 0.f;

    // staging geometry: as t2s6_kernel (one item per thread)
    const bool live = gt < N_ITEMS;
    const int ee = live ? gt : 0;
    const int dq = ee & 3, rest = ee >> 2, cg = rest % NG, q = ((rest / NG) & 1) * 4 + dq, row = rest / (NG * 2);
    const bool left = x0 == 0 && cg == 0, last_col = cg == NG - 1, rowout = yh - 1 + row < 0;
    const int gy = rowout ? 0 : yh - 1 + row;
    const unsigned g_off = (unsigned)((2 * q * p.H + gy) * p.W + x0 - 1 + 4 * cg + (left ? 1 : 0) - (last_col ? 3 : 0));
    const int l_off = ((row * 2 + (q >> 2)) * CW + 4 * cg) * 4 + (q & 3);
    const unsigned q2 = 2u * q;
    const int MT = p.M >> 5;
    f32x4 rin[2];
    f32x2 rsc = {1.f, 1.f};
    const int nstage = p.K / KC, nimg = 2 * nstage;
    auto fetch_scales = [&](int s) {
        if (ISC) rsc = *reinterpret_cast<const f32x2u*>(iscb + s * KC + q2);
    };
    auto fetch_item = [&](int s) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) rin[h2] = *reinterpret_cast<const f32x4u*>(inb + ((size_t)s * KC + h2) * iplane + g_off);
    };
    // weight part `uh` of image j = 2 s + m into buffer j & 1: uh = 0: taps 0-4 (30 slots), uh = 1: taps 5-8 (24 slots)
    auto issue_u = [&](int uh, int j) {
        const u32x4* us = p.U + (size_t)(j >> 1) * 27 * MT * 64;
        const int mb = 2 * mbq + (j & 1);
        u32x4* ub = ul + (j & 1) * (U_SLOTS * 64);
        const int ntap = uh ? NTAP - UA_TAPS : UA_TAPS, tap0 = uh ? UA_TAPS : 0, n = ntap * 6;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int jw = wq + 4 * r;
            if (jw >= n) break;
            const int piece = jw / (ntap * 2), rem = jw % (ntap * 2), tap = tap0 + (rem >> 1), mt = rem & 1;
            const int pt = piece * NTAP + tap;
            const u32x4* g = us + ((size_t)pt * MT + 2 * mb + mt) * 64 + (unsigned)lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(ub + (pt * 2 + mt) * 64), 16, 0, 0);
        }
    };
    unsigned res[4][3];
    float te = 0.f, to = 0.f, fe = 0.f, fo = 0.f;
    auto arith = [&](int k, int sa) {           // t2s6_kernel's 17-slot program
        if (k < 0) {
        } else if (k == 0) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                f32x4 v = rin[h2];
                if (left) { v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = 0.f; }
                if (last_col) v[0] = v[3];
                if (rowout) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
                rin[h2] = ISC ? v * rsc[h2] : v;
                asm volatile("" : "+v"(rin[h2]));
            }
            if (store_col) {
                if (last_col && live && !rowout) {
                    float* cb = p.colbuf + ((size_t)b * p.K + (size_t)sa * KC + q2) * p.H + (yh - 1 + row);
                    cb[0] = rin[0][0];
                    cb[p.H] = rin[1][0];
                }
            }
        } else if (k < N_SLOT) {
            const int c = (k - 1) >> 2, j = (k - 1) & 3;
            if (j == 0) {
                te = rin[0][c]; to = rin[1][c];
                const f32x2 t = {te, to};
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                res[c][0] = h;
                fe = __builtin_bit_cast(float, h << 16);
                fo = __builtin_bit_cast(float, h & 0xFFFF0000u);
            } else if (j == 1) {
                te -= fe; to -= fo;
            } else if (j == 2) {
                const f32x2 t = {te, to};
                const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                res[c][1] = m;
                fe = __builtin_bit_cast(float, m << 16);
                fo = __builtin_bit_cast(float, m & 0xFFFF0000u);
            } else {
                te -= fe; to -= fo;
                const f32x2 t = {te, to};
                res[c][2] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                asm volatile("" : "+v"(res[c][2]));
            }
            asm volatile("" : "+v"(te), "+v"(to), "+v"(fe), "+v"(fo));
        }
    };
    auto write_res = [&]() {
        if (!live) return;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c > 0 && last_col) continue;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) tl[l_off + c * 4 + pc * TP_PLANE] = res[c][pc];
        }
    };
    const int rr = l31 >> 4, jj = l31 & 15;
    const int b_chunk = ((2 * wrl + rr + 1) * 2 + half) * CW + jj + 1;
    const int a_chunk = wm * 64 + lane;

    // one multiplying phase (global phase ph: image ph >> 1) with accumulator set MSET; behind the MFMAs: MSET 0 the fetch of stage
    // `fs`, MSET 1 the staging arithmetic on it
    auto multiply = [&](auto mset_tag, int ph, int fs) {
        constexpr int MSET = decltype(mset_tag)::value;
        bf16x8 av[2][3], bv[2][3];
        const u32x4* ua = ul + ((ph >> 1) & 1) * (U_SLOTS * 64) + a_chunk;
        auto rd1 = [&](int t, int slot, int qq) {
            const int ky = t / 3, kx = t % 3;
            if (qq < 3) av[slot][qq] = __builtin_bit_cast(bf16x8, ua[(qq * NTAP + t) * 128]);
            else bv[slot][qq - 3] = __builtin_bit_cast(bf16x8, tl4[b_chunk - (ky == 2 ? 2 * CW : 0) - (kx == 2 ? 1 : 0) + (qq - 3) * (TP_PLANE / 4)]);
        };
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};        // small terms first: mm, hl, lh, hm, mh, hh
#pragma unroll
        for (int qq = 0; qq < 6; ++qq) rd1(0, 0, qq);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int slot = t & 1, c = ((t / 3) & 1) * 2 + ((t % 3) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qq = 0; qq < 6; ++qq) {
                acc[MSET][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[slot][PA[qq]], bv[slot][PB[qq]], acc[MSET][c], 0, 0, 0);
                if (t + 1 < NTAP && qq < 3) { rd1(t + 1, slot ^ 1, 2 * qq); rd1(t + 1, slot ^ 1, 2 * qq + 1); }
                const int k = t * 6 + qq;
                if (MSET == 0) {
                    if (k == 11) fetch_scales(fs);            // (rin is free: the m = 1 phase consumed it)
                    if (k == 12) fetch_item(fs);
                } else {
                    arith(k - SLOT0, fs);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        t2_barrier();                                  // end of phase
    };
    // one phase in the staging role: this group's part (ph & 1) of image (ph >> 1) + 1 into the buffer image (ph >> 1) - 1 was read from,
    // waited for at the end of the phase; WRITE: the parked split results go to the half tile
    auto stage = [&](int ph, bool write) {
        const int cw = (ph >> 1) + 1;
        if (cw < nimg) issue_u(ph & 1, cw);
        __builtin_amdgcn_sched_barrier(0);
        if (write) write_res();
        t2_wait_vm();
        t2_barrier();                                  // end of phase
    };

    // prologue: every group splits and writes its half of stage 0; group 0 brings in the whole weight image 0
    fetch_scales(0);
    fetch_item(0);
    if (grp == 0) { issue_u(0, 0); issue_u(1, 0); }
#pragma unroll
    for (int k = 0; k < N_SLOT; ++k) arith(k, 0);
    write_res();
    t2_wait_vm();
    t2_barrier();
    int ph = 0;
    if (grp == 1) { stage(0, false); ph = 1; }
    for (int s = 0; s < nstage; ++s) {
        const int fs = min(s + 1, nstage - 1);
        multiply(std::integral_constant<int, 0>{}, ph, fs);
        stage(ph + 1, false);
        multiply(std::integral_constant<int, 1>{}, ph + 2, fs);
        if (!(grp == 1 && s == nstage - 1)) stage(ph + 3, true);
        ph += 4;
    }

    // epilogue: t2s6_kernel's, once per accumulator set
    const int ci = yh + 2 * wrl + rr, cj = x0 + jj;
    const float g_pos = p.act == 3 ? 1.4142135623730951f : 1.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int mbase = (2 * mbq + m) * BM + wm * 32;
        const size_t off0 = ((size_t)b * p.M + mbase) * oplane + (size_t)(2 * ci) * p.Wo + 2 * cj;
        float scv[16], biv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
            scv[r] = p.osc ? p.osc[(size_t)b * p.M + mbase + dm] : 1.f;
            biv[r] = p.bias ? p.bias[mbase + dm] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float v0 = acc[m][a * 2][r] * scv[r] + biv[r], v1 = acc[m][a * 2 + 1][r] * scv[r] + biv[r];
                if (p.act >= 3) {
                    v0 = (v0 > 0.f ? v0 : v0 * 0.2f) * g_pos;
                    v1 = (v1 > 0.f ? v1 : v1 * 0.2f) * g_pos;
                }
                f32x2 v; v[0] = v0; v[1] = v1;
                *reinterpret_cast<f32x2u*>(p.out + off0 + (size_t)dm * oplane + (size_t)a * p.Wo) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: the LAST OUTPUT ROW AND COLUMN of the transposed kind (cells i = H / j = W) as one small vector-ALU launch.
//
// Until now they were two thin regions of conv_mfma_kernel<TE_CONV_T2>: a few dozen blocks of 128-cell tiles with one valid cell row or
// column each, every block walking the whole channel loop - 130 - 150 us for 0.8 GFLOP, ~2.5 ms of the 111 ms FFHQ-256 iteration (19
// launches), more than a quarter of a 128-channel launch.  The work is two "line" problems that only see one input row / column:
//     out[m][2H][2j + 0] = sum_k s_k ( w[m][k][2][0] in[k][H-1][j] + w[m][k][2][2] in[k][H-1][j-1] )      j = 0 .. W - 1
//     out[m][2H][2j + 1] = sum_k s_k   w[m][k][2][1] in[k][H-1][j]
//     out[m][2i + 0][2W] = sum_k s_k ( w[m][k][0][2] in[k][i][W-1] + w[m][k][2][2] in[k][i-1][W-1] )      i = 0 .. H (the corner: i = H)
//     out[m][2i + 1][2W] = sum_k s_k   w[m][k][1][2] in[k][i][W-1]
// (model_spatial_query.py:310-321: conv_transpose2d(stride 2, pad 0); taps as packed: t = 3 ky + kx).  Block = 64 output channels x 64 (32)
// line cells of one sample, 8 waves: a wave owns a 32 x 32 (channel x cell) tile - even and odd outputs, 2 x 16 accumulator registers -
// and a share of every 32-channel chunk's channel PAIRS; per pair three v_mfma_f32_32x32x2_f32 (near, odd, far: bit for bit an fp32 FMA
// chain) fed by five 4-byte LDS reads; the wave groups meet through LDS at the end in a fixed order; the next chunk's global loads are in
// flight during the multiply-adds.  (First versions on the vector ALU: scalar FMAs 54 - 92 us, v_pk_fma_f32 37 - 70 us per launch at the
// two large shapes - packed fp32 issues at half rate here -; with loads inside branches 82 - 130 us, latency-bound.)  Weights from the plain
// fp32 copy behind the split layout (TE_PACK_T6FWD / T6SWAP: Wp[tap][Kp][Mp]), plain fp32 arithmetic, fixed summation order.
constexpr int EM = 64, EKC = 32, ET = 512;
struct T2EdgeArgs {
    float* out; const float* in; const float* wp; const float* isc; const float* osc; const float* bias; int act;
    const float* colbuf;     // optional [B][K][H]: the style-scaled last input column as left by t2s6_kernel (NULL: gathered from `in`)
    int B, K, M, H, W, Kp, Mp, Ho, Wo, row_tiles, line_tiles, mtiles;
};

// CT: line tile of 16 CT cells (4: lines of >= 64 cells; 2: 32-cell tiles)
template <bool ISC, int CT>
__global__ __launch_bounds__(ET) void t2_edge_kernel(const T2EdgeArgs p) {
    constexpr int EC = 16 * CT, XS = EC + 4, NXV = EKC * (EC + 2), NW = 3 * EKC * (EM / 4);      // line tile, LDS row stride, staged values
    constexpr int RW = (NW + ET - 1) / ET, RX = (NXV + ET - 1) / ET;                              // 16-byte weight / 4-byte line loads per thread and chunk
    constexpr int NT = 2 * (EC / 32), KG = 8 / NT;             // 32 x 32 (channel x cell) wave tiles of the block; wave groups that share a chunk's channels
    constexpr int SMEM = 3 * EKC * EM + EKC * XS;              // (>= NT * 64 * 32: the partial tiles of one wave group)
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* ws = smem;                           // [tap: even-near, even-far, odd][k][m]
    float* xs = smem + 3 * EKC * EM;            // [k][cell c0 - 1 .. c0 + EC] (+ pad)
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tw = wid % NT, kg = wid / NT, wm = tw & 1, wc = tw >> 1;      // the wave's tile: channels m0 + 32 wm .., cells c0 + 32 wc ..; its channel group
    int bx = blockIdx.x;
    const int lt = bx % p.line_tiles; bx /= p.line_tiles;
    const int mt = bx % p.mtiles, b = bx / p.mtiles;
    const bool col = lt >= p.row_tiles;                                     // block-uniform: which of the two lines
    const int c0 = (col ? lt - p.row_tiles : lt) * EC, m0 = mt * EM;
    const int NX = col ? p.H : p.W;                                         // input cells of the line
    const int NOUT = col ? 2 * p.H + 1 : 2 * p.W;                           // outputs of the line (the corner belongs to the column)
    const bool compact = col && p.colbuf != nullptr;                        // block-uniform: the column comes from the compact buffer, already scaled
    const int sx = col && !compact ? p.W : 1, so = col ? p.Wo : 1;
    const size_t xbase = compact ? 0 : (col ? (size_t)(p.W - 1) : (size_t)(p.H - 1) * p.W);
    const size_t obase = col ? (size_t)(2 * p.W) : (size_t)(2 * p.H) * p.Wo;
    const int tap_near = col ? 2 : 6, tap_far = 8, tap_odd = col ? 5 : 7;  // even outputs: (ky, kx) = (0, 2) | (2, 0) and (2, 2); odd: (1, 2) | (2, 1)
    const size_t iplane = compact ? (size_t)p.H : (size_t)p.H * p.W, oplane = (size_t)p.Ho * p.Wo;       // (iplane: channel stride of the line source)
    const float* inb = (compact ? p.colbuf : p.in) + (size_t)b * p.K * iplane + xbase;
    const float* iscb = ISC ? p.isc + (size_t)b * p.K : nullptr;
    f32x16 ae, ao;                              // even / odd outputs of the wave's 32 channels x 32 cells
#pragma unroll
    for (int r = 0; r < 16; ++r) { ae[r] = 0.f; ao[r] = 0.f; }

    // per-thread staging geometry (the same for every chunk): weight items e = tid + 512 r -> (tap, k, 4 channels); line items -> (k, cell).
    // Every load is UNCONDITIONAL (clamped address; what does not exist becomes zero when the chunk is written to LDS) and the style scale
    // is applied at that write: a load inside a branch that also consumes it made the compiler wait for each one in turn (five serial
    // round trips per chunk in the first version).
    const float* wsrc[RW];
    int wk[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int e = tid + ET * r, ee = e < NW ? e : 0, t3 = ee / (EKC * 16), k = (ee >> 4) % EKC, m4 = ee & 15;
        const int tap = t3 == 0 ? tap_near : (t3 == 1 ? tap_far : tap_odd);
        wk[r] = e < NW ? k : -1;
        wsrc[r] = p.wp + (size_t)tap * p.Kp * p.Mp + m0 + 4 * m4;
    }
    int xk[RX], xl[RX];
    size_t xoff[RX];
    bool xok[RX];
#pragma unroll
    for (int r = 0; r < RX; ++r) {
        const int e = tid + ET * r, ee = e < NXV ? e : 0, k = ee / (EC + 2), i = ee - k * (EC + 2), c = c0 - 1 + i;
        xk[r] = k;
        xl[r] = e < NXV ? k * XS + i : -1;
        xok[r] = e < NXV && c >= 0 && c < NX;
        xoff[r] = xok[r] ? (size_t)c * sx : 0;
    }
    f32x4 wv[RW];
    float xv[RX], sv[RX];
    auto fetch = [&](int kc) {          // the chunk's loads, all in flight together
#pragma unroll
        for (int r = 0; r < RW; ++r) wv[r] = *reinterpret_cast<const f32x4*>(wsrc[r] + (size_t)min(kc + max(wk[r], 0), p.K - 1) * p.Mp);
#pragma unroll
        for (int r = 0; r < RX; ++r) {
            const int k = min(kc + xk[r], p.K - 1);
            xv[r] = inb[(size_t)k * iplane + xoff[r]];
            sv[r] = (ISC && !compact) ? iscb[k] : 1.f;
        }
    };
    // operands of v_mfma_f32_32x32x2_f32 for channel pair j of the chunk: A[m][kk] = weight of channel 2 j + kk (lane = m + 32 kk),
    // B[kk][n] = line value of cell n (+ 1 for the near tap: xs index 0 is cell c0 - 1)
    const float* wa = ws + half * EM + wm * 32 + l31;           // + (tap * EKC + 2 j) * EM
    const float* xb = xs + half * XS + wc * 32 + l31;           // + 2 j * XS (+ 1)
    fetch(0);
    for (int kc = 0; kc < p.K; kc += EKC) {
        __syncthreads();                  // the previous chunk's reads are done
        // (K % 32 == 16: the upper half of the last chunk is zeros)
#pragma unroll
        for (int r = 0; r < RW; ++r)
            if (wk[r] >= 0) *reinterpret_cast<f32x4*>(&ws[(tid + ET * r) * 4]) = kc + wk[r] < p.K ? wv[r] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < RX; ++r)
            if (xl[r] >= 0) xs[xl[r]] = (xok[r] && kc + xk[r] < p.K) ? xv[r] * sv[r] : 0.f;
        __syncthreads();
        if (kc + EKC < p.K) fetch(kc + EKC);          // in flight during the multiply-adds below
#pragma unroll
        for (int jj = 0; jj < EKC / 2 / KG; ++jj) {
            const int j = kg + KG * jj;                // this wave group's channel pairs of the chunk
            const float an = wa[(0 * EKC + 2 * j) * EM], af = wa[(1 * EKC + 2 * j) * EM], aod = wa[(2 * EKC + 2 * j) * EM];
            const float bn = xb[2 * j * XS + 1], bf = xb[2 * j * XS];
            ae = __builtin_amdgcn_mfma_f32_32x32x2f32(an, bn, ae, 0, 0, 0);
            ao = __builtin_amdgcn_mfma_f32_32x32x2f32(aod, bn, ao, 0, 0, 0);
            ae = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, ae, 0, 0, 0);
        }
    }
    // the wave groups meet in group 0, one after the other (fixed order: ((g0 + g1) + g2) + g3), then the direct kernel's epilogue:
    // demodulation scale, bias, leaky ReLU
    float* red = smem + tw * (64 * 32);         // 32 partial sums per lane of this tile
#pragma unroll
    for (int g = 1; g < KG; ++g) {
        __syncthreads();
        if (kg == g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { red[r * 64 + lane] = ae[r]; red[(16 + r) * 64 + lane] = ao[r]; }
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { ae[r] += red[r * 64 + lane]; ao[r] += red[(16 + r) * 64 + lane]; }
        }
    }
    if (kg != 0) return;
    const float gain = p.act == 3 ? 1.4142135623730951f : 1.f;
    const int o = 2 * (c0 + wc * 32 + l31);     // the lane's cell: outputs o (even) and o + 1 (odd)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float sc = p.osc ? p.osc[(size_t)b * p.M + mm] : 1.f, bi = p.bias ? p.bias[mm] : 0.f;
        float* orow = p.out + ((size_t)b * p.M + mm) * oplane + obase;
        float ve = ae[r] * sc + bi, vo = ao[r] * sc + bi;
        if (p.act >= 3) {
            ve = (ve > 0.f ? ve : ve * 0.2f) * gain;
            vo = (vo > 0.f ? vo : vo * 0.2f) * gain;
        }
        if (o < NOUT) orow[(size_t)o * so] = ve;
        if (o + 1 < NOUT) orow[(size_t)(o + 1) * so] = vo;
    }
}

}  // namespace

#ifdef T2_PROF
extern "C" int te_debug_t2s6_prof(void* host_dst, int64_t bytes) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(te_t2s6_prof_buf), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int te_conv_t2s6_supported(int B, int K, int M, int H, int W) {
    if (!(B > 0 && K >= 32 && K % KC == 0 && M >= BM && M % BM == 0 && H >= TH && H % TH == 0 && W >= TWC && W % TWC == 0)) return 0;
    return ((int64_t)M * (2 * H + 1) * (2 * W + 1) * 4 < 0x7FFFFFFF && (int64_t)K * H * W * 4 < 0x7FFFFFFF &&
            (int64_t)B * (H / TH) * (W / TWC) * (M / BM) < 0x7FFFFFF0) ? 1 : 0;
}

// kernel form of TE_CONV_T2S6: 0 = ping-pong (t2s6_kernel); 1 = the two-image form t2s6q_kernel where M % 128 == 0 and the grid still
// gives every CU a block, the ping-pong form elsewhere; 2 = the two-image form wherever M % 128 == 0 (tests).  Same results bit for bit.
// A process-wide A/B switch like te_conv_wino6_form (te_hip.h); TE_T2S6_FORM in the environment sets the initial value.
static std::atomic<int> g_t2_form{[] { const char* e = getenv("TE_T2S6_FORM"); return e ? atoi(e) : 1; }()};
extern "C" int te_conv_t2s6_form(int form) {
    const int old = g_t2_form.load(std::memory_order_relaxed);
    if (form >= 0 && form <= 2) g_t2_form.store(form, std::memory_order_relaxed);
    return old;
}

// the body cells [0, H) x [0, W) of the transposed convolution (output rows 0 .. 2H - 1, columns 0 .. 2W - 1); conv.hip adds the last
// output row and column as two thin regions of the fp32 kernel
int te_t2s6_launch(float* out, const float* in, const float* U, const float* isc, const float* osc, const float* bias, int act,
                   int B, int K, int M, int H, int W, hipStream_t s, float* colbuf) {
    TE_REQUIRE(te_conv_t2s6_supported(B, K, M, H, W), TE_ERR_UNSUPPORTED,
               "te_conv_f32(TE_CONV_T2S6): needs K %% 16 == 0 (>= 32), M %% 64 == 0, H %% 8 == 0, W %% 16 == 0 (te_conv_t2s6_supported)");
    TE_REQUIRE((reinterpret_cast<uintptr_t>(U) & 15) == 0 && (reinterpret_cast<uintptr_t>(in) & 3) == 0, TE_ERR_UNSUPPORTED,
               "te_conv_f32(TE_CONV_T2S6): 16-byte aligned packed weights required");
    T2Args a{};
    a.out = out; a.in = in; a.U = reinterpret_cast<const u32x4*>(U); a.isc = isc; a.osc = osc; a.bias = bias; a.act = act; a.colbuf = colbuf;
    a.B = B; a.K = K; a.M = M; a.H = H; a.W = W; a.Ho = 2 * H + 1; a.Wo = 2 * W + 1;
    a.tiles_x = W / TWC; a.tiles_y = H / TH; a.mblocks = M / BM;
    a.ntiles = B * a.tiles_x * a.tiles_y;
    a.nt8 = te::xcd_banded() ? (int)te::cdiv(a.ntiles, 8) : 0;
    const int64_t blocks = te::cdiv(a.ntiles, 8) * 8 * a.mblocks;
    const size_t lds = 2 * (size_t)U_SLOTS * 1024 + 2 * (size_t)TP_DWORDS * 4;
    static std::atomic<uint64_t> attr_done{0};
    const int form = g_t2_form.load(std::memory_order_relaxed);
    const int64_t blocks_q = te::cdiv(a.ntiles, 8) * 8 * (M / (2 * BM));
    if (form >= 1 && M % (2 * BM) == 0 && (form == 2 || blocks_q >= te::kNumCU)) {
        static std::atomic<uint64_t> attr_done_q{0}, attr_done_qs{0};
        a.mblocks = M / (2 * BM);
        if (isc) {
            te::allow_big_lds(attr_done_qs, (const void*)t2s6q_kernel<true>, 160 * 1024);
            t2s6q_kernel<true><<<dim3((unsigned)blocks_q), WT, lds, s>>>(a);
        } else {
            te::allow_big_lds(attr_done_q, (const void*)t2s6q_kernel<false>, 160 * 1024);
            t2s6q_kernel<false><<<dim3((unsigned)blocks_q), WT, lds, s>>>(a);
        }
    } else if (isc) {
        static std::atomic<uint64_t> attr_done_sc{0};
        te::allow_big_lds(attr_done_sc, (const void*)t2s6_kernel<true>, 160 * 1024);
        t2s6_kernel<true><<<dim3((unsigned)blocks), WT, lds, s>>>(a);
    } else {
        te::allow_big_lds(attr_done, (const void*)t2s6_kernel<false>, 160 * 1024);
        t2s6_kernel<false><<<dim3((unsigned)blocks), WT, lds, s>>>(a);
    }
    return te::launch_status("te_conv_f32(TE_CONV_T2S6)");
}

// the last output row and column of the transposed kind (t2_edge_kernel); wplain = the Wp[tap][Kp][Mp] copy of the weights
int te_t2s6_edge_launch(float* out, const float* in, const float* wplain, const float* isc, const float* osc, const float* bias, int act,
                        int B, int K, int M, int H, int W, int Kp, int Mp, hipStream_t s, const float* colbuf) {
    TE_REQUIRE(K % 16 == 0 && M % EM == 0 && (reinterpret_cast<uintptr_t>(wplain) & 15) == 0 && Mp % 4 == 0, TE_ERR_UNSUPPORTED,
               "te_conv_f32(TE_CONV_T2S6): the edge kernel needs K %% 16 == 0, M %% 64 == 0 and 16-byte aligned weights");
    T2EdgeArgs a{};
    a.out = out; a.in = in; a.wp = wplain; a.isc = isc; a.osc = osc; a.bias = bias; a.act = act; a.colbuf = colbuf;
    a.B = B; a.K = K; a.M = M; a.H = H; a.W = W; a.Kp = Kp; a.Mp = Mp; a.Ho = 2 * H + 1; a.Wo = 2 * W + 1;
    const bool wide = W >= 64 && H >= 64;      // (32-cell tiles for every shape measured the same: 43.5 us against 42.2 us per launch on average)
    const int EC = wide ? 64 : 32;
    a.row_tiles = (int)te::cdiv(W, EC);                       // cells 0 .. W - 1 of the last output row
    a.line_tiles = a.row_tiles + (int)te::cdiv(H + 1, EC);    // + cells 0 .. H of the last output column
    a.mtiles = M / EM;
    const int64_t blocks = (int64_t)a.line_tiles * a.mtiles * B;
    TE_REQUIRE(blocks < 0x7FFFFFF0, TE_ERR_UNSUPPORTED, "te_conv_f32(TE_CONV_T2S6): too many edge blocks");
    if (wide) {
        if (isc) t2_edge_kernel<true, 4><<<dim3((unsigned)blocks), ET, 0, s>>>(a);
        else t2_edge_kernel<false, 4><<<dim3((unsigned)blocks), ET, 0, s>>>(a);
    } else {
        if (isc) t2_edge_kernel<true, 2><<<dim3((unsigned)blocks), ET, 0, s>>>(a);
        else t2_edge_kernel<false, 2><<<dim3((unsigned)blocks), ET, 0, s>>>(a);
    }
    return te::launch_status("te_conv_f32(TE_CONV_T2S6) edge");
}
