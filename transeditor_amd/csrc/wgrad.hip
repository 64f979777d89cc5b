// F1 (backward): weight-gradient correlation of the modulated-convolution family on the fp32 matrix pipe,
// plus the reduction that turns the per-sample correlation slabs into dW, d(style scale) and d(demod scale).
//
//   slab[b][s][co][ci][tap] = sum_{cells of chunk s of sample b}  g[b,co, gpos(cell,tap)] * x[b,ci, xpos(cell,tap)]
//
//   3X3 / 1X1 : cells = output pixels p;   g at p (plain operand),           x at p + tap - pad (shifted operand)
//   T2        : cells = low-res pixels;    g at (2i+ky, 2j+kx) (shifted),    x at (i,j) (plain)
//
// GEMM view: rows = co (A operand = g), cols = ci (B operand = x), reduction = cells, one 32x32 accumulator
// per tap -> each wave owns 1 x 1 x NTAP tiles (144 accumulator registers for 3x3; 192 in the pair form of the 3x3
// kernel, which trades 18 MFMAs per 4 cells for 12 - see wgrad_mfma_kernel); a block of 4 waves covers 64 co x 64 ci.  The reduction over cells is split over blocks (per sample b, S chunks per sample) and every
// block writes its own slab: no atomics, deterministic.  Keeping the slabs PER SAMPLE with NO modulation
// applied is what lets one pass over the activations produce all three gradients (te_wgrad_reduce_f32):
//   dW[co,ci,t]  = sum_b osc[b,co] isc[b,ci] slab_b      d isc[b,ci] = sum_{co,t} W osc[b,co] slab_b
//   d osc[b,co]  = sum_{ci,t} W isc[b,ci] slab_b
// LDS tiles are [channel][position] with an ODD channel stride, so the 32 lanes of an MFMA operand read
// (32 different channels, same position) hit 32 different banks.
#include "te_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int QCH = 64;         // channels of the shifted operand per block (2 waves)
// NWP waves along the plain operand's channels (4 -> 512 threads, 128 x 64 tile; 2 -> 256 threads, 64 x 64 tile)
constexpr unsigned OOB = 0x80000000u;   // buffer offset beyond num_records: the load returns 0

struct WgArgs {
    float* slabs;
    const float* g;
    const float* x;
    int B, Co, Ci, H, W;     // low-res (cell grid) size
    int Hg, Wg, Hx, Wx;      // spatial size of g and x
    int S;                   // chunks per sample group
    int NB;                  // samples per group (1: a slab per sample; > 1: the plain, unmodulated gradient of small images -
                             // a block walks the cell tiles of NB consecutive samples and the group shares one slab)
    int TH, TW, lgTW, NC, lgNC;   // cell tile (NC = TH*TW cells per stage)
    int tiles_x, tiles_y;
    int QH, QW, QS;          // shifted-operand tile: rows, cols, channel stride (odd)
    int PS;                  // plain-operand channel stride (odd)
    unsigned magic_qt, magic_qw;   // ceil(2^32 / (QH*QW)), ceil(2^32 / QW): exact division of small indices
    unsigned magic_q8, magic_q2;   // ceil(2^32 / (QH*8)), ceil(2^32 / (QH*2))
    unsigned magic_q16, magic_q1;  // ceil(2^32 / (QH*16)), ceil(2^32 / QH)   (transposed kind: 65-wide rows = 16 x 16 B + 1)
    int vec;                       // 16-byte staging path: TW == 32 and W % 32 == 0 (rows of both operands 16 B aligned)
};

template <int KIND> struct WK;
// NP = plain elements per thread = 32*NCELL/128 (independent of NWP); NQ8 = shifted elements per thread at 512 threads
template <> struct WK<TE_CONV_3X3> { static constexpr int NT = 9, NCELL = 64, NP = 16, NQ8 = 17; };
template <> struct WK<TE_CONV_1X1> { static constexpr int NT = 1, NCELL = 64, NP = 16, NQ8 = 8; };
template <> struct WK<TE_CONV_T2>  { static constexpr int NT = 9, NCELL = 32, NP = 8,  NQ8 = 25; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

#ifdef TE_CONV_PROF
// experimental builds: per-phase cycle counts (s_memtime) of every wave, read back with te_debug_wgrad_prof
__device__ unsigned long long te_wgrad_prof_buf[8192 * 8];
#define WPROF(i) { const unsigned long long t_ = __builtin_readcyclecounter(); pc[i] += t_ - tlast; tlast = t_; }
#else
#define WPROF(i)
#endif

// WINO (3x3 only): the cells of a stage are taken as horizontal PAIRS and the three taps of a kernel row come out of FOUR
// products per pair instead of six - the 1-D Winograd form F(3,2), the transpose of the F(2,3) the forward kernel (wino.hip)
// uses.  For the pair (g0, g1) of the output gradient and the four inputs e0..e3 under it (one tap row ky):
//     u = [g0, g0+g1, g0-g1, -g1]     t = [e0-e2, e1+e2, e2-e1, e1-e3]     m_c = sum over pairs u_c t_c   (one accumulator each)
//     dW[ky][0] = m0 + (m1+m2)/2      dW[ky][1] = (m1-m2)/2                dW[ky][2] = (m1+m2)/2 + m3
// so a k-step of the matrix pipe (two pairs: lane half h takes pair 2*ks + h) issues 12 MFMAs for 4 cells where the direct form
// issues 18.  The 12 accumulators are folded back to the 9 taps when the slab is written: slab format, reducers and callers
// are the same as for the direct form.
template <int KIND, int NWP, bool WINO>
__global__ __launch_bounds__(NWP * 128, 2) void wgrad_mfma_kernel(const WgArgs p) {
    static_assert(!WINO || KIND == TE_CONV_3X3, "pair form: 3x3 only");
    constexpr int NTHREADS = NWP * 128;
    constexpr int PCH = NWP * 32;
    constexpr int NT = WK<KIND>::NT;
    constexpr int NACC = WINO ? 12 : NT;
    constexpr bool GSHIFT = (KIND == TE_CONV_T2);   // which operand carries the tap shift
    constexpr int NP = WK<KIND>::NP;                // plain-tile elements per thread   (PCH*NC / 512)
    constexpr int NQ = (WK<KIND>::NQ8 * 512 + NTHREADS - 1) / NTHREADS;   // shifted-tile elements per thread

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* pl = smem;                      // plain operand   [PCH][PS]
    float* ql = smem + PCH * p.PS;         // shifted operand [QCH][QS]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wp = wid >> 1, wq = wid & 1;          // wave -> (plain channel block, shifted channel block)

    const int s_chunk = blockIdx.x % p.S, bgrp = blockIdx.x / p.S, b = bgrp * p.NB;     // b: first sample of the group
    // rows of the result = co (A operand = g), cols = ci (B operand = x)
    // (an XCD-aware 1-D block order that runs the channel blocks of a chunk back to back on one XCD was measured in round 4:
    // same time, same FETCH_SIZE - profiles/experiments/r04_wgrad_xcd_ab.log - and dropped)
    const int co0 = blockIdx.y * (GSHIFT ? QCH : PCH), ci0 = blockIdx.z * (GSHIFT ? PCH : QCH);

    const float* gP = p.g + (size_t)b * p.Co * p.Hg * p.Wg;
    const float* xP = p.x + (size_t)b * p.Ci * p.Hx * p.Wx;
    const int pC = GSHIFT ? p.Ci : p.Co, qC = GSHIFT ? p.Co : p.Ci;
    const int pc0 = GSHIFT ? ci0 : co0, qc0 = GSHIFT ? co0 : ci0;
    const int pH = GSHIFT ? p.Hx : p.Hg, pW = GSHIFT ? p.Wx : p.Wg;
    const int qH = GSHIFT ? p.Hg : p.Hx, qW = GSHIFT ? p.Wg : p.Wx;
    // one buffer descriptor per operand and sample: 32-bit offsets, hardware zero fill for masked elements
    // (a group's descriptors span its NB samples: host-checked to stay below the 2 GiB the zero-fill offset needs)
    const unsigned p_sample = (unsigned)pC * pH * pW * 4u, q_sample = (unsigned)qC * qH * qW * 4u;      // bytes per sample
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(GSHIFT ? xP : gP, p_sample * (unsigned)p.NB);
    const __amdgpu_buffer_rsrc_t qrs = make_rsrc(GSHIFT ? gP : xP, q_sample * (unsigned)p.NB);
    const int q_tile = p.QH * p.QW;

    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Narrow layers (4-wave tile only): when a side of the block tile has a single valid 32-channel block (32- / 48-channel
    // layers of the 512 / 1024 px generator tail, or the last block of a 96-channel side), the waves that would multiply
    // zero padding take a share of the staged cells instead (K split 2 or 4 ways) and the partial tiles are combined
    // through LDS before the slab is written.
    int pb = wp, qb = wq, kw = 0, kstride = 1, ngrp = 1, grp = 0;
    if constexpr (NWP == 2) {
        const int nPB = (min(PCH, pC - pc0) + 31) >> 5, nQB = (min(QCH, qC - qc0) + 31) >> 5;
        const int kp = (nPB >= 2) ? 1 : 2, kq = (nQB >= 2) ? 1 : 2;
        pb = (nPB >= 2) ? wp : 0;
        qb = (nQB >= 2) ? wq : 0;
        kstride = kp * kq;
        kw = ((nPB >= 2) ? 0 : wp) * kq + ((nQB >= 2) ? 0 : wq);
        ngrp = 4 / kstride;
        grp = (nPB >= 2) ? ((nQB >= 2) ? wid : wp) : ((nQB >= 2) ? wq : 0);
    }
    // lane bases inside the LDS tiles: g is the A operand, x the B operand
    const int a_ch = (GSHIFT ? qb : pb) * 32 + l31;     // co of this lane inside the block tile
    const int b_ch = (GSHIFT ? pb : qb) * 32 + l31;     // ci of this lane inside the block tile

    const int tiles_1 = p.tiles_x * p.tiles_y;      // cell tiles of one sample
    const int n_tiles = tiles_1 * p.NB;             // tile index -> (sample of the group, cell tile)
    const int t_begin = (int)((int64_t)n_tiles * s_chunk / p.S), t_end = (int)((int64_t)n_tiles * (s_chunk + 1) / p.S);

    float preg[NP], qreg[NQ];
#ifdef TE_CONV_PROF
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
    const unsigned long long pstart = tlast;
#endif
    // ---- tile loop, double-buffered LDS (two [plain | shifted] operand images, <= 2 x 68 KB): while the MFMAs of tile t run
    // on one image, tile t+1 (already in registers) is written to the other and the loads of tile t+2 are issued, so a
    // tile costs ONE workgroup barrier and no phase in which the matrix pipe waits for staging.
    constexpr bool QVEC = (KIND != TE_CONV_T2);     // the shifted operand of T2 has odd row width: scalar loads
    constexpr int NP4 = NP / 4, NQ4 = (KIND == TE_CONV_3X3) ? (NQ - 1) / 4 : NQ / 4;
    constexpr bool TVEC = (KIND == TE_CONV_T2);
    const int bufsz = PCH * p.PS + QCH * p.QS;      // floats of one operand image
    // ---- 16-byte staging path (p.vec: full 32-cell rows, NC == NCELL): every stride is a compile-time constant and a thread's
    // loads differ by uniform steps, so a tile costs a handful of vector-ALU instructions of address work (each of them has to
    // find an issue slot between the other wave's MFMAs: ~50 cycles apiece when the matrix pipe is saturated).
    //   plain operand  : float4 index tid + NTHREADS * r -> channel step CSTEP per r, same cells
    //   shifted operand: 8 threads per channel (sub = tid & 7), load r = tile row r (T2: row r >> 1, half r & 1), segment sub
    constexpr int NCV = WK<KIND>::NCELL, THV = NCV / 32, PSV = NCV | 1;
    constexpr int QHV = (KIND == TE_CONV_3X3) ? THV + 2 : (KIND == TE_CONV_T2 ? 2 * THV + 1 : THV);
    constexpr int QWV = (KIND == TE_CONV_3X3) ? 34 : (KIND == TE_CONV_T2 ? 65 : 32);
    constexpr int QSV = (QHV * QWV) | 1;
    constexpr int FP4 = NCV / 4, CSTEP = NTHREADS / FP4;            // plain: float4 per channel, channels per sweep
    constexpr int QSW = (QCH * 8 + NTHREADS - 1) / NTHREADS;        // shifted: sweeps of NTHREADS / 8 channels
    constexpr int QRL = (KIND == TE_CONV_T2) ? 2 * QHV : QHV;       // 16-byte loads per thread and sweep
    static_assert(NP4 * CSTEP == PCH && QSW * QRL * 4 + (KIND == TE_CONV_1X1 ? 0 : QSW) <= NQ, "staging registers");
    const int v_chl0 = tid / FP4, v_c4 = (tid % FP4) * 4;
    const int v_pgo = ((pc0 + v_chl0) * pH + (v_c4 >> 5)) * pW + (v_c4 & 31);        // elements, tile origin excluded
    const int v_plo = v_chl0 * PSV + v_c4;
    const int v_qch = tid >> 3, v_sub = tid & 7;
    const int v_qgo = (qc0 + v_qch) * qH * qW + 4 * v_sub;
    const int v_qlo = v_qch * QSV + ((KIND == TE_CONV_3X3) ? 1 : 0) + 4 * v_sub;
    auto issue_vec = [&](int tn) {
        unsigned pso = 0, qso = 0;                       // byte offset of the tile's sample inside the group
        if (p.NB > 1) { const int bb = tn / tiles_1; tn -= bb * tiles_1; pso = bb * p_sample; qso = bb * q_sample; }
        const int ty0 = (tn / p.tiles_x) * p.TH, tx0 = (tn % p.tiles_x) * p.TW;   // first cell of the tile
        const int qy0 = (KIND == TE_CONV_3X3) ? ty0 - 1 : (KIND == TE_CONV_T2 ? 2 * ty0 : ty0);
        const int qx0 = (KIND == TE_CONV_T2) ? 2 * tx0 : tx0;
        const int pb4 = (v_pgo + ty0 * pW + tx0) * 4;
#pragma unroll
        for (int r = 0; r < NP4; ++r) {
            // channels past pC: beyond num_records -> 0 (one sample per group; a group's layers have whole channel blocks, host-checked)
            const f32x4 v = buf_load4(prs, pso + (unsigned)(pb4 + r * CSTEP * pH * pW * 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) preg[4 * r + j] = v[j];
        }
        const int qb = v_qgo + qy0 * qW + qx0;          // may be "negative" (row -1 of channel 0): only used when the row is valid
#pragma unroll
        for (int sw = 0; sw < QSW; ++sw) {
#pragma unroll
            for (int r = 0; r < QRL; ++r) {
                const int ry = (KIND == TE_CONV_T2) ? (r >> 1) : r, hx = (KIND == TE_CONV_T2) ? 32 * (r & 1) : 0;
                const bool rok = (unsigned)(qy0 + ry) < (unsigned)qH;                       // wave-uniform
                const f32x4 v = buf_load4(qrs, rok ? qso + (unsigned)((qb + sw * (NTHREADS / 8) * qH * qW + ry * qW + hx) * 4) : OOB);
#pragma unroll
                for (int j = 0; j < 4; ++j) qreg[4 * (sw * QRL + r) + j] = v[j];
            }
        }
        if (KIND == TE_CONV_3X3) {        // the two halo columns of every (channel, row): thread sub -> (row sub >> 1, side sub & 1)
            const int ry = v_sub >> 1, side = v_sub & 1;
            const bool ok = (unsigned)(qy0 + ry) < (unsigned)qH && (side ? tx0 + 32 < qW : tx0 > 0);
#pragma unroll
            for (int sw = 0; sw < QSW; ++sw)
                qreg[4 * QSW * QRL + sw] = buf_load(qrs, ok ? qso + (unsigned)((qb - 4 * v_sub + sw * (NTHREADS / 8) * qH * qW + ry * qW + (side ? 32 : -1)) * 4) : OOB);
        } else if (KIND == TE_CONV_T2) {  // column 64 of the 65-wide rows: threads with sub < 3 take row sub
            const bool ok = v_sub < QHV;
#pragma unroll
            for (int sw = 0; sw < QSW; ++sw)
                qreg[4 * QSW * QRL + sw] = buf_load(qrs, ok ? qso + (unsigned)((qb - 4 * v_sub + sw * (NTHREADS / 8) * qH * qW + v_sub * qW + 64) * 4) : OOB);
        }
    };
    auto commit_vec = [&](float* pl, float* ql) {
#pragma unroll
        for (int r = 0; r < NP4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) pl[v_plo + r * CSTEP * PSV + j] = preg[4 * r + j];
#pragma unroll
        for (int sw = 0; sw < QSW; ++sw) {
#pragma unroll
            for (int r = 0; r < QRL; ++r) {
                const int ry = (KIND == TE_CONV_T2) ? (r >> 1) : r, hx = (KIND == TE_CONV_T2) ? 32 * (r & 1) : 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) ql[v_qlo + sw * (NTHREADS / 8) * QSV + ry * QWV + hx + j] = qreg[4 * (sw * QRL + r) + j];
            }
        }
        if (KIND == TE_CONV_3X3) {
            const int ry = v_sub >> 1, side = v_sub & 1;
#pragma unroll
            for (int sw = 0; sw < QSW; ++sw)
                ql[(v_qch + sw * (NTHREADS / 8)) * QSV + ry * QWV + (side ? QWV - 1 : 0)] = qreg[4 * QSW * QRL + sw];
        } else if (KIND == TE_CONV_T2) {
            if (v_sub < QHV) {
#pragma unroll
                for (int sw = 0; sw < QSW; ++sw) ql[(v_qch + sw * (NTHREADS / 8)) * QSV + v_sub * QWV + 64] = qreg[4 * QSW * QRL + sw];
            }
        }
    };
    auto commit = [&](float* pl, float* ql) {       // staged registers -> one LDS image
        if (p.vec) {
            commit_vec(pl, ql);
        } else {
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                int e = tid + NTHREADS * r;
                asm volatile("" : "+v"(e));      // keep the index math inside the loop (hoisting it costs ~100 VGPRs)
                const int ch = e >> p.lgNC;
                if (ch < PCH) pl[ch * p.PS + (e & (p.NC - 1))] = preg[r];
            }
#pragma unroll
            for (int r = 0; r < NQ; ++r) {
                unsigned e = tid + NTHREADS * r;
                asm volatile("" : "+v"(e));
                const unsigned ch = __umulhi(e, p.magic_qt), rem = e - ch * q_tile;
                if (ch < QCH) ql[ch * p.QS + rem] = qreg[r];       // tile stored [row][col] with stride QW == linear rem
            }
        }
    };
    auto issue = [&](int tn) {                      // global loads of tile tn -> registers
        if (p.vec) { issue_vec(tn); return; }
        unsigned pso = 0, qso = 0;
        if (p.NB > 1) { const int bb = tn / tiles_1; tn -= bb * tiles_1; pso = bb * p_sample; qso = bb * q_sample; }
        const int ty0 = (tn / p.tiles_x) * p.TH, tx0 = (tn % p.tiles_x) * p.TW;   // first cell of the tile
        int qy0, qx0;
        if (KIND == TE_CONV_3X3) { qy0 = ty0 - 1; qx0 = tx0 - 1; }
        else if (KIND == TE_CONV_T2) { qy0 = 2 * ty0; qx0 = 2 * tx0; }
        else { qy0 = ty0; qx0 = tx0; }
        {
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                int e = tid + NTHREADS * r;
                asm volatile("" : "+v"(e));
                const int chl = e >> p.lgNC, cell = e & (p.NC - 1);
                const int y = ty0 + (cell >> p.lgTW), xx = tx0 + (cell & (p.TW - 1)), ch = pc0 + chl;
                const bool ok = chl < PCH && y < p.H && xx < p.W && ch < pC;    // plain operand lives on the cell grid
                preg[r] = buf_load(prs, ok ? pso + (unsigned)((ch * pH + y) * pW + xx) * 4u : OOB);
            }
#pragma unroll
            for (int r = 0; r < NQ; ++r) {
                unsigned e = tid + NTHREADS * r;
                asm volatile("" : "+v"(e));
                const unsigned chl = __umulhi(e, p.magic_qt), rem = e - chl * q_tile;
                const unsigned ry = __umulhi(rem, p.magic_qw), rx = rem - ry * p.QW;
                const int y = qy0 + (int)ry, xx = qx0 + (int)rx, ch = qc0 + (int)chl;
                const bool ok = chl < QCH && (unsigned)y < (unsigned)qH && (unsigned)xx < (unsigned)qW && ch < qC;
                qreg[r] = buf_load(qrs, ok ? qso + (unsigned)((ch * qH + y) * qW + xx) * 4u : OOB);
            }
            }
    };

    // ---- MFMAs over the cells of a staged tile (2 cells per k-step: lane half h takes cell 2*ks + h);
    //      operands of step ks+1 are read from LDS before the MFMAs of step ks are issued
    const int nks = WINO ? p.NC >> 2 : p.NC >> 1;
    float a_c[GSHIFT ? NT : 1], b_c[GSHIFT ? 1 : NT];
    auto pos_shift = [&](int ks) {
        const int c0 = 2 * ks, cy = c0 >> p.lgTW, cx = c0 & (p.TW - 1);
        return (KIND == TE_CONV_T2) ? 2 * cy * p.QW + 2 * cx : cy * p.QW + cx;
    };
    float w_g0 = 0.f, w_g1 = 0.f, w_e0 = 0.f, w_e1 = 0.f, w_e2 = 0.f, w_e3 = 0.f;     // pair form: operands in flight
    int w_xo = 0;
    auto first_operands = [&](const float* g_l, const float* x_l) {
        if constexpr (WINO) {
            const int c0 = 4 * min(kw, nks - 1) + 2 * half;         // first cell of this lane's pair
            w_xo = (c0 >> p.lgTW) * p.QW + (c0 & (p.TW - 1));
            w_g0 = g_l[c0]; w_g1 = g_l[c0 + 1];
            w_e0 = x_l[w_xo]; w_e1 = x_l[w_xo + 1]; w_e2 = x_l[w_xo + 2]; w_e3 = x_l[w_xo + 3];
            return;
        }
        const int k0 = min(kw, nks - 1);
        const int sh = pos_shift(k0);
        if (!GSHIFT) {
            a_c[0] = g_l[2 * k0];
#pragma unroll
            for (int t = 0; t < NT; ++t) b_c[t] = x_l[sh + ((NT == 1) ? 0 : (t / 3) * p.QW + (t % 3))];
        } else {
            b_c[0] = x_l[2 * k0];
#pragma unroll
            for (int t = 0; t < NT; ++t) a_c[t] = g_l[sh + (t / 3) * p.QW + (t % 3)];
        }
    };
    auto steps = [&](const float* g_l, const float* x_l, int ks_from, int ks_to) {      // k-steps ks_from <= ks < ks_to
        if constexpr (WINO) {
            // software pipeline at tap-row granularity: the four inputs of the NEXT tap row (after row 2: row 0 and the g pair of
            // the next k-step) are read from LDS before the four MFMAs of the current row are issued
#pragma unroll 1
            for (int ks = ks_from; ks < ks_to; ks += kstride) {
                const int kn = (ks + kstride < nks) ? ks + kstride : ks;          // last step re-reads itself (harmless)
                const int cn = 4 * kn + 2 * half, xo_n = (cn >> p.lgTW) * p.QW + (cn & (p.TW - 1));
                const float* xr = x_l + w_xo;
                const float u0 = w_g0, u1 = w_g0 + w_g1, u2 = w_g0 - w_g1, u3 = -w_g1;
                float t0 = w_e0 - w_e2, t1 = w_e1 + w_e2, t2 = w_e2 - w_e1, t3 = w_e1 - w_e3;
                w_e0 = xr[p.QW]; w_e1 = xr[p.QW + 1]; w_e2 = xr[p.QW + 2]; w_e3 = xr[p.QW + 3];
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, t0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1, t1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2, t2, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(u3, t3, acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                t0 = w_e0 - w_e2; t1 = w_e1 + w_e2; t2 = w_e2 - w_e1; t3 = w_e1 - w_e3;
                w_e0 = xr[2 * p.QW]; w_e1 = xr[2 * p.QW + 1]; w_e2 = xr[2 * p.QW + 2]; w_e3 = xr[2 * p.QW + 3];
                __builtin_amdgcn_sched_barrier(0);
                acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, t0, acc[4], 0, 0, 0);
                acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1, t1, acc[5], 0, 0, 0);
                acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2, t2, acc[6], 0, 0, 0);
                acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(u3, t3, acc[7], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                t0 = w_e0 - w_e2; t1 = w_e1 + w_e2; t2 = w_e2 - w_e1; t3 = w_e1 - w_e3;
                w_e0 = x_l[xo_n]; w_e1 = x_l[xo_n + 1]; w_e2 = x_l[xo_n + 2]; w_e3 = x_l[xo_n + 3];
                w_g0 = g_l[cn]; w_g1 = g_l[cn + 1];
                w_xo = xo_n;
                __builtin_amdgcn_sched_barrier(0);
                acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, t0, acc[8], 0, 0, 0);
                acc[9] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1, t1, acc[9], 0, 0, 0);
                acc[10] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2, t2, acc[10], 0, 0, 0);
                acc[11] = __builtin_amdgcn_mfma_f32_32x32x2f32(u3, t3, acc[11], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#pragma unroll 2
        for (int ks = ks_from; ks < ks_to; ks += kstride) {
            float a_n[GSHIFT ? NT : 1], b_n[GSHIFT ? 1 : NT];
            const int kn = (ks + kstride < nks) ? ks + kstride : ks;          // last step re-reads itself (harmless)
            const int sh = pos_shift(kn);
            if (!GSHIFT) {
                a_n[0] = g_l[2 * kn];
#pragma unroll
                for (int t = 0; t < NT; ++t) b_n[t] = x_l[sh + ((NT == 1) ? 0 : (t / 3) * p.QW + (t % 3))];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[0], b_c[t], acc[t], 0, 0, 0);
                a_c[0] = a_n[0];
#pragma unroll
                for (int t = 0; t < NT; ++t) b_c[t] = b_n[t];
            } else {
                b_n[0] = x_l[2 * kn];
#pragma unroll
                for (int t = 0; t < NT; ++t) a_n[t] = g_l[sh + (t / 3) * p.QW + (t % 3)];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[t], b_c[0], acc[t], 0, 0, 0);
                b_c[0] = b_n[0];
#pragma unroll
                for (int t = 0; t < NT; ++t) a_c[t] = a_n[t];
            }
        }
    };

    // (pair form: the lane half selects a PAIR, applied inside the step)
    const int g_off = a_ch * (GSHIFT ? p.QS : p.PS) + (WINO ? 0 : (GSHIFT ? 2 * half : half)) + (GSHIFT ? PCH * p.PS : 0);   // inside an image
    const int x_off = b_ch * (GSHIFT ? p.PS : p.QS) + (WINO ? 0 : half) + (GSHIFT ? 0 : PCH * p.PS);
    if (t_begin < t_end) {
        issue(t_begin);
        commit(smem, smem + PCH * p.PS);
        if (t_begin + 1 < t_end) issue(t_begin + 1);
    }
    __syncthreads();
    // split points of the k-step range (multiples of kstride past kw)
    const int nst = (nks - kw + kstride - 1) / kstride;
    const int ks1 = kw + (nst / 4) * kstride, ks2 = kw + (nst / 2) * kstride;      // (offsetting the split points between the two
                                                                                      // waves of a SIMD measured slower: 124 vs 130 TFLOP/s)
    // 1x1 (one MFMA per k-step: too few to hide anything behind) and the 4-wave tile of the narrow layers (two images would
    // leave room for ONE block = one wave per SIMD on a CU; measured 1.17 vs 0.89 ms at 64 -> 64 @512^2) keep a single image and
    // two barriers per tile, so that several blocks share a CU instead
    constexpr bool DB = (KIND != TE_CONV_1X1) && (NWP == 4);
    for (int tl = t_begin; tl < t_end; ++tl) {
        WPROF(5)
        float* cur = smem + (DB ? ((tl - t_begin) & 1) * bufsz : 0);
        float* nxt = smem + (DB ? (((tl - t_begin) & 1) ^ 1) * bufsz : 0);
        const float* g_l = cur + g_off;
        const float* x_l = cur + x_off;
        first_operands(g_l, x_l);
        if (DB) {
            steps(g_l, x_l, kw, ks1);
            WPROF(4)
            if (tl + 1 < t_end) commit(nxt, nxt + PCH * p.PS);      // tile tl+1: loaded while tile tl-1 was multiplied
            WPROF(1)
            steps(g_l, x_l, ks1, ks2);
            WPROF(4)
            if (tl + 2 < t_end) issue(tl + 2);
            WPROF(3)
            steps(g_l, x_l, ks2, nks);
            WPROF(4)
        } else {
            steps(g_l, x_l, kw, nks);
            __syncthreads();                // everyone is done reading the image
            if (tl + 1 < t_end) commit(nxt, nxt + PCH * p.PS);
            if (tl + 2 < t_end) issue(tl + 2);
        }
        __syncthreads();                    // everyone is done reading `cur` and writing `nxt`
        WPROF(0)
    }
#ifdef TE_CONV_PROF
    {
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (lane == 0 && lin * (NTHREADS / 64) + wid < 8192) {
            unsigned long long* d = te_wgrad_prof_buf + ((size_t)lin * (NTHREADS / 64) + wid) * 8;
            for (int i = 0; i < 6; ++i) d[i] = pc[i];
            d[6] = __builtin_readcyclecounter() - pstart;
            d[7] = t_end - t_begin;
        }
    }
#endif

    if constexpr (NWP == 2) {
        if (kstride > 1) {        // block-uniform: combine the K-split partial tiles, one round per extra share
            float* red = smem + (size_t)grp * (NACC * 16 * 64);
            for (int rnd = 1; rnd < kstride; ++rnd) {
                __syncthreads();
                if (kw == rnd) {
#pragma unroll
                    for (int t = 0; t < NACC; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
                }
                __syncthreads();
                if (kw == 0) {
#pragma unroll
                    for (int t = 0; t < NACC; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][r] += red[(t * 16 + r) * 64 + lane];
                }
            }
            if (kw != 0) return;
        }
    }
    (void)ngrp;

    // ---- write the slab tile: rows = co, cols = ci;  slab[b][s][co][ci][tap]
    float* sl = p.slabs + ((size_t)bgrp * p.S + s_chunk) * p.Co * p.Ci * NT;
    const int ci = ci0 + b_ch;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + (GSHIFT ? qb : pb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < p.Co && ci < p.Ci) {
            float* dst = sl + ((size_t)co * p.Ci + ci) * NT;
            if constexpr (WINO) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float hs = 0.5f * (acc[4 * ky + 1][r] + acc[4 * ky + 2][r]);
                    dst[3 * ky + 0] = acc[4 * ky + 0][r] + hs;
                    dst[3 * ky + 1] = 0.5f * (acc[4 * ky + 1][r] - acc[4 * ky + 2][r]);
                    dst[3 * ky + 2] = hs + acc[4 * ky + 3][r];
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) dst[t] = acc[t][r];
            }
        }
    }
}

// block tile: 8 waves (128 x 64 channels) by default; narrow layers (both channel counts <= 64, the 512 / 1024 px
// generator tail) use the 4-wave 64 x 64 tile so half of the plain-operand rows are not zero padding

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline int pow2ceil(int v) { return 1 << ilog2(v); }
inline unsigned magic(unsigned d) { return (unsigned)(((1ull << 32) + d - 1) / d); }   // exact floor(e/d) for e < 2^16, d < 2^16

template <int KIND>
bool fill_geometry(WgArgs& a) {
    constexpr int NCELL = WK<KIND>::NCELL;
    a.TW = std::max(2, std::min(32, pow2ceil(a.W)));      // a k-step multiplies two horizontally adjacent cells: 1-pixel-wide
                                                          // images get a second, masked column
    a.TH = std::max(1, std::min(pow2ceil(a.H), NCELL / a.TW));
    a.NC = a.TH * a.TW;                            // cells per stage (<= NCELL)
    a.lgTW = ilog2(a.TW);
    a.lgNC = ilog2(a.NC);
    a.tiles_x = (a.W + a.TW - 1) / a.TW;
    a.tiles_y = (a.H + a.TH - 1) / a.TH;
    if (KIND == TE_CONV_3X3) { a.QH = a.TH + 2; a.QW = a.TW + 2; }
    else if (KIND == TE_CONV_T2) { a.QH = 2 * a.TH + 1; a.QW = 2 * a.TW + 1; }
    else { a.QH = a.TH; a.QW = a.TW; }
    a.QS = (a.QH * a.QW) | 1;
    a.PS = a.NC | 1;
    a.magic_qt = magic((unsigned)(a.QH * a.QW));
    a.magic_qw = magic((unsigned)a.QW);
    a.magic_q8 = magic((unsigned)(a.QH * 8));
    a.magic_q2 = magic((unsigned)(a.QH * 2));
    a.magic_q16 = magic((unsigned)(a.QH * 16));
    a.magic_q1 = magic((unsigned)a.QH);
    // the 16-byte staging path needs full 32-cell rows whose global rows are 16 B aligned, and its one edge pass
    // (QCH * QH * 2 = 512 scalars at the full 64-cell tile) takes 512 / NTHREADS sweeps of the block
    a.vec = (a.TW == 32 && a.W % 32 == 0 && a.NC == WK<KIND>::NCELL && a.H % a.TH == 0 && QCH * a.QH * 2 <= 512 &&
             (((uintptr_t)a.g | (uintptr_t)a.x) & 15) == 0) ? 1 : 0;
    return QCH * a.QH * a.QW <= WK<KIND>::NQ8 * 512 && a.NC <= WK<KIND>::NCELL;
}

inline int pick_nwp(int Co, int Ci) { return (Co <= 64 && Ci <= 64) ? 2 : 4; }

// The pair form (WINO) of the 3x3 kernel: 8-wave tile only (twelve accumulators plus the 4-wave tile's 34 staging registers do
// not fit 256 VGPRs), stages of whole pair couples (a k-step = 4 cells).  Measured on the same box at the FFHQ-256 batch-16
// shapes (profiles/experiments/r04_wgrad_pair_ab.log): 125.6 -> 153.4, 124.4 -> 158.4, 133.0 -> 171.4 TFLOP/s algorithmic,
// same deviation from an fp64 reference (1.0e-6 vs 1.1e-6 relative).  TE_WGRAD_DIRECT=1 (read once) keeps the direct form, for A/B runs.
inline bool pair_form_3x3(int Co, int Ci, int NC) {
    static const bool direct = getenv("TE_WGRAD_DIRECT") && atoi(getenv("TE_WGRAD_DIRECT"));
    return !direct && pick_nwp(Co, Ci) == 4 && NC % 4 == 0;
}

template <int KIND, int NWP, bool WINO>
void launch_wgrad_t(const WgArgs& a, hipStream_t s) {
    constexpr int PCH = NWP * 32;
    size_t lds = ((KIND == TE_CONV_1X1 || NWP == 2) ? 1 : 2) * sizeof(float) * ((size_t)PCH * a.PS + (size_t)QCH * a.QS);   // 8-wave 3x3 / T2: two operand images
    if (NWP == 2) lds = std::max(lds, sizeof(float) * 2 * (WINO ? 12 : WK<KIND>::NT) * 16 * 64);     // K-split partial tiles (<= 2 groups)
    static std::atomic<uint64_t> attr_done{0};
    te::allow_big_lds(attr_done, (const void*)wgrad_mfma_kernel<KIND, NWP, WINO>, 160 * 1024);
    constexpr bool GSHIFT = (KIND == TE_CONV_T2);
    dim3 grid((unsigned)(a.B / a.NB * a.S), (unsigned)te::cdiv(a.Co, GSHIFT ? QCH : PCH), (unsigned)te::cdiv(a.Ci, GSHIFT ? PCH : QCH));
    wgrad_mfma_kernel<KIND, NWP, WINO><<<grid, NWP * 128, lds, s>>>(a);
}

template <int KIND>
int launch_wgrad(WgArgs a, hipStream_t s) {
    if (!fill_geometry<KIND>(a)) return te::fail(TE_ERR_UNSUPPORTED, "te_wgrad_f32: unsupported image size %dx%d", a.H, a.W);
    if ((int64_t)a.NB * a.Co * a.Hg * a.Wg * 4 >= (int64_t)OOB || (int64_t)a.NB * a.Ci * a.Hx * a.Wx * 4 >= (int64_t)OOB)
        return te::fail(TE_ERR_UNSUPPORTED, "te_wgrad_f32: a sample group exceeds 2 GiB");
    if constexpr (KIND == TE_CONV_3X3) {
        if (pair_form_3x3(a.Co, a.Ci, a.NC)) { launch_wgrad_t<KIND, 4, true>(a, s); return 0; }
    }
    if (pick_nwp(a.Co, a.Ci) == 2) launch_wgrad_t<KIND, 2, false>(a, s);
    else launch_wgrad_t<KIND, 4, false>(a, s);
    return 0;
}

inline int n_cell_tiles(int kind, int H, int W) {
    const int ncell = (kind == TE_CONV_T2) ? 32 : 64;
    const int TW = std::max(2, std::min(32, pow2ceil(W)));
    const int TH = std::max(1, std::min(pow2ceil(H), ncell / TW));
    return ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
}

// ------------------------------------------------------------------------------------------------ reduce
// Two streaming kernels over the slabs (both fully parallel, loads independent of each other):
//   reduce_w : one thread per (co, ci, tap) element, loops over a chunk of (b, s)           -> dW
//   reduce_sc: one block per (b, 32 output channels), threads over ci                        -> d isc, d osc
constexpr int RTHREADS = 256;
constexpr int COB = 32;    // output channels per block in reduce_sc

__global__ __launch_bounds__(RTHREADS) void wgrad_reduce_w_kernel(float* __restrict__ gw, const float* __restrict__ slabs,
                                                                  float wscale, const float* __restrict__ isc,
                                                                  const float* __restrict__ osc, int B, int S, int Co, int Ci,
                                                                  int NT, int nchunk) {
    const int64_t E = (int64_t)Co * Ci * NT;
    const int64_t e = (int64_t)blockIdx.x * RTHREADS + threadIdx.x;
    if (e >= E) return;
    const int ci = (int)((e / NT) % Ci), co = (int)(e / ((int64_t)NT * Ci));
    const int BS = B * S;
    const int j0 = (int)((int64_t)BS * blockIdx.y / nchunk), j1 = (int)((int64_t)BS * (blockIdx.y + 1) / nchunk);
    float acc = 0.f;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        const int b = j / S;
        float v = slabs[(size_t)j * E + e];
        if (osc) v *= osc[(size_t)b * Co + co];
        if (isc) v *= isc[(size_t)b * Ci + ci];
        acc += v;
    }
    // nchunk > 1: `gw` is the workspace [nchunk][E]; sum_parts_kernel adds the chunks in a fixed order
    gw[(size_t)blockIdx.y * E + e] = acc * wscale;
}

// 16 bytes per lane: requires E % 4 == 0 and 16-byte aligned slabs
__global__ __launch_bounds__(RTHREADS) void wgrad_reduce_w4_kernel(float* __restrict__ gw, const float* __restrict__ slabs,
                                                                   float wscale, const float* __restrict__ isc,
                                                                   const float* __restrict__ osc, int B, int S, int Co, int Ci,
                                                                   int NT, int nchunk) {
    const int64_t E = (int64_t)Co * Ci * NT;
    const int64_t e = ((int64_t)blockIdx.x * RTHREADS + threadIdx.x) * 4;
    if (e >= E) return;
    int ci[4], co[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ci[j] = (int)(((e + j) / NT) % Ci); co[j] = (int)((e + j) / ((int64_t)NT * Ci)); }
    const int BS = B * S;
    const int j0 = (int)((int64_t)BS * blockIdx.y / nchunk), j1 = (int)((int64_t)BS * (blockIdx.y + 1) / nchunk);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        const int b = j / S;
        float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)j * E + e);
        float sc[4] = {1.f, 1.f, 1.f, 1.f};
        if (osc) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sc[q] *= osc[(size_t)b * Co + co[q]];
        }
        if (isc) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sc[q] *= isc[(size_t)b * Ci + ci[q]];
        }
        acc.x += v.x * sc[0]; acc.y += v.y * sc[1]; acc.z += v.z * sc[2]; acc.w += v.w * sc[3];
    }
    *reinterpret_cast<float4*>(gw + (size_t)blockIdx.y * E + e) =
        make_float4(acc.x * wscale, acc.y * wscale, acc.z * wscale, acc.w * wscale);
}

template <int NT>
__global__ __launch_bounds__(RTHREADS) void wgrad_reduce_sc_kernel(float* __restrict__ gisc, float* __restrict__ gosc,
                                                                   const float* __restrict__ slabs, const float* __restrict__ w,
                                                                   float wscale, const float* __restrict__ isc,
                                                                   const float* __restrict__ osc, int B, int S, int Co, int Ci) {
    __shared__ float red[4][COB];
    const int b = blockIdx.x, cob = blockIdx.y * COB;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const size_t slab_sz = (size_t)Co * Ci * NT;
    const float* sb = slabs + (size_t)b * S * slab_sz;
    float osum = 0.f;                                      // thread j < COB: d osc of channel cob + j, summed over the passes
    // d isc: this block's share goes to its own row of `gisc` ([gridDim.y][B][Ci] when there are several channel blocks: the
    // fixed-order second pass adds the rows; the final tensor itself when there is one)
    float* gi_row = gisc ? gisc + (size_t)blockIdx.y * B * Ci : nullptr;
    for (int c0 = 0; c0 < Ci; c0 += RTHREADS) {            // usually one pass (Ci <= 256) or two (512)
        const int ci = c0 + threadIdx.x;
        const bool live = ci < Ci;
        const int cic = live ? ci : Ci - 1;
        const float is = isc ? isc[(size_t)b * Ci + cic] : 1.f;
        float pisc = 0.f;
        float posc[COB];
#pragma unroll
        for (int j = 0; j < COB; ++j) {
            const int co = cob + j;
            float dot = 0.f;
            if (co < Co) {
                const size_t off = ((size_t)co * Ci + cic) * NT;
                float sl[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) sl[t] = 0.f;
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) sl[t] += sb[(size_t)s * slab_sz + off + t];
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) dot += w[off + t] * sl[t];
                dot *= wscale;
                pisc += (osc ? osc[(size_t)b * Co + co] : 1.f) * dot;
            }
            posc[j] = live ? is * dot : 0.f;
        }
        if (gi_row && live) gi_row[(size_t)b * Ci + ci] = pisc;
        if (gosc) {
#pragma unroll
            for (int j = 0; j < COB; ++j) {
                float v = posc[j];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
                if (lane == 0) red[wid][j] = v;
            }
            __syncthreads();
            if (threadIdx.x < COB)
                osum += (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            __syncthreads();
        }
    }
    if (gosc && threadIdx.x < COB && cob + threadIdx.x < Co) gosc[(size_t)b * Co + cob + threadIdx.x] = osum;      // the block owns these
}

// taps == 1 with <= 4 output channels (ToRGB): block (ci chunk, b); a thread owns one input channel of one sample, sums
// its slabs, writes its gisc[b,ci] (it sees every output channel) and its share of gw as parts for the second pass.
__global__ __launch_bounds__(RTHREADS) void wgrad_reduce_few_kernel(float* __restrict__ gw, float* __restrict__ gisc,
                                                                    const float* __restrict__ slabs, const float* __restrict__ w,
                                                                    float wscale, const float* __restrict__ isc, int B, int S,
                                                                    int Co, int Ci) {
    const int ci = blockIdx.x * RTHREADS + threadIdx.x, b = blockIdx.y;
    if (ci >= Ci) return;
    const size_t E = (size_t)Co * Ci;
    const float* sp = slabs + (size_t)b * S * E + ci;
    const int s0 = (int)((int64_t)S * blockIdx.z / gridDim.z), s1 = (int)((int64_t)S * (blockIdx.z + 1) / gridDim.z);
    float u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = s0; s < s1; ++s) {
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < Co) u[o] += sp[(size_t)s * E + (size_t)o * Ci];
    }
    const float is = isc ? isc[(size_t)b * Ci + ci] : 1.f;
    float gi = 0.f;
    // gw: every (chunk, sample) block writes its own part [gridDim.z * B][Co * Ci]; gisc: [gridDim.z][B * Ci] (the final tensor
    // when there is one chunk).  The fixed-order second pass adds the parts.
    float* gwp = gw ? gw + ((size_t)blockIdx.z * B + b) * E : nullptr;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        if (o < Co) {
            gi += u[o] * w[(size_t)o * Ci + ci];
            if (gwp) gwp[(size_t)o * Ci + ci] = u[o] * is * wscale;
        }
    }
    if (gisc) gisc[((size_t)blockIdx.z * B + b) * Ci + ci] = gi * wscale;
}

// One pass over the slabs for all three gradients (3x3 kinds, Co % 16 == 0, Ci % 32 == 0).  A block of 2 waves owns a
// 16 co x 32 ci tile (x 9 taps); thread (row r = tid / 8, l8 = tid % 8) owns the 36 consecutive floats of 4 input
// channels x 9 taps of row r, so the tap sums stay in registers, the reduction over ci (-> gosc) is 3 lane exchanges
// inside an 8-lane group and the reduction over co (-> gisc) 3 exchanges across the wave's 8 rows plus one LDS hop
// between the two waves.  gw accumulates in registers over the samples (plain stores, no atomics); two samples are
// in flight per iteration to hide the load latency of the short per-thread streams.
constexpr int FROWS = 16, FTHREADS = FROWS * 8;

struct FusedCtx {
    float* gisc; float* gosc; const float* isc; const float* osc;
    int Co, Ci, co, cib, ci0, l8, lane, wid, tid;
};

__device__ __forceinline__ void fused_consume(const FusedCtx& c, int b, const f32x4 (&u)[9], const f32x4 (&wv)[9],
                                              f32x4 (&acc)[9], float* red) {
    float is[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) is[k] = c.isc ? c.isc[(size_t)b * c.Ci + c.ci0 + k] : 1.f;
    const float os = c.osc ? c.osc[(size_t)b * c.Co + c.co] : 1.f;
    float pi[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 36; ++j) {
        const int k = j / 9;
        const float uv = u[j >> 2][j & 3];
        pi[k] += uv * wv[j >> 2][j & 3];
        acc[j >> 2][j & 3] += uv * (is[k] * os);
    }
    if (c.gosc) {
        float po = (pi[0] * is[0] + pi[1] * is[1]) + (pi[2] * is[2] + pi[3] * is[3]);
        po += __shfl_xor(po, 1, 64);
        po += __shfl_xor(po, 2, 64);
        po += __shfl_xor(po, 4, 64);
        if (c.l8 == 0) c.gosc[(size_t)b * c.Co + c.co] = po;             // this block's part (c.gosc points at its row)
    }
    if (c.gisc) {
        float* rb = red + (b & 1) * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = pi[k] * os;
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (c.lane < 8) rb[c.wid * 32 + c.l8 * 4 + k] = v;
        }
        __syncthreads();                   // one barrier per sample: the buffer alternates with b
        if (c.tid < 32) c.gisc[(size_t)b * c.Ci + c.cib + c.tid] = rb[c.tid] + rb[32 + c.tid];      // this block's part
    }
}

__global__ __launch_bounds__(FTHREADS) void wgrad_reduce_fused_kernel(float* __restrict__ gw, float* __restrict__ gisc,
                                                                      float* __restrict__ gosc, const float* __restrict__ slabs,
                                                                      const float* __restrict__ w, float wscale,
                                                                      const float* __restrict__ isc, const float* __restrict__ osc,
                                                                      int B, int S, int Co, int Ci, int nzs) {
    // blockIdx.z = zb * nzs + zs splits the slab chunks of every sample (zs: narrow layers - a 32 x 32 weight has 2 tiles but
    // hundreds of chunks) and, round 6, the BATCH (zb: the 128- / 256-channel layers have 32 / 128 tiles of two waves and one or
    // two chunks per sample - a quarter of the CUs got a block and the pass ran at 0.31 of the HBM peak); all three outputs are
    // linear in the slab sums.  NO atomics: d osc of a (sample, channel) gets a share from every ci block and chunk range, d isc
    // from every co block and chunk range, dW (when split) from every z - each block writes its share to its own row of the
    // workspace (gosc: [nzs * gridDim.x][B][Co], gisc: [nzs * gridDim.y][B][Ci] - a sample belongs to ONE zb -, gw: [gridDim.z][E])
    // and sum_parts_kernel adds the rows in a fixed order: bit-reproducible gradients.
    const int zs = blockIdx.z % nzs, zb = blockIdx.z / nzs, nzb = gridDim.z / nzs;
    const int s0 = (int)((int64_t)S * zs / nzs), s1 = (int)((int64_t)S * (zs + 1) / nzs);
    const int b0 = (int)((int64_t)B * zb / nzb), b1 = (int)((int64_t)B * (zb + 1) / nzb);
    __shared__ float red[2 * 64];
    FusedCtx c;
    c.gisc = gisc ? gisc + ((size_t)zs * gridDim.y + blockIdx.y) * B * Ci : nullptr;
    c.gosc = gosc ? gosc + ((size_t)zs * gridDim.x + blockIdx.x) * B * Co : nullptr;
    c.isc = isc; c.osc = osc; c.Co = Co; c.Ci = Ci;
    c.tid = threadIdx.x; c.l8 = c.tid & 7; c.lane = c.tid & 63; c.wid = c.tid >> 6;
    c.co = blockIdx.y * FROWS + (c.tid >> 3); c.cib = blockIdx.x * 32; c.ci0 = c.cib + c.l8 * 4;
    const size_t E = (size_t)Co * Ci * 9;
    const size_t off = ((size_t)c.co * Ci + c.ci0) * 9;
    f32x4 wv[9], acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        wv[i] = *reinterpret_cast<const f32x4*>(w + off + 4 * i) * wscale;
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int b = b0;
    for (; b + 1 < b1; b += 2) {
        f32x4 u0[9], u1[9];
        const float* p0 = slabs + ((size_t)b * S + s0) * E + off;
        const float* p1 = p0 + (size_t)S * E;
#pragma unroll
        for (int i = 0; i < 9; ++i) u0[i] = *reinterpret_cast<const f32x4*>(p0 + 4 * i);
#pragma unroll
        for (int i = 0; i < 9; ++i) u1[i] = *reinterpret_cast<const f32x4*>(p1 + 4 * i);
        for (int s = 1; s < s1 - s0; ++s) {
            const float* q0 = p0 + (size_t)s * E;
            const float* q1 = p1 + (size_t)s * E;
#pragma unroll
            for (int i = 0; i < 9; ++i) u0[i] += *reinterpret_cast<const f32x4*>(q0 + 4 * i);
#pragma unroll
            for (int i = 0; i < 9; ++i) u1[i] += *reinterpret_cast<const f32x4*>(q1 + 4 * i);
        }
        fused_consume(c, b, u0, wv, acc, red);
        fused_consume(c, b + 1, u1, wv, acc, red);
    }
    if (b < b1) {
        f32x4 u0[9];
        const float* p0 = slabs + ((size_t)b * S + s0) * E + off;
#pragma unroll
        for (int i = 0; i < 9; ++i) u0[i] = *reinterpret_cast<const f32x4*>(p0 + 4 * i);
        for (int s = 1; s < s1 - s0; ++s) {
            const float* q0 = p0 + (size_t)s * E;
#pragma unroll
            for (int i = 0; i < 9; ++i) u0[i] += *reinterpret_cast<const f32x4*>(q0 + 4 * i);
        }
        fused_consume(c, b, u0, wv, acc, red);
    }
    if (gw) {
        float* gwz = gw + (size_t)blockIdx.z * E;           // (one chunk: the final tensor itself)
#pragma unroll
        for (int i = 0; i < 9; ++i) *reinterpret_cast<f32x4*>(gwz + off + 4 * i) = acc[i] * wscale;
    }
}

// Fixed-order second pass of the reducers: out[i] = sum_p parts[p][i], p ascending; up to three outputs per launch.
struct SumJob { float* out; const float* parts; int nparts; int64_t n; };
struct SumJobs { SumJob j[3]; };

__global__ __launch_bounds__(256) void sum_parts_kernel(SumJobs jobs) {
    const SumJob J = jobs.j[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < J.n; i += (int64_t)gridDim.x * 256) {
        // (fixed order p ascending; eight loads in flight per lane: with one, a 128-part sum of a [B, C]-sized output - the narrow layers of
        //  FFHQ-1024 - was a chain of 128 L2 round trips, 60+ us)
        float a = 0.f;
        int p = 0;
        for (; p + 8 <= J.nparts; p += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = J.parts[(size_t)(p + q) * J.n + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) a += v[q];
        }
        for (; p < J.nparts; ++p) a += J.parts[(size_t)p * J.n + i];
        J.out[i] = a;
    }
}

inline void launch_sum_parts(const SumJob* jobs, int n, hipStream_t s) {
    if (!n) return;
    SumJobs t{};
    int64_t mx = 0;
    for (int i = 0; i < n; ++i) { t.j[i] = jobs[i]; mx = std::max(mx, jobs[i].n); }
    sum_parts_kernel<<<dim3((unsigned)std::min<int64_t>(te::cdiv(mx, 256), 4 * te::kNumCU), (unsigned)n), 256, 0, s>>>(t);
}

// chunk counts of the reducer paths (shared by te_wgrad_reduce_ws_floats and the launch)
inline bool reduce_fused_ok(int Co, int Ci, int taps) { return taps == 9 && Co % FROWS == 0 && Ci % 32 == 0 && Co / FROWS <= 65535; }
// (chunk splits nzs, batch splits nzb) of the fused reducer: two blocks of two waves per CU where the problem allows, every block
// still sums >= 4 (sample, chunk) slabs so that the dW parts stay <= a quarter of the slab bytes
inline void reduce_fused_split(int B, int S, int Co, int Ci, int& nzs, int& nzb) {
    const int64_t tiles = (int64_t)(Ci / 32) * (Co / FROWS);
    int64_t nz = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(te::cdiv(2 * te::kNumCU, tiles), (int64_t)B * S / 4), 64));     // (<= 64 parts for the second pass)
    nzs = (int)std::max<int64_t>(1, std::min<int64_t>(S, nz));
    nzb = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(1, B / 2), nz / nzs));
}
inline int reduce_few_nz(int B, int S, int Ci) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(S / 4, te::cdiv(te::kNumCU, (int64_t)te::cdiv(Ci, RTHREADS) * B)));
}
inline int reduce_w_nchunk(int B, int S, int64_t E, bool vec) {
    const int64_t blocks = te::cdiv(vec ? E / 4 : E, RTHREADS);
    return (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)B * S, te::cdiv(2 * te::kNumCU, blocks)));
}

}  // namespace

#ifdef TE_CONV_PROF
extern "C" int te_debug_wgrad_prof(void* host_dst, int64_t bytes) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(te_wgrad_prof_buf), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int te_wgrad_split_bf16(int on);
extern "C" int te_wgrad_split_supported(int kind, int Co, int Ci, int H, int W);

extern "C" int te_wgrad_pair_form(int kind, int Co, int Ci, int H, int W) {
    if (kind != TE_CONV_3X3 || Co <= 0 || Ci <= 0 || H <= 0 || W <= 0) return 0;
    const int TW = std::max(2, std::min(32, pow2ceil(W))), TH = std::max(1, std::min(pow2ceil(H), 64 / TW));
    return pair_form_3x3(Co, Ci, TH * TW) ? 1 : 0;
}

extern "C" int te_wgrad_slab_count(int kind, int B, int Co, int Ci, int H, int W) {
    if (B <= 0 || Co <= 0 || Ci <= 0 || H <= 0 || W <= 0) return TE_ERR_SHAPE;
    const int tiles = n_cell_tiles(kind, H, W);
    int64_t mn = te::cdiv((int64_t)Co * Ci, (pick_nwp(Co, Ci) * 32) * QCH) * (int64_t)B;
    // the sample-pair form of csrc/wgrad6.hip (Co == Ci == 32) runs one block per PAIR of samples and chunk
    if (te_wgrad_split_bf16(-1) == 1 && te_wgrad_split_supported(kind, Co, Ci, H, W) == 2 && B % 2 == 0) mn = B / 2;
    // the split 1x1 kernel (wgrad6p_kernel, round 6) has 128 x 128 channel blocks
    if (kind == TE_CONV_1X1 && te_wgrad_split_bf16(-1) == 1 && te_wgrad_split_supported(kind, Co, Ci, H, W) == 1) mn = (int64_t)(Co / 128) * (Ci / 128) * B;
    // blocks per CU the split aims at: the 8-wave 3x3 / T2 blocks own a CU (two 68 KB operand images), so ONE round of them
    // does the same work as two with half the slab bytes for the reducer (same-box A/B, round 3: +0.8 % / +1.2 % on the
    // kernels, half the te_wgrad_reduce traffic of the narrow layers); the light 1x1 blocks share a CU and want two
#ifndef TE_WGRAD_ROUNDS
#define TE_WGRAD_ROUNDS ((kind == TE_CONV_1X1) ? 2 : 1)
#endif
    int64_t S = te::cdiv((int64_t)TE_WGRAD_ROUNDS * te::kNumCU, mn);
    S = std::max<int64_t>(1, std::min<int64_t>(S, tiles));
    return (int)S;
}

// Plan of the GROUPED form (te_wgrad_group_f32): *NB samples share a slab, *S chunks per group.  Grouping pays where the slabs
// are the traffic - small images under big weights (512 x 512 x 9 floats = 9.4 MB per slab: 302 MB written and read back for a
// 4x4 ... 32x32 layer at batch 32 against 0.1 - 34 MB of activations) - and is only valid for the PLAIN gradient (no per-sample
// modulation, no style / demodulation gradients: the discriminator's layers and the closed trio of op/modconv.py).
// NB = the largest divisor of B that still leaves every CU a block; 1 when a sample alone already splits into chunks.
extern "C" int te_wgrad_group_plan(int kind, int B, int Co, int Ci, int H, int W, int* NB_out, int* S_out) {
    if (B <= 0 || Co <= 0 || Ci <= 0 || H <= 0 || W <= 0 || !NB_out || !S_out) return TE_ERR_SHAPE;
    const int tiles = n_cell_tiles(kind, H, W);
    int64_t mn1 = te::cdiv((int64_t)Co * Ci, (pick_nwp(Co, Ci) * 32) * QCH);
    if (kind == TE_CONV_1X1 && te_wgrad_split_bf16(-1) == 1 && te_wgrad_split_supported(kind, Co, Ci, H, W) == 1) mn1 = (int64_t)(Co / 128) * (Ci / 128);
    const int64_t per_sample = std::max<int64_t>((int64_t)Co * (kind == TE_CONV_T2 ? (2 * H + 1) * (2 * W + 1) : H * W), (int64_t)Ci * H * W) * 4;
    int NB = 1;
    // whole channel blocks only: the 16-byte staging path masks a channel tail through the END of the sample's buffer range, and
    // a group's range continues into the next sample
    if (te_wgrad_slab_count(kind, B, Co, Ci, H, W) == 1 && Co % 128 == 0 && Ci % 128 == 0) {
        for (int d = 2; d <= B; ++d)
            if (B % d == 0 && (B / d) * mn1 >= te::kNumCU && (int64_t)d * per_sample < (int64_t)OOB) NB = d;
    }
    const int64_t mn = mn1 * (B / NB);
    int64_t S = te::cdiv((int64_t)((kind == TE_CONV_1X1) ? 2 : 1) * te::kNumCU, mn);
    S = std::max<int64_t>(1, std::min<int64_t>(S, (int64_t)tiles * NB));
    *NB_out = NB; *S_out = (int)S;
    return 0;
}

static int wgrad_launch(float* slabs, const float* g, const float* x, int kind, int B, int Co, int Ci, int H, int W, int S, int NB,
                        te_stream_t stream_);

extern "C" int te_wgrad_group_f32(float* slabs, const float* g, const float* x, int kind, int B, int Co, int Ci, int H, int W,
                                  int S, int NB, te_stream_t stream_) {
    TE_REQUIRE(NB > 0 && B > 0 && B % NB == 0, TE_ERR_SHAPE, "te_wgrad_group_f32: the group size must divide the batch");
    return wgrad_launch(slabs, g, x, kind, B, Co, Ci, H, W, S, NB, stream_);
}

extern "C" int te_wgrad_f32(float* slabs, const float* g, const float* x, int kind, int B, int Co, int Ci, int H, int W,
                            int S, te_stream_t stream_) {
    return wgrad_launch(slabs, g, x, kind, B, Co, Ci, H, W, S, 1, stream_);
}

// csrc/wgrad6.hip: the 3x3 correlation on the bf16 matrix pipe (three-piece split, fp32-equivalent); 1 = launched, 0 = not applicable
int te_wgrad6_launch(float* slabs, const float* g, const float* x, int kind, int B, int Co, int Ci, int H, int W, int S, int NB, hipStream_t s);

static int wgrad_launch(float* slabs, const float* g, const float* x, int kind, int B, int Co, int Ci, int H, int W, int S, int NB,
                        te_stream_t stream_) {
    TE_REQUIRE(slabs && g && x, TE_ERR_NULL, "te_wgrad_f32: NULL pointer");
    TE_REQUIRE(B > 0 && Co > 0 && Ci > 0 && H > 0 && W > 0 && S > 0, TE_ERR_SHAPE, "te_wgrad_f32: bad dims");
    TE_REQUIRE((int64_t)B * S <= 0x7FFFFFFF && te::cdiv(Co, QCH) <= 65535 && te::cdiv(Ci, QCH) <= 65535, TE_ERR_SHAPE,
               "te_wgrad_f32: grid too large");
    if (te_wgrad6_launch(slabs, g, x, kind, B, Co, Ci, H, W, S, NB, (hipStream_t)stream_) == 1) return te::launch_status("te_wgrad_f32");
    WgArgs a{};
    a.slabs = slabs; a.g = g; a.x = x; a.B = B; a.Co = Co; a.Ci = Ci; a.H = H; a.W = W; a.S = S; a.NB = NB;
    a.Hx = H; a.Wx = W;
    hipStream_t s = (hipStream_t)stream_;
    int rc;
    switch (kind) {
        case TE_CONV_3X3: a.Hg = H; a.Wg = W; rc = launch_wgrad<TE_CONV_3X3>(a, s); break;
        case TE_CONV_1X1: a.Hg = H; a.Wg = W; rc = launch_wgrad<TE_CONV_1X1>(a, s); break;
        case TE_CONV_T2: a.Hg = 2 * H + 1; a.Wg = 2 * W + 1; rc = launch_wgrad<TE_CONV_T2>(a, s); break;
        default: return te::fail(TE_ERR_UNSUPPORTED, "te_wgrad_f32: unknown kind %d", kind);
    }
    if (rc) return rc;
    return te::launch_status("te_wgrad_f32");
}

// floats of workspace te_wgrad_reduce_f32 needs for this problem (an upper bound over the alignment-dependent paths)
extern "C" int64_t te_wgrad_reduce_ws_floats(int B, int S, int Co, int Ci, int taps, int want_w, int want_isc, int want_osc) {
    if (B <= 0 || S <= 0 || Co <= 0 || Ci <= 0 || (taps != 1 && taps != 9)) return TE_ERR_SHAPE;
    const int64_t E = (int64_t)Co * Ci * taps;
    int64_t n = 0;
    if (reduce_fused_ok(Co, Ci, taps)) {
        int nzs, nzb;
        reduce_fused_split(B, S, Co, Ci, nzs, nzb);
        if (want_osc) n += (int64_t)nzs * (Ci / 32) * B * Co;
        if (want_isc) n += (int64_t)nzs * (Co / FROWS) * B * Ci;
        if (want_w && nzs * nzb > 1) n += (int64_t)nzs * nzb * E;
    }
    int64_t m = 0;                                         // the generic path (also the fall-back of a misaligned fused problem)
    if (want_w) m += (int64_t)std::max(reduce_w_nchunk(B, S, E, false), reduce_w_nchunk(B, S, E, E % 4 == 0)) * E;
    if (want_isc) m += (int64_t)te::cdiv(Co, COB) * B * Ci;
    int64_t f = 0;                                         // ToRGB path
    if (taps == 1 && Co <= 4) {
        const int nzf = reduce_few_nz(B, S, Ci);
        if (want_w) f += (int64_t)nzf * B * E;
        if (want_isc && nzf > 1) f += (int64_t)nzf * B * Ci;
    }
    return std::max(n, std::max(m, f));
}

extern "C" int te_wgrad_reduce_f32(float* gw, float* gisc, float* gosc, float* ws, const float* slabs, const float* w, float wscale,
                                   const float* isc, const float* osc, int B, int S, int Co, int Ci, int taps,
                                   te_stream_t stream_) {
    TE_REQUIRE(slabs && w, TE_ERR_NULL, "te_wgrad_reduce_f32: NULL pointer");
    TE_REQUIRE(B > 0 && S > 0 && Co > 0 && Ci > 0 && (taps == 1 || taps == 9), TE_ERR_SHAPE, "te_wgrad_reduce_f32: bad dims");
    TE_REQUIRE(ws || te_wgrad_reduce_ws_floats(B, S, Co, Ci, taps, gw != nullptr, gisc != nullptr, gosc != nullptr) == 0, TE_ERR_NULL,
               "te_wgrad_reduce_f32: this problem needs the workspace (te_wgrad_reduce_ws_floats)");
    hipStream_t s = (hipStream_t)stream_;
    const int64_t E = (int64_t)Co * Ci * taps;
    SumJob jobs[3];
    int nj = 0;
    const uintptr_t al = reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(gw) |
                         reinterpret_cast<uintptr_t>(ws);
    if (reduce_fused_ok(Co, Ci, taps) && (al & 15) == 0) {                                          // single-pass path
        int nzs, nzb;
        reduce_fused_split(B, S, Co, Ci, nzs, nzb);
        const int nz = nzs * nzb, nX = Ci / 32, nY = Co / FROWS;
        float* p = ws;
        float* po = nullptr; float* pi = nullptr; float* pw = gw;
        if (gosc) { po = p; p += (size_t)nzs * nX * B * Co; jobs[nj++] = SumJob{gosc, po, nzs * nX, (int64_t)B * Co}; }
        if (gisc) { pi = p; p += (size_t)nzs * nY * B * Ci; jobs[nj++] = SumJob{gisc, pi, nzs * nY, (int64_t)B * Ci}; }
        if (gw && nz > 1) { pw = p; p += (size_t)nz * E; jobs[nj++] = SumJob{gw, pw, nz, E}; }
        dim3 grid((unsigned)nX, (unsigned)nY, (unsigned)nz);
        wgrad_reduce_fused_kernel<<<grid, FTHREADS, 0, s>>>(pw, pi, po, slabs, w, wscale, isc, osc, B, S, Co, Ci, nzs);
        launch_sum_parts(jobs, nj, s);
        return te::launch_status("te_wgrad_reduce_f32");
    }
    if (taps == 1 && Co <= 4 && !gosc && !osc) {                                                   // ToRGB
        // few, long per-thread streams when the image is large (S up to a few hundred): split them over blockIdx.z
        const int nzf = reduce_few_nz(B, S, Ci);
        float* p = ws;
        float* pw = nullptr; float* pi = gisc;
        if (gw) { pw = p; p += (size_t)nzf * B * E; jobs[nj++] = SumJob{gw, pw, nzf * B, E}; }
        if (gisc && nzf > 1) { pi = p; p += (size_t)nzf * B * Ci; jobs[nj++] = SumJob{gisc, pi, nzf, (int64_t)B * Ci}; }
        dim3 grid((unsigned)te::cdiv(Ci, RTHREADS), (unsigned)B, (unsigned)nzf);
        wgrad_reduce_few_kernel<<<grid, RTHREADS, 0, s>>>(pw, pi, slabs, w, wscale, isc, B, S, Co, Ci);
        launch_sum_parts(jobs, nj, s);
        return te::launch_status("te_wgrad_reduce_f32");
    }
    float* p = ws;
    if (gw) {
        const bool vec = (E % 4 == 0) && ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(gw) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0;
        const int64_t blocks = te::cdiv(vec ? E / 4 : E, RTHREADS);
        const int nchunk = reduce_w_nchunk(B, S, E, vec);
        float* pw = gw;
        if (nchunk > 1) { pw = p; p += (size_t)nchunk * E; jobs[nj++] = SumJob{gw, pw, nchunk, E}; }
        if (vec) wgrad_reduce_w4_kernel<<<dim3((unsigned)blocks, (unsigned)nchunk), RTHREADS, 0, s>>>(pw, slabs, wscale, isc, osc, B, S,
                                                                                                    Co, Ci, taps, nchunk);
        else wgrad_reduce_w_kernel<<<dim3((unsigned)blocks, (unsigned)nchunk), RTHREADS, 0, s>>>(pw, slabs, wscale, isc, osc, B, S, Co,
                                                                                               Ci, taps, nchunk);
    }
    if (gisc || gosc) {
        const int nY = (int)te::cdiv(Co, COB);
        float* pi = gisc;
        if (gisc && nY > 1) { pi = p; p += (size_t)nY * B * Ci; jobs[nj++] = SumJob{gisc, pi, nY, (int64_t)B * Ci}; }
        dim3 grid((unsigned)B, (unsigned)nY);
        if (taps == 9) wgrad_reduce_sc_kernel<9><<<grid, RTHREADS, 0, s>>>(pi, gosc, slabs, w, wscale, isc, osc, B, S, Co, Ci);
        else wgrad_reduce_sc_kernel<1><<<grid, RTHREADS, 0, s>>>(pi, gosc, slabs, w, wscale, isc, osc, B, S, Co, Ci);
    }
    launch_sum_parts(jobs, nj, s);
    return te::launch_status("te_wgrad_reduce_f32");
}
