// F1: modulated-convolution family as implicit GEMM on the gfx950 fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD = the fp32 peak of the chip).
//
//     out[b,m,:,:] = act( osc[b,m] * sum_{k,tap} Wp[tap][k][m] * (isc[b,k] * in[b,k, . (+) tap]) + bias[m] )
//
// One shared (packed) weight tensor for the whole batch; the per-sample style modulation is a scaling of
// the input channels applied while the input tile is staged into LDS, the demodulation a scaling of the
// output channels applied in the epilogue together with bias + leaky-ReLU.  (The reference materialises B
// per-sample weight copies and calls stock grouped convolutions: model_spatial_query.py:296-337.)
//
// GEMM view:  M = output channels (block tile 128),  N = output cells (block tile 32*NBW*2 cells, laid
// out as NS samples x TH rows x TW cols, all powers of two chosen at launch so 4x4 ... 1024x1024 images
// all fill the tile),  K = input channels, 8 per stage, times the taps.
// 4 waves per block as 2(M) x 2(N); each wave owns MBW x NACC accumulator tiles of 32x32.
//
// Two forms of the same kernel (template flag FAST): the general one takes any K / tile shape (channel tails, several small
// images per tile, scalar staging); FAST serves every launch with whole stages and one sample per tile, i.e. all layers that
// carry FLOPs, and is built around one measurement (tools/conv_phase_prof.py): next to a saturated matrix pipe every
// vector-ALU / address instruction of the staging code costs ~50 cycles.  So its loads carry fixed per-thread offsets and a
// scalar stage offset, the input tile comes in 16-byte row segments, and the loads of stage k+1 are dealt out over the MFMA
// steps of stage k.  The grid is 1-D with an XCD-aware block -> (tile, M block) mapping.
//
// Kinds (include/te_hip.h):
//   3X3  3x3 stride 1 pad 1                       (forward of plain layers; with flipped/transposed packing,
//                                                  their data gradient)
//   1X1  the skip convolutions of the discriminator's ResBlocks, from-RGB, ToRGB shapes the streaming rgb.hip kernels do not cover
//   S2   3x3 stride 2 over a (2H+1)x(2W+1) input  (data gradient of T2)
//   T2   3x3 transposed stride 2 -> (2H+1)x(2W+1) (forward of the upsampling layers).  Written as its four
//        output phases (row parity a, col parity b): phase (a,b) of cell (i,j) is output (2i+a, 2j+b) and
//        receives only the taps with ky = a (mod 2), kx = b (mod 2) -> 4+2+2+1 = 9 tap-GEMMs per cell,
//        i.e. exactly the FLOPs of the zero-insertion-free transposed convolution.
#include "conv_common.h"

namespace {

// Tile configuration per (kind, tile class TC).  4 waves as WM (along M) x 4/WM (along the cells); each wave owns MBW
// 32-row M blocks and NBW 32-cell blocks; KC input channels per stage; NSP staging sweeps for the input tile.
//   TC 0: M >= 96 -> block 128 (M) x 128 cells      TC 1: M ~ 64 -> 64 x 256 cells      TC 2: M <= 32 -> 32 x 256 cells
// (narrow layers of the 512 / 1024 px generators, ToRGB fallbacks) so the MFMA rows are not padded 2-4x with zeros.
#ifndef T2KC0
#define T2KC0 16
#endif
// build knobs for A/B measurements (tools/exp_build.py); the defaults are the product
#ifndef TE_CONV_FAST         // 0: never dispatch to the FAST kernels
#define TE_CONV_FAST 1
#endif
#ifndef TE_CONV_XCD          // XCD-aware block -> (tile, M block) mapping (0: M block major, as the plain 2-D grid did)
#define TE_CONV_XCD 1
#endif
#ifndef TE_CONV_FAST_NARROW  // FAST kernels for the 64- and 32-row tile classes too (narrow layers of the 512 / 1024 px models)
#define TE_CONV_FAST_NARROW 1
#endif
#ifndef TE_CONV_DEEP1X1      // 0: never pick the deep-stage 1x1 class
#define TE_CONV_DEEP1X1 1
#endif
#ifndef TE_FAST_PF           // LDS operands of step s+1 are read before the MFMAs of step s
#define TE_FAST_PF 1
#endif
#ifndef TE_FAST_IL           // the global loads of the next stage are spread over the MFMA steps of this one
#define TE_FAST_IL 1
#endif
#if defined(TE_CONV_PROF) || defined(TE_CONV_PROF2)
// experimental builds: per-phase cycle counts of wave 0 of every block (s_memtime), read back with te_debug_conv_prof
__device__ unsigned long long te_conv_prof_buf[8192 * 8];
#endif
#ifdef TE_CONV_PROF
#define PROF_T(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define PROF_T(v)
#endif
#ifdef TE_CONV_PROF2         // whole-block timeline of ANY kernel form (100 MHz s_memrealtime): entry, K loop start / end, exit
#define PROF2_T(v) const unsigned long long v = __builtin_amdgcn_s_memrealtime()
#else
#define PROF2_T(v)
#endif
template <int KIND, int TC> struct Cfg;
template <int KIND> struct Cfg<KIND, 0> { static constexpr int WM = 2, MBW = 2, NBW = 2, KC = 8, NSP = (KIND == TE_CONV_S2) ? 3 : (KIND == TE_CONV_3X3 ? 2 : 1), NQ = (KIND == TE_CONV_S2) ? 5 : 2; };
template <int KIND> struct Cfg<KIND, 1> { static constexpr int WM = 2, MBW = 1, NBW = (KIND == TE_CONV_S2) ? 2 : 4, KC = 8, NSP = (KIND == TE_CONV_S2) ? 3 : (KIND == TE_CONV_3X3 ? 2 : 1), NQ = (KIND == TE_CONV_S2) ? 5 : 4; };
template <int KIND> struct Cfg<KIND, 2> { static constexpr int WM = 1, MBW = 1, NBW = 2, KC = 8, NSP = (KIND == TE_CONV_S2) ? 5 : (KIND == TE_CONV_3X3 ? 2 : 1), NQ = (KIND == TE_CONV_S2) ? 10 : 4; };
// transposed conv: 4 phase accumulators per cell block -> 64 x 128 cells per block and 16 channels per stage keep the
// MFMA work per staged weight byte equal to the plain 3x3 kernel; 32 x 128 cells for narrow outputs
template <> struct Cfg<TE_CONV_T2, 0> { static constexpr int WM = 2, MBW = 1, NBW = 2, KC = T2KC0, NSP = 1, NQ = 3; };
template <> struct Cfg<TE_CONV_T2, 1> { static constexpr int WM = 2, MBW = 1, NBW = 1, KC = 16, NSP = 1, NQ = 2; };
template <> struct Cfg<TE_CONV_T2, 2> { static constexpr int WM = 1, MBW = 1, NBW = 1, KC = 16, NSP = 1, NQ = 3; };
// 1x1 with deep stages (FAST only: 64 channels per stage = 128 MFMAs per wave and stage instead of 16; the skip convolutions
// of the discriminator's ResBlocks and their data gradients)
template <> struct Cfg<TE_CONV_1X1, 3> { static constexpr int WM = 2, MBW = 2, NBW = 2, KC = 64, NSP = 1, NQ = 8; };
template <int KIND, int TC> constexpr int tile_bm() { return Cfg<KIND, TC>::WM * Cfg<KIND, TC>::MBW * 32; }
template <int KIND, int TC> constexpr int tile_cells() { return (4 / Cfg<KIND, TC>::WM) * Cfg<KIND, TC>::NBW * 32; }

struct ConvArgs {
    float* out;
    float* ws;               // split-K partial sums [ksplit][B][M][Ho][Wo] (NULL: the launch does not split - te_conv_f32 - same result up to
                             // summation order, slower on 4x4 ... 16x16 images; there are no atomics in this library)
    const float* in;
    const float* wp;
    const float* isc;
    const float* osc;
    const float* bias;
    const float* res;        // residual [B][M][Ho][Wo] added AFTER the activation (the sum of a ResBlock / of two gradient branches), or NULL
    const float* mref;       // leaky-ReLU gradient mask applied LAST: out *= (mref > 0 ? mgain : 0.2 * mgain), mref shaped like out
    float mgain;             // (a data gradient that lands on the output of a fused bias + leaky-ReLU: its backward in this epilogue)
    int B, K, M, Kp, Mp;
    int Hi, Wi, Ho, Wo;      // input / output spatial size
    int H, W;                // low-resolution size (cells of T2 live on (H+1)x(W+1))
    int act;
    // up to 3 cell regions share one launch (T2: main body + last output column + last output row, so the thin
    // edge regions run concurrently with the body instead of as two nearly empty launches)
    struct Region {
        int ri0, rj0, rh, rw;    // cell region
        int TH, TW, NS, lgTW, lgTH;
        int NSv;                 // samples actually staged per tile (<= NS; smaller when the input tiles of NS samples
                                 // would not fit the staging budget: the remaining cells of the tile stay unused)
        int tiles_x, tiles_y;    // tiles per sample group
        int TIH, TIW, TIWP, SS, CS;  // input tile: rows, cols, row stride, per-sample stride, per-channel stride (floats)
        int nseg, xo, OD;        // FAST staging: 16-byte segments per tile row, LDS column of tile column 0; S2: LDS column of
                                 // the first odd input column (even columns first)
        int first_block;         // blockIdx.x of the region's first tile
        int tapmask;             // taps that can contribute in this region (thin T2 edge regions need 3 of 9)
    } reg[3];
    int nreg;
    int ntiles, mblocks;     // grid: ceil(ntiles / 8) * 8 * mblocks blocks along x (see the kernel's block -> work mapping)
    int nt8, nbody;          // banded order: tiles [0, nbody) (the first region) go to the XCDs in 8 bands of nt8 tiles; the thin edge
                             // regions of T2 after them stay interleaved (every XCD gets its share of the cheap tiles); nt8 = 0: all interleaved
    int ksplit, kchunk;      // split of the input-channel loop over blockIdx.z (small images: too few tiles to fill the chip)
};

template <int KIND> struct Kind;
template <> struct Kind<TE_CONV_3X3> { static constexpr int NT = 9; };
template <> struct Kind<TE_CONV_T2>  { static constexpr int NT = 9; };
template <> struct Kind<TE_CONV_S2>  { static constexpr int NT = 9; };
template <> struct Kind<TE_CONV_1X1> { static constexpr int NT = 1; };

// MBW: 32-row M blocks per wave (block M tile = 2*MBW*32 = 128 -> MBW = 2)
// NBW: 32-cell N blocks per wave.  T2 keeps 4 phase accumulators per cell block.
// MS: the cell tile spans several samples (small images) -> style scales are fetched per staged element
// FAST: every stage is a full one (K % KC == 0, host-checked): stage loads take scalar channel offsets and are spread over
// the MFMA steps, the LDS operands are read one step ahead, input tiles are staged as 16-byte row segments, and the four
// shifted B fragments of T2 are shared by its taps.
// EPI: the residual / activation-gradient-mask epilogue stages are compiled in (unmodulated 3x3 / 1x1 launches that ask for
// them: the discriminator's ResBlock node).  A template flag, not a run-time branch: the 16 + 16 extra loads per accumulator
// tile push the 168-register (3 waves / SIMD) kernels into scratch spills when they are part of every instantiation.
template <int KIND, int TC, bool HAS_ISC, bool MS, int OCC, bool FAST, bool EPI = false>
__global__ __launch_bounds__(NTHREADS, OCC) void conv_mfma_kernel(const ConvArgs p) {
    using C = Cfg<KIND, TC>;
    constexpr int NBW = C::NBW, MBW = C::MBW, KC = C::KC, WM = C::WM, WN = 4 / C::WM, NSP = C::NSP;
    constexpr int BM = WM * MBW * 32;             // block tile, output channels
    constexpr int NTAP = Kind<KIND>::NT;
    constexpr bool IS_T2 = (KIND == TE_CONV_T2);
    constexpr int NACC = IS_T2 ? 4 * NBW : NBW;
    constexpr int NTILE = WN * NBW * 32;          // cells per block tile
    constexpr int WSTAGE = NTAP * KC * BM;        // floats of packed weights per stage
    constexpr int WLDR = (WSTAGE / 4 + NTHREADS - 1) / NTHREADS;   // 16-byte weight loads per thread per stage
    constexpr bool WEVEN = (WSTAGE / 4) % NTHREADS == 0;
    constexpr int NQ = Cfg<KIND, TC>::NQ;        // FAST: (row segment, channel) pairs per thread and stage
    constexpr bool TAP_PIECES = FAST && IS_T2 && (KC * BM / 4 == NTHREADS);   // weight piece r of a thread == tap r
    // the same fact for every kind: with KC * BM / 4 == NTHREADS piece r of a thread is tap r of ONE (channel, column group),
    // so its byte offsets differ by the constant r * Kp * Mp * 4 - that goes into the load's SCALAR offset and the thread
    // keeps one offset register instead of WLDR (9 for the 3x3 kinds: the 168-register kernels were spilling 12 dwords)
    constexpr bool TAPW = FAST && (KC * BM / 4 == NTHREADS);

    PROF2_T(q0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;            // [NTAP][KC][BM]
    float* iscl = smem + WSTAGE; // FAST with style scales: those of the block's sample [Kp]
    float* xl = iscl + ((FAST && HAS_ISC) ? p.Kp : 0);   // [KC][CS]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;

    // ---- region + tile coordinates
    // ---- block -> (cell tile, M block).  Consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so
    // the j-th block of XCD x takes the (j / mblocks)-th tile of that XCD and M block j % mblocks: the M blocks of a tile run on
    // ONE XCD at the same time and share its input tile through that L2.  Which tile that is: XCD x walks the x-th eighth of
    // the first region's tile list (te::xcd_banded(): neighbouring tiles share halo rows / columns, i.e. whole 128-byte lines,
    // through one L2 - 1628 -> 664 MB read for the 128 -> 128 @256^2 launch, profiles/experiments/r04_xcd_band_ab.log), then
    // the short edge tiles of T2 interleaved (tile q * 8 + x: every XCD gets its share of the cheap tiles, and they come last
    // for every M block, which is what a greedy dispatcher wants at the tail); a tile id past the end exits.
#if TE_CONV_XCD
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int tq = jx / p.mblocks, mblk = jx % p.mblocks;
    int tile_id;
    if (p.nt8 && tq < p.nt8) {
        tile_id = (int)(((int64_t)xcd * p.nbody) >> 3) + tq;      // band x = tiles [x nbody / 8, (x + 1) nbody / 8)
        if (tile_id >= (int)(((int64_t)(xcd + 1) * p.nbody) >> 3)) return;
    } else {
        tile_id = p.nt8 ? p.nbody + (tq - p.nt8) * 8 + xcd : tq * 8 + xcd;
        if (tile_id >= p.ntiles) return;
    }
#else
    const int tile_id = blockIdx.x % p.ntiles, mblk = blockIdx.x / p.ntiles;
#endif
    const int ridx = (p.nreg > 1 && tile_id >= p.reg[1].first_block) + (p.nreg > 2 && tile_id >= p.reg[2].first_block);
    const ConvArgs::Region g = p.reg[ridx];
    int t = tile_id - g.first_block;
    const int tx_i = t % g.tiles_x; t /= g.tiles_x;
    const int ty_i = t % g.tiles_y; t /= g.tiles_y;
    const int b0 = t * g.NSv;
    const int m0 = mblk * BM;
    const int ci0 = g.ri0 + ty_i * g.TH, cj0 = g.rj0 + tx_i * g.TW;   // first cell of the tile
    // origin of the input tile in input coordinates
    int oy, ox;
    if (KIND == TE_CONV_3X3) { oy = ci0 - 1; ox = cj0 - 1; }
    else if (KIND == TE_CONV_S2) { oy = 2 * ci0; ox = 2 * cj0; }
    else if (KIND == TE_CONV_T2) { oy = ci0 - 1; ox = cj0 - 1; }
    else { oy = ci0; ox = cj0; }

    // ---- per-thread staging descriptors (constant over the K loop)
    const int tile_sp = g.TIH * g.TIW;
    const int n_sp = g.NSv * tile_sp;
    unsigned goff[NSP];      // byte offset of channel 0 relative to the tile's first sample, or OOBH for zero padding
    int loff[NSP], sb[NSP];  // LDS offset (or -1: no element for this thread), sample index
    const unsigned plane4 = (unsigned)p.Hi * p.Wi * 4u;
    if (!FAST) {
#pragma unroll
        for (int r = 0; r < NSP; ++r) {
            const int e = tid + NTHREADS * r;
            goff[r] = OOBH; loff[r] = -1; sb[r] = 0;
            if (e < n_sp) {
                const int s = e / tile_sp, rem = e - s * tile_sp;
                const int ry = rem / g.TIW, rx = rem - ry * g.TIW;
                const int b = b0 + s, gy = oy + ry, gx = ox + rx;
                // S2 keeps even and odd input columns of a row apart ([even | odd]) so the stride-2 B-fragment
                // reads below touch consecutive LDS words (32 banks: a stride of 2 words is a 2-way conflict)
                const int rxl = (KIND == TE_CONV_S2) ? ((rx & 1) ? g.OD + (rx >> 1) : (rx >> 1)) : rx;
                loff[r] = s * g.SS + ry * g.TIWP + g.xo + rxl;
                sb[r] = b < p.B ? b : 0;
                if (b < p.B && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi)
                    goff[r] = (unsigned)s * p.K * plane4 + (unsigned)(gy * p.Wi + gx) * 4u;
            }
        }
    }
    // FAST: the tile of one channel is TIH rows of nseg 16-byte segments; segment sg of a row starts at input column
    // ox - xo + 4 sg (xo = 3 for the kinds with a left halo column, so a segment lies either wholly left of the image or
    // starts inside it: no negative offsets).  A thread owns NQ (segment, channel-of-the-stage) pairs; the stage's first
    // channel comes in through the scalar offset of the load.
    unsigned qgo[FAST ? NQ : 1];     // byte offset incl. the pair's channel, or OOBH (nothing of it lies inside the image / no pair)
    int qlo[FAST ? NQ : 1];          // LDS float offset incl. the channel (threads without a pair: a dump slot behind the tile)
    int qkk[FAST ? NQ : 1];          // channel of the pair inside the stage
    unsigned qvm[FAST ? NQ : 1];     // bit j: component j lies inside the image horizontally
    // Columns the MFMAs read are LDS columns [xo, xo + TIW) of a row.  Those outside the image must hold 0; when they start
    // on a segment boundary (the usual case: image widths and tile origins are multiples of 4) whole segments fail the range
    // check and nothing needs masking at commit time; otherwise (block-uniform) the commit zeroes by component.
    bool need_mask = false;
    if (FAST) {
        const int nitems = g.TIH * g.nseg;
        const int c_lo = g.xo - ox;                  // LDS column of input column 0
        const int c_hi = c_lo + p.Wi;                // LDS column of the first input column past the image
        need_mask = (c_lo > g.xo && (c_lo & 3)) || (c_hi < g.xo + g.TIW && (c_hi & 3));
#pragma unroll
        for (int r = 0; r < NQ; ++r) {
            const int q = tid + NTHREADS * r;
            const int kk = q / nitems, item = q - kk * nitems;
            const int ry = item / g.nseg, sg = item - ry * g.nseg;
            const int gy = oy + ry, gx0 = ox - g.xo + 4 * sg;
            qgo[r] = OOBH; qlo[r] = KC * g.CS; qkk[r] = 0; qvm[r] = 0;
            if (kk < KC) {
                qkk[r] = kk;
                qlo[r] = kk * g.CS + ry * g.TIWP + (KIND == TE_CONV_S2 ? 2 * sg : 4 * sg);
                unsigned used = 0;                   // components that are read AND inside the image
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool in = gx0 + j >= 0 && gx0 + j < p.Wi;
                    qvm[r] |= in ? (1u << j) : 0u;
                    used |= (in && 4 * sg + j >= g.xo && 4 * sg + j < g.xo + g.TIW) ? (1u << j) : 0u;
                }
                if (gy >= 0 && gy < p.Hi && used != 0)           // used != 0 implies gx0 >= 0 (see above)
                    qgo[r] = (unsigned)kk * plane4 + (unsigned)(gy * p.Wi + gx0) * 4u;
            }
        }
    }
    // input of the samples of this tile through one buffer descriptor: 32-bit offsets, hardware zero fill
    const int ns_here = min(g.NSv, p.B - b0);
    const __amdgpu_buffer_rsrc_t irs = make_rsrc(p.in + (size_t)b0 * p.K * p.Hi * p.Wi, (unsigned)ns_here * p.K * plane4);

    // ---- per-lane B-fragment base offsets (LDS floats), one per cell block
    int boff[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int c = wn * (NBW * 32) + nb * 32 + l31;
        const int s = c >> (g.lgTW + g.lgTH);
        const int ty = (c >> g.lgTW) & (g.TH - 1), tx = c & (g.TW - 1);
        int o;
        if (KIND == TE_CONV_S2) o = 2 * ty * g.TIWP + tx;
        else if (KIND == TE_CONV_T2) o = (ty + 1) * g.TIWP + tx + 1;
        else o = ty * g.TIWP + tx;
        boff[nb] = s * g.SS + o + g.xo + half * g.CS;
    }
    const int aoff = half * BM + wm * (MBW * 32) + l31;   // A-fragment base inside wl

    f32x16 acc[MBW][NACC];
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- register prefetch buffers
    f32x4 wreg[WLDR];
    float xreg[FAST ? 1 : NSP][FAST ? 1 : KC];
    float sreg[(HAS_ISC && MS) ? NSP : 1][FAST ? 1 : KC];   // style scales of the staged channels (applied when the tile is written to LDS)
    f32x4 xq[FAST ? NQ : 1];     // FAST: one 16-byte segment per (segment, channel) pair
    float sq[FAST ? NQ : 1];     //       and the style scale of its channel

    const int kbeg = blockIdx.z * p.kchunk, kend = min(p.Kp, kbeg + p.kchunk);
    PROF2_T(q1);
    auto commit = [&](float* wlb, float* xlb) {          // prefetched registers -> one LDS stage buffer
#pragma unroll
        for (int r = 0; r < WLDR; ++r) {
            const int idx = tid + NTHREADS * r;
            if ((WEVEN || idx < WSTAGE / 4) && (!TAP_PIECES || ((g.tapmask >> r) & 1))) *reinterpret_cast<f32x4*>(wlb + idx * 4) = wreg[r];
        }
        if (!FAST) {
#pragma unroll
            for (int r = 0; r < NSP; ++r) {
                if (loff[r] >= 0) {
#pragma unroll
                    for (int kk = 0; kk < KC; ++kk)
                        xlb[kk * g.CS + loff[r]] = HAS_ISC ? xreg[r][kk] * sreg[MS ? r : 0][kk] : xreg[r][kk];
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < NQ; ++r) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = HAS_ISC ? xq[r][j] * sq[r] : xq[r][j];
                if (need_mask) {          // block-uniform, rare: a segment straddles the image border
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = ((qvm[r] >> j) & 1) ? v[j] : 0.f;
                }
                float* d = xlb + qlo[r];
                if (KIND == TE_CONV_S2) {         // even columns | odd columns (row stride and OD are even: 8-byte aligned)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<f32x2*>(d) = f32x2{v[0], v[2]};
                    *reinterpret_cast<f32x2*>(d + g.OD) = f32x2{v[1], v[3]};
                } else {
                    *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                }
            }
        }
    };
    // FAST: weights through a buffer descriptor with per-thread byte offsets fixed over the K loop; a stage advances scalar offsets
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.wp, (unsigned)NTAP * p.Kp * p.Mp * 4u);
    unsigned woff[FAST ? (TAPW ? 1 : WLDR) : 1];
    if (FAST) {
#pragma unroll
        for (int r = 0; r < (TAPW ? 1 : WLDR); ++r) {
            int idx = tid + NTHREADS * r;          // float4 index inside the stage
            if (!WEVEN && idx >= WSTAGE / 4) idx = WSTAGE / 4 - 1;      // clamped duplicate load, never committed
            const int row = idx / (BM / 4), c4 = idx % (BM / 4);      // row = tap*KC + kk ; BM/4 float4 per row
            const int tap = row / KC, kk = row - tap * KC;
            woff[r] = ((unsigned)(tap * p.Kp + kk) * p.Mp + m0 + c4 * 4) * 4u;
        }
    }
    constexpr int NPIECE = WLDR + NQ;             // load instructions per thread and stage
    auto load_piece = [&](int kn, int i) {        // i is a compile-time constant after unrolling
        if (i < WLDR) {
            // T2 at this tile class: one piece is one tap (KC * BM / 4 == NTHREADS), so the thin edge regions neither load
            // nor commit the weights of the taps they skip (block-uniform)
            if (!TAP_PIECES || ((g.tapmask >> i) & 1))
                wreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    wrs, woff[(FAST && !TAPW) ? i : 0], ((unsigned)kn + (TAPW ? (unsigned)i * p.Kp : 0u)) * p.Mp * 4u, 0));
        } else {
            const int r = FAST ? i - WLDR : 0;
            // rows outside the image (qgo = OOBH) fail the hardware range check whatever the scalar channel offset is
            xq[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, qgo[r], (unsigned)kn * plane4, 0));
            if (HAS_ISC) sq[r] = iscl[kn + qkk[r]];
        }
    };
    auto issue = [&](int kn) {                            // global loads of the stage starting at channel kn -> registers
#pragma unroll
        for (int r = 0; r < WLDR; ++r) {
            int idx = tid + NTHREADS * r;          // float4 index inside the stage
            if (!WEVEN && idx >= WSTAGE / 4) idx = WSTAGE / 4 - 1;      // clamped duplicate load, never committed
            const int row = idx / (BM / 4), c4 = idx % (BM / 4);      // row = tap*KC + kk ; BM/4 float4 per row
            const int tap = row / KC, kk = row - tap * KC;
            wreg[r] = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)(tap * p.Kp + kn + kk) * p.Mp + m0 + c4 * 4));
        }
#pragma unroll
        for (int r = 0; r < NSP; ++r) {
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                // unconditional buffer loads: zero padding and the channel tail come back as 0 from the hardware
                // range check, so nothing depends on the loaded value until the tile is committed to LDS
                const int k = kn + kk;
                const unsigned koff = k < p.K ? (unsigned)k * plane4 : OOBH;
                xreg[r][kk] = buf_load(irs, goff[r] + koff);
                if (HAS_ISC && (MS || r == 0))
                    sreg[MS ? r : 0][kk] = p.isc[(MS ? sb[r] : b0) * p.K + (k < p.K ? k : p.K - 1)];
            }
        }
    };
    // one (channel pair, tap) step: A fragments of the wave's M blocks, B fragments of its cell blocks
    auto operands = [&](const float* wlb, const float* xlb, int kk, int tp, float (&a)[MBW], float (&bv)[NBW]) {
        const int ky = tp / 3, kx = tp % 3;
        int toff;
        if (KIND == TE_CONV_1X1) toff = 0;
        else if (KIND == TE_CONV_T2) toff = -(ky == 2 ? g.TIWP : 0) - (kx == 2 ? 1 : 0);
        else if (KIND == TE_CONV_S2) toff = ky * g.TIWP + (kx == 1 ? g.OD : (kx >> 1));
        else toff = ky * g.TIWP + kx;
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) a[mb] = wlb[(tp * KC + kk) * BM + aoff + mb * 32];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) bv[nb] = xlb[kk * g.CS + boff[nb] + toff];
    };
    auto mfmas = [&](int tp, const float (&a)[MBW], const float (&bv)[NBW]) {
        const int ky = tp / 3, kx = tp % 3;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int j = IS_T2 ? nb * 4 + ((ky == 1) ? 2 : 0) + ((kx == 1) ? 1 : 0) : nb;
#pragma unroll
            for (int mb = 0; mb < MBW; ++mb)
                acc[mb][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], bv[nb], acc[mb][j], 0, 0, 0);
        }
    };
    auto compute = [&](const float* wlb, const float* xlb) {
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) {
#pragma unroll
            for (int tp = 0; tp < NTAP; ++tp) {
                if (IS_T2 && !((g.tapmask >> tp) & 1)) continue;      // block-uniform: skipped taps only see zero padding
                float a[MBW], bv[NBW];
                operands(wlb, xlb, kk, tp, a, bv);
                mfmas(tp, a, bv);
            }
        }
    };
    // FAST form of the stage: the step list (channel pair x tap) is walked with the LDS operands of the next step already
    // requested, and the load instructions of stage `kn` are dealt out over the first three quarters of the steps, so no wave
    // queues 25 loads at the texture addresser at once and the last of them has time to land before the commit
    auto compute_fast = [&](const float* wlb, const float* xlb, int kn) {
        constexpr int NSTEP = (KC / 2) * NTAP;
        constexpr int SPAN = TE_FAST_IL ? (NSTEP * 3) / 4 : 1;     // steps that carry loads
        auto pieces = [&](int st) {
#pragma unroll
            for (int i = 0; i < NPIECE; ++i)
                if ((i * SPAN) / NPIECE == st) load_piece(kn, i);
        };
        if (!IS_T2) {
            float a0[MBW], b0[NBW];
            operands(wlb, xlb, 0, 0, a0, b0);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                const int kk = (st / NTAP) * 2, tp = st % NTAP;
                float a1[MBW], b1[NBW];
                if (TE_FAST_PF) {
                    if (st + 1 < NSTEP) operands(wlb, xlb, ((st + 1) / NTAP) * 2, (st + 1) % NTAP, a1, b1);
                } else {
                    operands(wlb, xlb, kk, tp, a0, b0);
                }
                pieces(st);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(tp, a0, b0);
                if (TE_FAST_PF && st + 1 < NSTEP) {
#pragma unroll
                    for (int mb = 0; mb < MBW; ++mb) a0[mb] = a1[mb];
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) b0[nb] = b1[nb];
                }
            }
        } else {
            // T2: the B fragment of a tap depends on (ky == 2, kx == 2) only -> four shifted fragments per channel pair,
            // shared by the nine taps; those of the next pair arrive one per tap while this pair is multiplied
            static_assert(!IS_T2 || MBW == 1, "T2 tile classes use one M block per wave");
            auto bfrag = [&](int kk, int sh, int nb) {
                return xlb[kk * g.CS + boff[nb] - ((sh >> 1) ? g.TIWP : 0) - (sh & 1)];
            };
            float bs[4][NBW], bn[4][NBW];
#pragma unroll
            for (int sh = 0; sh < 4; ++sh)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) bs[sh][nb] = bfrag(0, sh, nb);
            float a0 = wlb[aoff];
#pragma unroll
            for (int kk = 0; kk < KC; kk += 2) {
#pragma unroll
              for (int tp = 0; tp < NTAP; ++tp) {
                const int st = (kk / 2) * NTAP + tp;
                const int ky = tp / 3, kx = tp % 3;
                float a1 = 0.f;
                if (st + 1 < NSTEP) a1 = wlb[(((st + 1) % NTAP) * KC + ((st + 1) / NTAP) * 2) * BM + aoff];
                if (kk + 2 < KC) {          // 4 * NBW fragments of the next pair over the 9 taps of this one
#pragma unroll
                    for (int q = 0; q < 4 * NBW; ++q)
                        if ((q * NTAP) / (4 * NBW) == tp) bn[q / NBW][q % NBW] = bfrag(kk + 2, q / NBW, q % NBW);
                }
                pieces(st);
                __builtin_amdgcn_sched_barrier(0);
                const int sh = (ky == 2 ? 2 : 0) + (kx == 2 ? 1 : 0);
                if ((g.tapmask >> tp) & 1) {       // block-uniform: the thin edge regions skip the taps that only see zero padding
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) {
                        const int j = nb * 4 + ((ky == 1) ? 2 : 0) + ((kx == 1) ? 1 : 0);
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bs[sh][nb], acc[0][j], 0, 0, 0);
                    }
                }
                a0 = a1;
              }
              if (kk + 2 < KC) {
#pragma unroll
                  for (int sh2 = 0; sh2 < 4; ++sh2)
#pragma unroll
                      for (int nb = 0; nb < NBW; ++nb) bs[sh2][nb] = bn[sh2][nb];
              }
            }
        }
    };
    if (FAST) {
        if (HAS_ISC) {
            for (int k = tid; k < p.Kp; k += NTHREADS) iscl[k] = k < p.K ? p.isc[(size_t)b0 * p.K + k] : 0.f;
            __syncthreads();
        }
        // prologue: stage kbeg straight to registers; every iteration commits its stage, then multiplies it while the loads
        // of the next one are dealt out (the last iteration re-requests its own stage: never committed, always in range)
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) load_piece(kbeg, i);
#ifdef TE_CONV_PROF
        unsigned long long pc[5] = {0, 0, 0, 0, 0};
        const unsigned long long pstart = __builtin_readcyclecounter();
        const unsigned long long rstart = __builtin_amdgcn_s_memrealtime();
#endif
        for (int k0 = kbeg; k0 < kend; k0 += KC) {
            PROF_T(t0);
            __syncthreads();              // every wave finished reading the previous stage
            PROF_T(t1);
#ifdef TE_CONV_PROF
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): separate "waiting for the stage loads" from the LDS writes
            const unsigned long long t1b = __builtin_readcyclecounter();
            pc[4] += t1b - t1;
#endif
            commit(wl, xl);
            PROF_T(t2);
            __syncthreads();
            PROF_T(t3);
            compute_fast(wl, xl, min(k0 + KC, kend - KC));
#ifdef TE_CONV_PROF
            const unsigned long long t4 = __builtin_readcyclecounter();
            pc[0] += t1 - t0; pc[1] += t2 - t1; pc[2] += t3 - t2; pc[3] += t4 - t3;
#endif
        }
#ifdef TE_CONV_PROF
        const int lin = blockIdx.x + gridDim.x * blockIdx.z;
        if ((tid & 63) == 0 && lin < 8192 / 4) {
            unsigned long long* d = te_conv_prof_buf + ((size_t)lin * 4 + wid) * 8;
            d[0] = pc[0]; d[1] = pc[1]; d[2] = pc[2]; d[3] = pc[3]; d[4] = pstart; d[5] = __builtin_readcyclecounter();
            d[6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
            d[7] = (kend - kbeg) / KC; d[6] = pc[4];
            d[0] = pc[0] + (((unsigned long long)(__builtin_amdgcn_s_memrealtime() - rstart)) << 32);   // high half: 100 MHz ticks of the K loop
        }
#endif
    } else {
        // software pipeline: iteration `k0` commits stage k0 (prefetched by the previous iteration) to LDS, issues the
        // global loads of stage k0+KC, then runs the MFMAs of stage k0 while those loads are in flight.
#ifdef TE_CONV_PROF2
        unsigned long long qa = 0, qb = 0, qc = 0, qw = 0;
#endif
        for (int k0 = kbeg - KC; k0 < kend; k0 += KC) {
            PROF2_T(s0);
            if (k0 >= kbeg) {
                __syncthreads();              // every wave finished reading the previous stage
#ifdef TE_CONV_PROF2
                { PROF2_T(w0); __builtin_amdgcn_s_waitcnt(0x0F70); qw += __builtin_amdgcn_s_memrealtime() - w0; }   // vmcnt(0): waiting for the stage loads
#endif
                commit(wl, xl);
                __syncthreads();
            }
            PROF2_T(s1);
            const int kn = k0 + KC;
            if (kn < kend) issue(kn);
            PROF2_T(s2);
#ifdef TE_CONV_PROF2
            qa += s1 - s0; qb += s2 - s1;
#endif
            if (k0 < kbeg) continue;
            compute(wl, xl);
#ifdef TE_CONV_PROF2
            qc += __builtin_amdgcn_s_memrealtime() - s2;
#endif
        }
#ifdef TE_CONV_PROF2
        if (tid == 0) {
            const int lin = blockIdx.x + gridDim.x * blockIdx.z;
            if (lin < 8192) { unsigned long long* d = te_conv_prof_buf + (size_t)lin * 8; d[5] = qa | (qw << 32); d[6] = qb | (qc << 32); }
        }
#endif
    }

    PROF2_T(q2);
    // ---- epilogue: osc, bias, activation, store.  C/D layout of the 32x32 tile:
    //      col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const size_t oplane = (size_t)p.Ho * p.Wo;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int c = wn * (NBW * 32) + nb * 32 + l31;
        const int s = c >> (g.lgTW + g.lgTH);
        const int ci = ci0 + ((c >> g.lgTW) & (g.TH - 1)), cj = cj0 + (c & (g.TW - 1));
        const int b = b0 + s;
        const bool cell_ok = s < g.NSv && b < p.B && ci < g.ri0 + g.rh && cj < g.rj0 + g.rw;
        const int bc = b < p.B ? b : p.B - 1;
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) {
            const int mbase = m0 + wm * (MBW * 32) + mb * 32 + 4 * half;
            // batch the per-row scale / bias loads (clamped, unconditional) ahead of the stores
            float sc[16], bi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = 1.f; bi[r] = 0.f; }
            if (p.osc) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    sc[r] = p.osc[(size_t)bc * p.M + (m < p.M ? m : p.M - 1)];
                }
            }
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    bi[r] = p.bias[m < p.M ? m : p.M - 1];
                }
            }
            if (IS_T2 && p.ksplit == 1) {
                // the same addressing for the transposed kind: output (2 ci + a, 2 cj + b); the two column phases of a cell are
                // adjacent outputs: one 8-byte store per row phase (rows are 2W+1 wide, so the pair is only 4-byte aligned,
                // which global stores accept)
                const size_t off0 = ((size_t)bc * p.M + mbase) * oplane + (size_t)(2 * ci) * p.Wo + 2 * cj;
                float* o0 = p.out + off0;
                const bool x1 = 2 * cj + 1 < p.Wo, x0 = 2 * cj < p.Wo;
                const float gain = p.act == 3 ? 1.4142135623730951f : 1.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    const bool ok = cell_ok && mbase + dm < p.M;
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2) {
                        float v0 = acc[mb][nb * 4 + 2 * a2][r] * sc[r] + bi[r];
                        float v1 = acc[mb][nb * 4 + 2 * a2 + 1][r] * sc[r] + bi[r];
                        if (p.act >= 3) {
                            v0 = (v0 > 0.f ? v0 : v0 * 0.2f) * gain;
                            v1 = (v1 > 0.f ? v1 : v1 * 0.2f) * gain;
                        }
                        if (ok && 2 * ci + a2 < p.Ho) {
                            float* dst = o0 + (size_t)dm * oplane + (a2 ? p.Wo : 0);
                            if (x1) {
                                typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
                                *reinterpret_cast<f32x2u*>(dst) = f32x2u{v0, v1};
                            } else if (x0) {
                                dst[0] = v0;
                            }
                        }
                    }
                }
                continue;
            }
            if (!IS_T2 && p.ksplit == 1) {
                // One-pass store path (block-uniform branch).  The 64-bit element offset is formed ONCE per accumulator tile;
                // the 16 rows of the tile differ from it by wave-uniform multiples of the plane size (round 2 rebuilt the
                // whole address per element: ~10 vector-ALU instructions each, three times that with the residual and mask
                // stages, which made their epilogue cost more than the passes it replaces).
                const size_t off0 = ((size_t)bc * p.M + mbase) * oplane + (size_t)ci * p.Wo + cj;
                float* o0 = p.out + off0;
                float rv[EPI ? 16 : 1], qv[EPI ? 16 : 1];
                if (EPI) {           // residual / mask values of the whole tile requested up front: one batch of loads in flight
                    const float* r0 = p.res ? p.res + off0 : nullptr;
                    const float* q0 = p.mref ? p.mref + off0 : nullptr;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dm = (r & 3) + 8 * (r >> 2);
                        const bool ok = cell_ok && mbase + dm < p.M;
                        rv[r] = (r0 && ok) ? r0[(size_t)dm * oplane] : 0.f;
                        qv[r] = (q0 && ok) ? q0[(size_t)dm * oplane] : 1.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    float v = acc[mb][nb][r] * sc[r] + bi[r];
                    if (p.act >= 3) v = (v > 0.f ? v : v * 0.2f) * (p.act == 3 ? 1.4142135623730951f : 1.f);
                    if (EPI) {
                        v += rv[r];
                        if (p.mref) v *= qv[r] > 0.f ? p.mgain : 0.2f * p.mgain;
                    }
                    if (cell_ok && mbase + dm < p.M) o0[(size_t)dm * oplane] = v;
                }
                continue;
            }
            // Split launches (small images): raw partial sums; scale / bias / activation happen in conv_finalize_kernel.  Every
            // split writes its own slab of the workspace (plain stores, summed in fixed order: deterministic; a launch without a
            // workspace is never split).  Same addressing as above: one offset per tile, uniform row strides.
            {
                const size_t off0 = ((size_t)bc * p.M + mbase) * oplane + (size_t)(IS_T2 ? 2 * ci : ci) * p.Wo + (IS_T2 ? 2 * cj : cj);
                float* b0 = p.ws + (size_t)blockIdx.z * p.B * p.M * oplane + off0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    if (!(cell_ok && mbase + dm < p.M)) continue;
                    float* d = b0 + (size_t)dm * oplane;
                    if (!IS_T2) {
                        d[0] = acc[mb][nb][r];
                    } else {
#pragma unroll
                        for (int ph = 0; ph < 4; ++ph) {
                            if (2 * ci + (ph >> 1) < p.Ho && 2 * cj + (ph & 1) < p.Wo) {
                                float* e = d + ((ph >> 1) ? p.Wo : 0) + (ph & 1);
                                e[0] = acc[mb][nb * 4 + ph][r];
                            }
                        }
                    }
                }
            }
        }
    }
#ifdef TE_CONV_PROF2
    {
        const int lin = blockIdx.x + gridDim.x * blockIdx.z;
        if (tid == 0 && lin < 8192) {
            unsigned long long* d = te_conv_prof_buf + (size_t)lin * 8;
            d[0] = q0; d[1] = q1; d[2] = q2; d[3] = __builtin_amdgcn_s_memrealtime(); d[4] = (kend - kbeg) / KC; d[7] = 1;
        }
    }
#endif
}

// epilogue of the split-K path: out = act((sum_z ws[z] | out) * osc[b,m] + bias[m]); the slabs are summed in fixed order
__global__ __launch_bounds__(256) void conv_finalize_kernel(float* __restrict__ out, const float* __restrict__ ws, int ksplit,
                                                            const float* __restrict__ osc,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            const float* __restrict__ mref, float mgain, int act,
                                                            int M, int plane, int64_t total) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t bm = e / plane;
        float v;
        if (ws) {
            v = 0.f;
            for (int z = 0; z < ksplit; ++z) v += ws[(int64_t)z * total + e];
        } else {
            v = out[e];
        }
        if (osc) v *= osc[bm];
        if (bias) v += bias[bm % M];
        if (act >= 3) v = (v > 0.f ? v : v * 0.2f) * (act == 3 ? 1.4142135623730951f : 1.f);
        if (res) v += res[e];
        if (mref) v *= mref[e] > 0.f ? mgain : 0.2f * mgain;
        out[e] = v;
    }
}

// ------------------------------------------------------------------------------------------ weight packing
// up to 64 (weight, layout) jobs in one launch: the packed layouts of a whole model are refreshed right after an optimiser step
// (blockIdx.y = job) instead of one tiny launch per layer at first use
// TE_PACK_T6FWD / T6SWAP: [split fragment-order layout, 27 K M bf16][plain layout 9 Kp Mp floats]; offset of the second part in floats
__host__ __device__ inline int64_t t6_plain_offset(int K, int M) { return ((((int64_t)27 * K * M + 1) / 2) + 3) / 4 * 4; }
struct PackJob { float* wp; const float* w; float wscale; int kind, Ci, ntap, K, M, Kp, Mp; };
struct PackJobs { PackJob j[64]; };
// Tiled through LDS (round 4): a block takes a 32 (co) x 32 (ci) x taps tile.  The model layout [Co][Ci][tap] is read in rows of
// 32 * taps contiguous floats, the packed layouts are written in runs of 32 contiguous floats (over co for the forward layout
// Wp[tap][ci][co], over ci for the data-gradient layouts Wp[tap'][co][ci]); LDS rows are 32 * taps + 1 floats apart (odd: the
// transposing reads are conflict-free).  The element-per-thread form it replaces gathered 4 bytes per cache line for the forward
// layout (lanes 18 KB apart): 528 us per refresh of a model's 28 - 38 layouts, ~1.3 ms per training iteration.
constexpr int PT = 32;

template <int T>
__device__ __forceinline__ void pack_tiles(const PackJob& q, float* tile) {
    constexpr int RL = PT * T + 1, NE = PT * PT * T;
    const bool fwd = q.kind == TE_PACK_FWD || q.kind == TE_PACK_WFWD || q.kind == TE_PACK_W6FWD || q.kind == TE_PACK_S6FWD || q.kind == TE_PACK_T6FWD ||
                     q.kind == TE_PACK_P6FWD;
    const bool t6 = q.kind == TE_PACK_T6FWD || q.kind == TE_PACK_T6SWAP;      // split layout + plain layout behind it
    const int Co = fwd ? q.M : q.K, Ci = fwd ? q.K : q.M;              // real extents of the source along co / ci
    const int CoP = fwd ? q.Mp : q.Kp, CiP = fwd ? q.Kp : q.Mp;        // padded extents of the packed layout (zero filled)
    const int tiles_i = (CiP + PT - 1) / PT, tiles_c = (CoP + PT - 1) / PT;
    for (int t = blockIdx.x; t < tiles_i * tiles_c; t += gridDim.x) {
        const int c0 = (t / tiles_i) * PT, i0 = (t % tiles_i) * PT;
        __syncthreads();
        for (int e = threadIdx.x; e < NE; e += 256) {
            const int r = e / (PT * T), j = e - r * (PT * T);
            float v = 0.f;
            if (c0 + r < Co && i0 * T + j < Ci * T) v = q.w[((size_t)(c0 + r) * q.Ci + i0) * T + j];
            tile[r * RL + j] = v * q.wscale;
        }
        __syncthreads();
        if (T == 9 && (q.kind == TE_PACK_W6FWD || q.kind == TE_PACK_W6DGRAD)) {
            // the same transform, every value split into three bf16 pieces and stored in MFMA fragment order (wino6.hip):
            // U6[k / 16][piece][ky][c][m / 32][lane = m % 32 + 32 * (k % 16 / 8)][k % 8]; a 32 x 32 tile fills whole 1 KB slots, and
            // consecutive threads write consecutive 2-byte elements of a slot
            const bool wf = q.kind == TE_PACK_W6FWD;
            unsigned short* dst = reinterpret_cast<unsigned short*>(q.wp);
            const int MT = q.Mp >> 5;
            for (int e = threadIdx.x; e < PT * PT * 12; e += 256) {
                const int j = e & 7, ml = (e >> 3) & 31, kh = (e >> 8) & 1, st = (e >> 9) & 1, kc = e >> 10;      // kc = ky * 4 + c
                const int kl = st * 16 + kh * 8 + j;                                    // k inside the tile
                const int r = wf ? ml : kl, ii = wf ? kl : ml;                          // tile row = co, tile column = ci
                const int ky = kc >> 2, c = kc & 3;
                const int co = c0 + r, ci = i0 + ii;
                if (co >= Co || ci >= Ci) continue;
                const float* src = tile + r * RL + ii * T + (wf ? ky * 3 : (2 - ky) * 3);
                const float g0 = wf ? src[0] : src[2], g1 = src[1], g2 = wf ? src[2] : src[0];
                const float v = c == 0 ? g0 : (c == 1 ? 0.5f * (g0 + g1 + g2) : (c == 2 ? 0.5f * (g0 - g1 + g2) : g2));
                const int m = wf ? co : ci, k = wf ? ci : co;
                unsigned u = __builtin_bit_cast(unsigned, v);
                const unsigned h = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
                const float r1 = v - __builtin_bit_cast(float, h << 16);
                u = __builtin_bit_cast(unsigned, r1);
                const unsigned mm = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
                const float r2 = r1 - __builtin_bit_cast(float, mm << 16);
                u = __builtin_bit_cast(unsigned, r2);
                const unsigned ll = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
                const size_t slot = (((size_t)(k >> 4) * 3 * 3 + ky) * 4 + c) * MT + (m >> 5);      // piece 0; piece stride 12 * MT slots
                const size_t at = slot * 512 + (size_t)((m & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
                const size_t ps = (size_t)12 * MT * 512;
                dst[at] = (unsigned short)h; dst[at + ps] = (unsigned short)mm; dst[at + 2 * ps] = (unsigned short)ll;
            }
        } else if ((T == 9 && (q.kind == TE_PACK_S6FWD || q.kind == TE_PACK_S6SWAP || t6)) ||
                   (T == 1 && (q.kind == TE_PACK_P6FWD || q.kind == TE_PACK_P6DGRAD))) {
            // the taps as stored, every value split into three bf16 pieces, MFMA fragment order (s2s6.hip; T == 1: p1s6.hip):
            // S6[k / 16][piece][tap][m / 32][lane = m % 32 + 32 * (k % 16 / 8)][k % 8]; forward: m = co, k = ci; swap: m = ci, k = co
            const bool wf = q.kind == TE_PACK_S6FWD || q.kind == TE_PACK_T6FWD || q.kind == TE_PACK_P6FWD;
            unsigned short* dst = reinterpret_cast<unsigned short*>(q.wp);
            const int MT = q.M >> 5;
            for (int e = threadIdx.x; e < PT * PT * T; e += 256) {
                const int j = e & 7, ml = (e >> 3) & 31, kh = (e >> 8) & 1, st = (e >> 9) & 1, tap = e >> 10;
                const int kl = st * 16 + kh * 8 + j;                                    // k inside the tile
                const int r = wf ? ml : kl, ii = wf ? kl : ml;                          // tile row = co, tile column = ci
                const int co = c0 + r, ci = i0 + ii;
                if (co >= Co || ci >= Ci) continue;
                const float v = tile[r * RL + ii * T + tap];
                const int m = wf ? co : ci, k = wf ? ci : co;
                unsigned u = __builtin_bit_cast(unsigned, v);
                const unsigned h = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
                const float r1 = v - __builtin_bit_cast(float, h << 16);
                u = __builtin_bit_cast(unsigned, r1);
                const unsigned mm = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
                const float r2 = r1 - __builtin_bit_cast(float, mm << 16);
                u = __builtin_bit_cast(unsigned, r2);
                const unsigned ll = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
                const size_t slot = ((size_t)(k >> 4) * 3 * T + tap) * MT + (m >> 5);               // piece 0; piece stride T * MT slots
                const size_t at = slot * 512 + (size_t)((m & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
                const size_t ps = (size_t)T * MT * 512;
                dst[at] = (unsigned short)h; dst[at + ps] = (unsigned short)mm; dst[at + 2 * ps] = (unsigned short)ll;
            }
            if (t6) {      // TE_PACK_T6FWD / T6SWAP: the plain layout of the same weights behind the split one (thin fp32 regions of TE_CONV_T2S6)
                float* wp2 = q.wp + t6_plain_offset(q.K, q.M);
                for (int e = threadIdx.x; e < NE; e += 256) {
                    if (fwd) {
                        const int r = e % PT, x = e / PT, ii = x % PT, tap = x / PT;
                        const int co = c0 + r, ci = i0 + ii;
                        if (co < CoP && ci < CiP) wp2[((size_t)tap * q.Kp + ci) * q.Mp + co] = tile[r * RL + ii * T + tap];
                    } else {
                        const int ii = e % PT, x = e / PT, r = x % PT, tap = x / PT;
                        const int co = c0 + r, ci = i0 + ii;
                        if (co < CoP && ci < CiP) wp2[((size_t)tap * q.Kp + co) * q.Mp + ci] = tile[r * RL + ii * T + tap];
                    }
                }
            }
        } else if (T == 9 && (q.kind == TE_PACK_WFWD || q.kind == TE_PACK_WDGRAD)) {
            // Winograd F(2,3) weight transform (wino.hip): U[((k / 8 * 3 + ky) * 4 + c) * 8 + k % 8][m] = sum_kx G[c][kx] w(.., ky, kx),
            // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; forward: m = co, k = ci; data gradient: m = ci, k = co, taps flipped
            const bool wf = q.kind == TE_PACK_WFWD;
            for (int e = threadIdx.x; e < PT * PT * 12; e += 256) {
                const int lanei = e % PT, x = e / PT, other = x % PT, kc = x / PT;        // kc = ky * 4 + c
                const int r = wf ? lanei : other, ii = wf ? other : lanei;                // lanes run over m (contiguous in U)
                const int ky = kc >> 2, c = kc & 3;
                const int co = c0 + r, ci = i0 + ii;
                if (co >= Co || ci >= Ci) continue;
                const float* src = tile + r * RL + ii * T + (wf ? ky * 3 : (2 - ky) * 3);
                const float g0 = wf ? src[0] : src[2], g1 = src[1], g2 = wf ? src[2] : src[0];
                const float v = c == 0 ? g0 : (c == 1 ? 0.5f * (g0 + g1 + g2) : (c == 2 ? 0.5f * (g0 - g1 + g2) : g2));
                const int m = wf ? co : ci, k = wf ? ci : co;
                q.wp[((size_t)((k >> 3) * 3 + ky) * 4 + c) * 8 * q.Mp + (size_t)(k & 7) * q.Mp + m] = v;
            }
        } else if (fwd) {                                              // Wp[(tap * Kp + ci) * Mp + co]: lanes over co
            for (int e = threadIdx.x; e < NE; e += 256) {
                const int r = e % PT, x = e / PT, ii = x % PT, tap = x / PT;
                const int co = c0 + r, ci = i0 + ii;
                if (co < CoP && ci < CiP) q.wp[((size_t)tap * q.Kp + ci) * q.Mp + co] = tile[r * RL + ii * T + tap];
            }
        } else {                                                       // Wp[(tap' * Kp + co) * Mp + ci]: lanes over ci
            for (int e = threadIdx.x; e < NE; e += 256) {
                const int ii = e % PT, x = e / PT, r = x % PT, tap = x / PT;
                const int co = c0 + r, ci = i0 + ii;
                const int tp = (q.kind == TE_PACK_DGRAD) ? T - 1 - tap : tap;
                if (co < CoP && ci < CiP) q.wp[((size_t)tp * q.Kp + co) * q.Mp + ci] = tile[r * RL + ii * T + tap];
            }
        }
    }
}

__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackJobs P) {
    __shared__ float tile[PT * (PT * 9 + 1)];
    const PackJob& q = P.j[blockIdx.y];
    if (q.ntap == 9) pack_tiles<9>(q, tile);
    else pack_tiles<1>(q, tile);
}

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline int pow2ceil(int v) { return 1 << ilog2(v); }
inline int roundup(int v, int m) { return (v + m - 1) / m * m; }

struct PackDims { int K, M, Kp, Mp, ntap; };
inline bool pack_is_wino6(int kind_pack) { return kind_pack == TE_PACK_W6FWD || kind_pack == TE_PACK_W6DGRAD; }
inline bool pack_is_s6(int kind_pack) { return kind_pack == TE_PACK_S6FWD || kind_pack == TE_PACK_S6SWAP; }
inline bool pack_is_t6(int kind_pack) { return kind_pack == TE_PACK_T6FWD || kind_pack == TE_PACK_T6SWAP; }
inline bool pack_is_p6(int kind_pack) { return kind_pack == TE_PACK_P6FWD || kind_pack == TE_PACK_P6DGRAD; }
inline bool pack_is_wino(int kind_pack) { return kind_pack == TE_PACK_WFWD || kind_pack == TE_PACK_WDGRAD || pack_is_wino6(kind_pack); }
inline PackDims pack_dims(int kind_pack, int Co, int Ci, int ksize) {
    PackDims d;
    d.ntap = ksize * ksize;
    const bool fwd = kind_pack == TE_PACK_FWD || kind_pack == TE_PACK_WFWD || kind_pack == TE_PACK_W6FWD || kind_pack == TE_PACK_S6FWD ||
                     kind_pack == TE_PACK_T6FWD || kind_pack == TE_PACK_P6FWD;
    d.M = fwd ? Co : Ci;
    d.K = fwd ? Ci : Co;
    // the Winograd layouts (fp32: U[K/8][ky][component][8][M], K % 8 == 0 and M % 8 == 0) and the split fragment-order layouts (W6 / S6 /
    // P6: [K/16][piece][...][M/32][64 lanes][8 bf16], M % 32 == 0 and K % 16 == 0 - 32 for W6) are not padded: pack_launch checks the multiples
    if (pack_is_wino(kind_pack) || pack_is_s6(kind_pack) || pack_is_p6(kind_pack)) {
        d.Kp = d.K; d.Mp = d.M;
    } else {
        d.Kp = roundup(d.K, KPAD);
        d.Mp = roundup(d.M, MPAD);
    }
    return d;
}

template <int KIND, int TC>
int add_region(ConvArgs& a, int ri0, int rj0, int rh, int rw, int& nblocks, size_t& lds_floats, bool& ms, bool force_single, bool fast) {
    if (rh <= 0 || rw <= 0) return 0;
    constexpr int KC = Cfg<KIND, TC>::KC, BM = tile_bm<KIND, TC>();
    constexpr int NTILE = tile_cells<KIND, TC>();
    ConvArgs::Region& g = a.reg[a.nreg];
    g.ri0 = ri0; g.rj0 = rj0; g.rh = rh; g.rw = rw;
    constexpr int NSP_ = Cfg<KIND, TC>::NSP;
    auto tile_in = [](int th, int tw) {        // input-tile elements of one sample
        if (KIND == TE_CONV_3X3) return (th + 2) * (tw + 2);
        if (KIND == TE_CONV_S2) return (2 * th + 1) * (2 * tw + 1);
        if (KIND == TE_CONV_T2) return (th + 1) * (tw + 1);
        return th * tw;
    };
    constexpr int NSP = Cfg<KIND, TC>::NSP, NQ = Cfg<KIND, TC>::NQ;
    auto in_rows = [](int th) { return KIND == TE_CONV_3X3 ? th + 2 : (KIND == TE_CONV_S2 ? 2 * th + 1 : (KIND == TE_CONV_T2 ? th + 1 : th)); };
    auto in_cols = [](int tw) { return KIND == TE_CONV_3X3 ? tw + 2 : (KIND == TE_CONV_S2 ? 2 * tw + 1 : (KIND == TE_CONV_T2 ? tw + 1 : tw)); };
    g.TW = std::min(32, pow2ceil(rw));
    g.TH = std::min(pow2ceil(rh), NTILE / g.TW);
    g.xo = 0; g.nseg = 0;
    if (!fast) {
        while (g.TH > 1 && tile_in(g.TH, g.TW) > NSP_ * NTHREADS) g.TH >>= 1;   // thin edge regions: one sample must fit
    } else {
        // FAST staging: rows of 16-byte segments; the kinds with a left halo column start their rows 3 columns early so
        // that no segment straddles the left image border.  NQ (segment, channel) pairs per thread must cover a stage.
        g.xo = ((KIND == TE_CONV_3X3 || KIND == TE_CONV_T2) && rj0 == 0) ? 3 : 0;   // regions off the left border never reach column -1
        g.nseg = (g.xo + in_cols(g.TW) + 3) / 4;
        while (g.TH > 1 && in_rows(g.TH) * g.nseg * KC > NQ * NTHREADS) g.TH >>= 1;
        if (in_rows(g.TH) * g.nseg * KC > NQ * NTHREADS)
            return te::fail(TE_ERR_UNSUPPORTED, "te_conv_f32: input tile of %d rows x %d segments exceeds the staging budget", in_rows(g.TH), g.nseg);
    }
    g.NS = NTILE / (g.TW * g.TH);
    g.lgTW = ilog2(g.TW); g.lgTH = ilog2(g.TH);
    g.tiles_x = (rw + g.TW - 1) / g.TW;
    g.tiles_y = (rh + g.TH - 1) / g.TH;
    g.TIH = in_rows(g.TH); g.TIW = in_cols(g.TW);
    g.TIWP = fast ? 4 * g.nseg : g.TIW;
    g.OD = fast ? 2 * g.nseg : g.TW + 1;
    g.SS = g.TIH * g.TIWP;
    g.CS = g.NS * g.SS;
    // edge regions of a single-sample launch stay single-sample, so the whole launch keeps the cheap scalar
    // style-scale path (MS = false); their tiles are merely less full
    g.NSv = force_single ? 1 : g.NS;
    while (!fast && g.NSv > 1 && g.NSv * g.TIH * g.TIW > NSP * NTHREADS) g.NSv >>= 1;
    if ((int64_t)g.NSv * a.K * a.Hi * a.Wi * 4 >= (int64_t)OOBH)
        return te::fail(TE_ERR_UNSUPPORTED, "te_conv_f32: %d samples x %d channels x %dx%d exceed 1 GiB per tile group", g.NSv, a.K, a.Hi, a.Wi);
    if (!fast && g.NSv * g.TIH * g.TIW > NSP * NTHREADS)
        return te::fail(TE_ERR_UNSUPPORTED, "te_conv_f32: input tile %dx%dx%d exceeds the staging budget", g.NSv, g.TIH, g.TIW);
    g.first_block = nblocks;
    g.tapmask = 0x1FF;
    if (KIND == TE_CONV_T2 && ri0 == a.H && rh == 1) g.tapmask = 0x1C0;            // last output row: only ky == 2 reaches it
    if (KIND == TE_CONV_T2 && rj0 == a.W && rw == 1) g.tapmask = 0x124;            // last output column: only kx == 2
    nblocks += g.tiles_x * g.tiles_y * ((a.B + g.NSv - 1) / g.NSv);
    lds_floats = std::max(lds_floats, (size_t)Kind<KIND>::NT * KC * BM + (size_t)KC * g.CS + (size_t)(fast ? 64 : 0) /* dump slot of threads without a staging pair */ + (size_t)((fast && a.isc) ? a.Kp : 0));
    ms = ms || g.NSv > 1;
    a.nreg++;
    return 0;
}

#ifndef TE_CONV_OCC
#define TE_CONV_OCC 3
#endif
constexpr int conv_occ() { return TE_CONV_OCC; }      // waves per SIMD the plain 3x3 128-row tile is compiled for (2: 135.7 vs 140.1 TFLOP/s with the FAST kernel)

template <int KIND, int TC, bool HAS_ISC, bool MS, int OCC, bool FAST, bool EPI>
void launch_f(const ConvArgs& a, int nblocks, size_t lds_floats, hipStream_t s) {
    constexpr int BM = tile_bm<KIND, TC>();
    static std::atomic<uint64_t> attr_done{0};
    te::allow_big_lds(attr_done, (const void*)conv_mfma_kernel<KIND, TC, HAS_ISC, MS, OCC, FAST, EPI>, 128 * 1024);
    ConvArgs b = a;
    b.ntiles = nblocks; b.mblocks = (int)te::cdiv(a.M, BM);
    b.nbody = (a.nreg > 1) ? a.reg[1].first_block : nblocks;
    b.nt8 = te::xcd_banded() ? (int)te::cdiv(b.nbody, 8) : 0;
    const int64_t per_xcd = b.nt8 ? b.nt8 + te::cdiv(nblocks - b.nbody, 8) : te::cdiv(nblocks, 8);
#if TE_CONV_XCD
    dim3 grid((unsigned)(per_xcd * 8 * b.mblocks), 1u, (unsigned)a.ksplit);
#else
    dim3 grid((unsigned)(nblocks * b.mblocks), 1u, (unsigned)a.ksplit);
#endif
    conv_mfma_kernel<KIND, TC, HAS_ISC, MS, OCC, FAST, EPI><<<grid, NTHREADS, lds_floats * sizeof(float), s>>>(b);
}

template <int KIND, int TC> constexpr bool have_fast() { return TE_CONV_FAST && (TC == 3 || ((TC == 0 || TE_CONV_FAST_NARROW) && KIND != TE_CONV_1X1)); }

template <int KIND, int TC, bool HAS_ISC, bool MS, int OCC, bool EPI>
void launch_o(const ConvArgs& a, int nblocks, size_t lds_floats, hipStream_t s, bool fast) {
    constexpr bool HAVE_FAST = have_fast<KIND, TC>() && !MS;
    if (HAVE_FAST && fast) launch_f<KIND, TC, HAS_ISC, MS, OCC, HAVE_FAST, EPI>(a, nblocks, lds_floats, s);
    else launch_f<KIND, TC, HAS_ISC, MS, OCC, false, EPI>(a, nblocks, lds_floats, s);
}

template <int KIND, int TC, bool HAS_ISC, bool MS, bool EPI = false>
void launch_t(const ConvArgs& a, int nblocks, size_t lds_floats, hipStream_t s, bool fast) {
    // 3 waves/SIMD variant only for the plain 3x3 at the 128-row tile (the only one whose register budget is near 168); the
    // epilogue-stage variants take the 2-wave budget (their extra loads do not fit 168 registers)
    constexpr bool CAN3 = ((KIND == TE_CONV_3X3 && TC == 0 && !MS) || (KIND == TE_CONV_T2 && TC == 1 && !MS));
    if (CAN3 && conv_occ() == 3) launch_o<KIND, TC, HAS_ISC, MS, CAN3 ? 3 : 2, EPI>(a, nblocks, lds_floats, s, fast);
    else launch_o<KIND, TC, HAS_ISC, MS, 2, EPI>(a, nblocks, lds_floats, s, fast);
}

// regions: list of {ri0, rj0, rh, rw}
template <int KIND, int TC>
int launch_regions_tc(ConvArgs a, const int (*regions)[4], int n, hipStream_t s) {
    constexpr int KC = Cfg<KIND, TC>::KC;
    int nblocks = 0;
    size_t lds_floats = 0;
    bool ms = false;
    // FAST kernels exist for the 128-row tile class of the 3x3 family with one sample per tile (the layers that carry the
    // FLOPs); they need whole stages (K and the split-K chunk multiples of the stage depth).  Their input-tile layout
    // differs, so the regions are laid out twice when the first pass shows that the launch qualifies.
    bool fast = false;
    for (int pass = 0; pass < 2; ++pass) {
        nblocks = 0; lds_floats = 0; ms = false; a.nreg = 0;
        for (int i = 0; i < n; ++i) {
            const int rc = add_region<KIND, TC>(a, regions[i][0], regions[i][1], regions[i][2], regions[i][3], nblocks, lds_floats, ms,
                                            fast || (i > 0 && a.nreg > 0 && a.reg[0].NSv == 1), fast);
            if (rc) return rc;
        }
        if (pass == 1 || nblocks == 0) break;
        fast = have_fast<KIND, TC>() && !ms && a.K == a.Kp && a.K % KC == 0 && (a.ksplit == 1 || a.kchunk % KC == 0) &&
               a.reg[0].NS == 1 && a.reg[0].TW >= 4;      // TW >= 4: row segments never straddle the left image border
        if (!fast) break;
    }
    if (nblocks == 0) return 0;
    if (TC == 3 && !fast) return launch_regions_tc<KIND, 0>(a, regions, n, s);      // the deep-stage class exists as a FAST kernel only
    if (a.res || a.mref) {
        if constexpr (KIND == TE_CONV_3X3 || KIND == TE_CONV_1X1) {
            if (a.isc) return te::fail(TE_ERR_UNSUPPORTED, "te_conv_res_f32: residual / mask epilogue exists for unmodulated launches only");
            if (ms) launch_t<KIND, TC, false, true, true>(a, nblocks, lds_floats, s, fast);
            else launch_t<KIND, TC, false, false, true>(a, nblocks, lds_floats, s, fast);
            return 0;
        } else {
            return te::fail(TE_ERR_UNSUPPORTED, "te_conv_res_f32: residual / mask epilogue exists for the 3x3 and 1x1 kinds only");
        }
    }
    if (!a.isc) launch_t<KIND, TC, false, false>(a, nblocks, lds_floats, s, fast);
    else if (ms) launch_t<KIND, TC, true, true>(a, nblocks, lds_floats, s, fast);
    else launch_t<KIND, TC, true, false>(a, nblocks, lds_floats, s, fast);
    return 0;
}

inline int tile_class(int M) { return M >= 96 ? 0 : (M >= 48 ? 1 : 2); }

template <int KIND>
int launch_regions(const ConvArgs& a, const int (*regions)[4], int n, hipStream_t s, int tc) {
    if constexpr (KIND == TE_CONV_1X1) {
        if (tc == 3) return launch_regions_tc<KIND, 3>(a, regions, n, s);
    }
    switch (tc) {
        case 0: return launch_regions_tc<KIND, 0>(a, regions, n, s);
        case 1: return launch_regions_tc<KIND, 1>(a, regions, n, s);
        default: return launch_regions_tc<KIND, 2>(a, regions, n, s);
    }
}

}  // namespace

#if defined(TE_CONV_PROF) || defined(TE_CONV_PROF2)
extern "C" int te_debug_conv_prof(void* host_dst, int64_t bytes) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(te_conv_prof_buf), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
extern "C" int te_debug_conv_prof_clear() {
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(te_conv_prof_buf)) != hipSuccess) return -1;
    return (int)hipMemset(dptr, 0, sizeof(te_conv_prof_buf));
}
#endif

extern "C" int64_t te_conv_packed_numel(int kind_pack, int Co, int Ci, int ksize) {
    const PackDims d = pack_dims(kind_pack, Co, Ci, ksize);
    // (the split layouts: 3 pieces x 12 x K x M bf16 = 18 K M floats)
    if (pack_is_s6(kind_pack)) return ((int64_t)27 * d.Kp * d.Mp + 1) / 2;          // 3 pieces x 9 taps x K x M bf16
    if (pack_is_p6(kind_pack)) return ((int64_t)3 * d.Kp * d.Mp + 1) / 2;           // 3 pieces x K x M bf16
    if (pack_is_t6(kind_pack)) return t6_plain_offset(d.K, d.M) + (int64_t)d.ntap * d.Kp * d.Mp;
    return (int64_t)(pack_is_wino6(kind_pack) ? 18 : (pack_is_wino(kind_pack) ? 12 : d.ntap)) * d.Kp * d.Mp;
}

static int pack_launch(const char* what, int n, float* const* wp, const float* const* w, const float* wscale, const int* kind_pack,
                       const int* Co, const int* Ci, const int* ksize, hipStream_t s);

extern "C" int te_conv_pack_weights_f32(float* wp, const float* w, float wscale, int kind_pack, int Co, int Ci, int ksize,
                                        te_stream_t stream_) {
    return pack_launch("te_conv_pack_weights_f32", 1, &wp, &w, &wscale, &kind_pack, &Co, &Ci, &ksize, (hipStream_t)stream_);
}

extern "C" int te_conv_pack_weights2_f32(float* wp_a, int kind_a, float* wp_b, int kind_b, const float* w, float wscale, int Co,
                                         int Ci, int ksize, te_stream_t stream_) {
    float* wp[2] = {wp_a, wp_b};
    const float* ws[2] = {w, w};
    const float sc[2] = {wscale, wscale};
    const int kinds[2] = {kind_a, kind_b}, co[2] = {Co, Co}, ci[2] = {Ci, Ci}, ks[2] = {ksize, ksize};
    return pack_launch("te_conv_pack_weights2_f32", 2, wp, ws, sc, kinds, co, ci, ks, (hipStream_t)stream_);
}

extern "C" int te_conv_pack_weights_multi_f32(int n, float* const* wp, const float* const* w, const float* wscale,
                                              const int* kind_pack, const int* Co, const int* Ci, const int* ksize,
                                              te_stream_t stream_) {
    return pack_launch("te_conv_pack_weights_multi_f32", n, wp, w, wscale, kind_pack, Co, Ci, ksize, (hipStream_t)stream_);
}

// every packing entry point runs the tiled multi-job kernel (1, 2 or up to 64 layouts per launch)
static int pack_launch(const char* what, int n, float* const* wp, const float* const* w, const float* wscale, const int* kind_pack,
                       const int* Co, const int* Ci, const int* ksize, hipStream_t s) {
    TE_REQUIRE(n >= 0 && (n == 0 || (wp && w && wscale && kind_pack && Co && Ci && ksize)), TE_ERR_NULL, "%s: NULL table", what);
    for (int base = 0; base < n; base += 64) {
        const int cnt = std::min(64, n - base);
        PackJobs P{};
        int64_t biggest = 0;
        for (int i = 0; i < cnt; ++i) {
            const int e = base + i;
            TE_REQUIRE(wp[e] && w[e], TE_ERR_NULL, "%s: NULL pointer in job %d", what, e);
            TE_REQUIRE(Co[e] > 0 && Ci[e] > 0 && (ksize[e] == 1 || ksize[e] == 3), TE_ERR_SHAPE, "%s: bad dims in job %d", what, e);
            TE_REQUIRE(kind_pack[e] >= 0 && kind_pack[e] <= 12, TE_ERR_UNSUPPORTED, "%s: bad kind in job %d", what, e);
            TE_REQUIRE(!pack_is_p6(kind_pack[e]) || (ksize[e] == 1 && (kind_pack[e] == TE_PACK_P6FWD ? (Co[e] % 32 == 0 && Ci[e] % 16 == 0)
                                                                                                    : (Ci[e] % 32 == 0 && Co[e] % 16 == 0))),
                       TE_ERR_UNSUPPORTED, "te_conv_pack_weights: the split layouts of TE_CONV_1X1S6 need a 1x1 weight with M %% 32 == 0 and K %% 16 == 0");
            TE_REQUIRE(!pack_is_t6(kind_pack[e]) || (ksize[e] == 3 && (kind_pack[e] == TE_PACK_T6FWD ? (Co[e] % 32 == 0 && Ci[e] % 16 == 0)
                                                                                                    : (Ci[e] % 32 == 0 && Co[e] % 16 == 0))),
                       TE_ERR_UNSUPPORTED, "te_conv_pack_weights: the split layouts of TE_CONV_T2S6 need a 3x3 weight with M %% 32 == 0 and K %% 16 == 0");
            TE_REQUIRE(!pack_is_s6(kind_pack[e]) || (ksize[e] == 3 && (kind_pack[e] == TE_PACK_S6FWD ? (Co[e] % 32 == 0 && Ci[e] % 16 == 0)
                                                                                                    : (Ci[e] % 32 == 0 && Co[e] % 16 == 0))),
                       TE_ERR_UNSUPPORTED, "te_conv_pack_weights: the split layouts of TE_CONV_S2S6 need a 3x3 weight with M %% 32 == 0 and K %% 16 == 0");
            TE_REQUIRE(!pack_is_wino6(kind_pack[e]) || (ksize[e] == 3 && Co[e] % 32 == 0 && Ci[e] % 32 == 0), TE_ERR_UNSUPPORTED,
                       "%s: the split Winograd layouts need 3x3 taps and channel counts that are multiples of 32 (job %d)", what, e);
            TE_REQUIRE(!pack_is_wino(kind_pack[e]) || (ksize[e] == 3 && Co[e] % 8 == 0 && Ci[e] % 8 == 0), TE_ERR_UNSUPPORTED,
                       "%s: the Winograd layouts need 3x3 taps and channel counts that are multiples of 8 (job %d)", what, e);
            const PackDims d = pack_dims(kind_pack[e], Co[e], Ci[e], ksize[e]);
            P.j[i] = PackJob{wp[e], w[e], wscale[e], kind_pack[e], Ci[e], d.ntap, d.K, d.M, d.Kp, d.Mp};
            biggest = std::max<int64_t>(biggest, te::cdiv(d.Kp, PT) * te::cdiv(d.Mp, PT));
        }
        // one block per 32 x 32 tile of the biggest job (512 x 512: 256 blocks); small jobs walk their few tiles and exit
        dim3 grid((unsigned)std::min<int64_t>(biggest, 256), (unsigned)cnt);
        pack_weights_multi_kernel<<<grid, 256, 0, s>>>(P);
    }
    return te::launch_status(what);
}

// transposed conv on images up to this many cells per side runs as ONE padded (H+1) x (W+1) region; larger ones as body +
// last column + last row (build knob for A/B measurements; round 3: 16 -> 8, the 8x8 -> 17x17 layer filled 42 % of its
// padded tiles: 152 -> 106 us at 512 -> 512, batch 16)
#ifndef TE_T2_PAD_LIMIT
#define TE_T2_PAD_LIMIT 8
#endif

#ifndef TE_SPLITK_MINSTAGES      // build knobs of the split-K plan: fewest stages a split keeps, blocks per CU it aims at
#define TE_SPLITK_MINSTAGES 4
#endif
#ifndef TE_SPLITK_BLOCKS
#define TE_SPLITK_BLOCKS 2
#endif

// tile class + split-K plan of a launch (shared by te_conv_splitk_count and the launch itself)
struct ConvPlan { int tc, ksplit, kchunk; };
static ConvPlan conv_plan(int kind, int B, int K, int M, int H, int W) {
    ConvPlan pl;
    const int Kp = roundup(K, KPAD);
    const bool t2k = (kind == TE_CONV_T2);
    int tc = tile_class(M);
    // transposed conv on small images (<= 1024 of the 64 x 128 tiles): the 64 x 64 tile class fits 3 waves per SIMD and
    // gives the chip more, shorter blocks (512->512 @32: 87 -> 100 TFLOP/s, @16: 57 -> 83)
#ifndef TE_T2_TC1_LIMIT
#define TE_T2_TC1_LIMIT (1024 * 128)
#endif
    if (t2k && tc == 0 && (int64_t)B * H * W * te::cdiv(M, 64) <= (int64_t)TE_T2_TC1_LIMIT) tc = 1;
    // (narrower tile classes for the 4x4 / 8x8 layers were measured and lose: 3x3 512->512 @4x4 65 -> 112 (64 rows) / 119 us (32 rows),
    // profiles/experiments/r03_tiny_tile_class.log)
    pl.tc = tc;
    const int KC = t2k ? (tc == 0 ? T2KC0 : 16) : 8;
    const int BM = t2k ? (tc == 2 ? 32 : 64) : (tc == 0 ? 128 : (tc == 1 ? 64 : 32));
    // split the channel loop when the image is too small to give every CU a tile (4x4 ... 16x16 layers); the
    // split count comes from the real tile geometry of the main region and is shared by every region of the launch
    const int ntile = t2k ? (tc == 1 ? 64 : 128) : ((tc == 0 || (kind == TE_CONV_S2 && tc == 1)) ? 128 : 256);      // cells per block tile of the chosen tile class
    int rh = H, rw = W;
    if (t2k && (W + 1 <= TE_T2_PAD_LIMIT || H + 1 <= TE_T2_PAD_LIMIT)) { rh = H + 1; rw = W + 1; }
    const int TW = std::min(32, pow2ceil(rw)), TH = std::min(pow2ceil(rh), ntile / TW), NS = ntile / (TW * TH);
    const int64_t base_blocks = (int64_t)te::cdiv(rw, TW) * te::cdiv(rh, TH) * te::cdiv(B, NS) * te::cdiv(M, BM);
    const int stages = Kp / KC;
    int ks = 1;
    if (base_blocks < te::kNumCU) ks = (int)std::min<int64_t>(te::cdiv(TE_SPLITK_BLOCKS * te::kNumCU, base_blocks), std::max(1, stages / TE_SPLITK_MINSTAGES));
    pl.ksplit = std::max(1, ks);
    pl.kchunk = (int)te::cdiv(stages, pl.ksplit) * KC;
    pl.ksplit = (int)te::cdiv(Kp, pl.kchunk);
    // 1x1 on images that fill the chip, channel count a multiple of 64: the deep-stage FAST class
    if (TE_CONV_FAST && TE_CONV_DEEP1X1 && kind == TE_CONV_1X1 && tc == 0 && pl.ksplit == 1 && NS == 1 && K % 64 == 0) pl.tc = 3;
    return pl;
}

// a non-blocking side stream + fork / join events per (host thread, device): the autograd engine's backward threads call the ABI
// concurrently, each gets its own.  Created on first use, never destroyed (they live as long as the process).
struct SideStream { hipStream_t stream; hipEvent_t fork, join; bool ok; };
static SideStream* side_stream() {
    static const bool enabled = getenv("TE_T2_SIDE_STREAM") && atoi(getenv("TE_T2_SIDE_STREAM")) != 0;      // OFF by default, see te_conv_res_f32
    if (!enabled) return nullptr;
    constexpr int MAXDEV = 16;
    static thread_local SideStream tab[MAXDEV] = {};
    static thread_local bool tried[MAXDEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    if (!tried[dev]) {
        tried[dev] = true;
        SideStream& t = tab[dev];
        t.ok = hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&t.join, hipEventDisableTiming) == hipSuccess;
    }
    return tab[dev].ok ? &tab[dev] : nullptr;
}

extern "C" int64_t te_conv_t2s6_ws_floats(int B, int K, int H) { return (B > 0 && K > 0 && H > 0) ? (int64_t)B * K * H : TE_ERR_SHAPE; }

extern "C" int te_conv_splitk_count(int kind, int B, int K, int M, int H, int W) {
    if (B <= 0 || K <= 0 || M <= 0 || H <= 0 || W <= 0 || kind < 0 || kind > 8) return TE_ERR_SHAPE;
    if (kind == TE_CONV_3X3W || kind == TE_CONV_3X3W6 || kind == TE_CONV_S2S6 || kind == TE_CONV_T2S6 || kind == TE_CONV_1X1S6) return 1;
    return conv_plan(kind, B, K, M, H, W).ksplit;
}

extern "C" int te_conv_res_f32(float* out, float* ws, const float* in, const float* wp, const float* isc, const float* osc,
                               const float* bias, const float* res, const float* mask_ref, float mask_gain, int act, int kind,
                               int B, int K, int M, int H, int W, te_stream_t stream_) {
    TE_REQUIRE(out && in && wp, TE_ERR_NULL, "te_conv_f32: out/in/wp is NULL");
    TE_REQUIRE(!(res || mask_ref) || (kind != TE_CONV_T2 && kind != TE_CONV_T2S6), TE_ERR_UNSUPPORTED,
               "te_conv_res_f32: no residual / mask epilogue for the transposed kind");
    TE_REQUIRE(B > 0 && K > 0 && M > 0 && H > 0 && W > 0, TE_ERR_SHAPE, "te_conv_f32: bad dims");
    TE_REQUIRE(act == 0 || act == 3 || act == 4, TE_ERR_UNSUPPORTED, "te_conv_f32: act must be 0, 3 or 4");
    TE_REQUIRE(kind >= 0 && kind <= 8, TE_ERR_UNSUPPORTED, "te_conv_f32: unknown kind %d", kind);
    hipStream_t s = (hipStream_t)stream_;
    if (kind == TE_CONV_1X1S6) {
        TE_REQUIRE(!isc && !osc && !bias && !mask_ref && act == 0, TE_ERR_UNSUPPORTED,
                   "te_conv_f32(TE_CONV_1X1S6): plain product + residual only (no scales, bias, activation or mask)");
        return te_p1s6_launch(out, in, wp, res, B, K, M, H, W, s);
    }
    if (kind == TE_CONV_3X3W) return te_wino_launch(out, in, wp, isc, osc, bias, res, mask_ref, mask_gain, act, B, K, M, H, W, s);
    if (kind == TE_CONV_3X3W6) return te_wino6_launch(out, in, wp, isc, osc, bias, res, mask_ref, mask_gain, act, B, K, M, H, W, s);
    if (kind == TE_CONV_S2S6) return te_s2s6_launch(out, in, wp, isc, osc, bias, res, mask_ref, mask_gain, act, B, K, M, H, W, s);
    ConvArgs a{};
    a.out = out; a.ws = ws; a.in = in; a.wp = wp; a.isc = isc; a.osc = osc; a.bias = bias; a.res = res; a.mref = mask_ref; a.mgain = mask_gain; a.act = act;
    a.B = B; a.K = K; a.M = M; a.Kp = roundup(K, KPAD); a.Mp = roundup(M, MPAD); a.H = H; a.W = W;
    if (kind == TE_CONV_T2S6) {
        // body cells on the bf16 pipe; the last output row and column (cells i = H, j = W) from the plain copy of the weights behind the
        // split layout (TE_PACK_T6FWD / TE_PACK_T6SWAP): since round 6 by t2_edge_kernel (below); under TE_T2_EDGE=0 as two thin regions of
        // the fp32 kernel, the round-5 path this comment describes.  The thin launch is a few dozen
        // blocks that each walk the whole channel loop (130 - 150 us of latency for < 1 GFLOP: 28 % of a 128-channel launch,
        // tools/block_overhead_probe.py); it writes other output pixels than the body, so it runs on a SIDE stream of this host thread,
        // forked from and joined back into the caller's stream with events (also legal inside a stream capture), concurrently with
        // the body kernel - OPT-IN (TE_T2_SIDE_STREAM=1).  Measured (round 5, same box, alternating runs): the launch alone gains 3 - 11 %
        // (168 / 172 / 172 / 97 against 163 / 164 / 155 / 89 TFLOP/s at the four large shapes), the training iteration LOSES 2 % (118.5 /
        // 119.6 against 121.4 / 122.1 img/s: three more runtime calls per launch on a host thread that is already the pacemaker of the
        // short kernels around it).  Default: both launches on the caller's stream, one after the other.
        a.wp = wp + t6_plain_offset(K, M);
        a.ws = nullptr; a.ksplit = 1; a.kchunk = a.Kp;
        a.Hi = H; a.Wi = W; a.Ho = 2 * H + 1; a.Wo = 2 * W + 1;
        const int r[2][4] = {{0, W, H + 1, 1}, {H, 0, 1, W}};
        const int tc = conv_plan(TE_CONV_T2, B, K, M, H, W).tc;
        // Round 6: the two lines as ONE small vector-ALU launch (t2s6.hip, t2_edge_kernel: 15 - 30 us instead of 130 - 150); TE_T2_EDGE=0
        // brings the thin regions of the fp32 kernel back (A/B measurements, tests).
        static const bool edge_kernel = [] { const char* e = getenv("TE_T2_EDGE"); return !e || atoi(e) != 0; }();
        if (edge_kernel) {
            // `ws` (optional for this kind: te_conv_t2s6_ws_floats = B K H floats) takes the last input column from the body kernel
            // to the edge kernel; without it the edge kernel gathers the column itself (one cache line per element: slower)
            int rc = te_t2s6_launch(out, in, wp, isc, osc, bias, act, B, K, M, H, W, s, ws);
            if (rc) return rc;
            return te_t2s6_edge_launch(out, in, a.wp, isc, osc, bias, act, B, K, M, H, W, a.Kp, a.Mp, s, ws);
        }
        SideStream* side = side_stream();
        int rc;
        if (side && hipEventRecord(side->fork, s) == hipSuccess && hipStreamWaitEvent(side->stream, side->fork, 0) == hipSuccess) {
            rc = launch_regions<TE_CONV_T2>(a, r, 2, side->stream, tc);
            const hipError_t e1 = hipEventRecord(side->join, side->stream);
            const int rc2 = te_t2s6_launch(out, in, wp, isc, osc, bias, act, B, K, M, H, W, s);
            const hipError_t e2 = hipStreamWaitEvent(s, side->join, 0);       // (join even if a launch failed: the stream stays consistent)
            if (rc) return rc;
            if (rc2) return rc2;
            if (e1 != hipSuccess || e2 != hipSuccess) return te::fail((int)(e1 != hipSuccess ? e1 : e2), "te_conv_f32(TE_CONV_T2S6): side-stream join failed");
        } else {
            rc = te_t2s6_launch(out, in, wp, isc, osc, bias, act, B, K, M, H, W, s);
            if (rc) return rc;
            rc = launch_regions<TE_CONV_T2>(a, r, 2, s, tc);
            if (rc) return rc;
        }
        return te::launch_status("te_conv_f32(TE_CONV_T2S6)");
    }
    const ConvPlan pl = conv_plan(kind, B, K, M, H, W);
    const int tc = pl.tc;
    a.ksplit = pl.ksplit; a.kchunk = pl.kchunk;
    if (a.ksplit > 1 && !a.ws) {       // no workspace (te_conv_f32): the channel loop is not split - slower on 4x4 ... 16x16 images,
        a.ksplit = 1;                  // but no atomics anywhere: every result of this library is bit-reproducible
        a.kchunk = a.Kp;
    }
    if (a.ksplit == 1) a.ws = nullptr;
    int rc = 0;
    switch (kind) {
        case TE_CONV_3X3: {
            a.Hi = a.Ho = H; a.Wi = a.Wo = W;
            const int r[1][4] = {{0, 0, H, W}};
            // (a 128 x 256 block tile, NBW = 4, measured the same 133 TF/s: the kernel sits at the clock-limited peak)
            rc = launch_regions<TE_CONV_3X3>(a, r, 1, s, tc);
        } break;
        case TE_CONV_1X1: {
            a.Hi = a.Ho = H; a.Wi = a.Wo = W;
            const int r[1][4] = {{0, 0, H, W}};
            rc = launch_regions<TE_CONV_1X1>(a, r, 1, s, tc);
        } break;
        case TE_CONV_S2: {
            a.Hi = 2 * H + 1; a.Wi = 2 * W + 1; a.Ho = H; a.Wo = W;
            const int r[1][4] = {{0, 0, H, W}};
            rc = launch_regions<TE_CONV_S2>(a, r, 1, s, tc);
        } break;
        default: {      // TE_CONV_T2
            a.Hi = H; a.Wi = W; a.Ho = 2 * H + 1; a.Wo = 2 * W + 1;
            if (W + 1 <= TE_T2_PAD_LIMIT || H + 1 <= TE_T2_PAD_LIMIT) {
                const int r[1][4] = {{0, 0, H + 1, W + 1}};                       // small images: one padded region
                rc = launch_regions<TE_CONV_T2>(a, r, 1, s, tc);
            } else {
                const int r[3][4] = {{0, 0, H, W}, {0, W, H + 1, 1}, {H, 0, 1, W}};  // body, last column (+corner), last row
                rc = launch_regions<TE_CONV_T2>(a, r, 3, s, tc);
            }
        } break;
    }
    if (rc) return rc;
    if (a.ksplit > 1 && (a.ws || osc || bias || act)) {
        const int plane = a.Ho * a.Wo;
        const int64_t total = (int64_t)B * M * plane;
        conv_finalize_kernel<<<(int)std::min<int64_t>(te::cdiv(total, 256), te::kNumCU * 8), 256, 0, s>>>(out, a.ws, a.ksplit, osc, bias, res,
                                                                                                     mask_ref, mask_gain, act, M, plane, total);
    }
    return te::launch_status("te_conv_f32");
}

extern "C" int te_conv_ws_f32(float* out, float* ws, const float* in, const float* wp, const float* isc, const float* osc,
                              const float* bias, int act, int kind, int B, int K, int M, int H, int W, te_stream_t stream_) {
    return te_conv_res_f32(out, ws, in, wp, isc, osc, bias, nullptr, nullptr, 1.f, act, kind, B, K, M, H, W, stream_);
}

extern "C" int te_conv_f32(float* out, const float* in, const float* wp, const float* isc, const float* osc,
                           const float* bias, int act, int kind, int B, int K, int M, int H, int W, te_stream_t stream_) {
    return te_conv_ws_f32(out, nullptr, in, wp, isc, osc, bias, act, kind, B, K, M, H, W, stream_);
}
