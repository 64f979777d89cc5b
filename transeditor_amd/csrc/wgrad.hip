// F1 (backward): weight-gradient correlation of the modulated-convolution family on the fp32 matrix pipe,
// plus the reduction that turns the per-sample correlation slabs into dW, d(style scale) and d(demod scale).
//
//   slab[b][s][co][ci][tap] = sum_{cells of chunk s of sample b}  g[b,co, gpos(cell,tap)] * x[b,ci, xpos(cell,tap)]
//
//   3X3 / 1X1 : cells = output pixels p;   g at p (plain operand),           x at p + tap - pad (shifted operand)
//   T2        : cells = low-res pixels;    g at (2i+ky, 2j+kx) (shifted),    x at (i,j) (plain)
//
// GEMM view: rows = co (A operand = g), cols = ci (B operand = x), reduction = cells, one 32x32 accumulator
// per tap -> each wave owns 1 x 1 x NTAP tiles (144 accumulator registers for 3x3); a block of 4 waves covers
// 64 co x 64 ci.  The reduction over cells is split over blocks (per sample b, S chunks per sample) and every
// block writes its own slab: no atomics, deterministic.  Keeping the slabs PER SAMPLE with NO modulation
// applied is what lets one pass over the activations produce all three gradients (te_wgrad_reduce_f32):
//   dW[co,ci,t]  = sum_b osc[b,co] isc[b,ci] slab_b      d isc[b,ci] = sum_{co,t} W osc[b,co] slab_b
//   d osc[b,co]  = sum_{ci,t} W isc[b,ci] slab_b
// LDS tiles are [channel][position] with an ODD channel stride, so the 32 lanes of an MFMA operand read
// (32 different channels, same position) hit 32 different banks.
#include "te_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NTHREADS = 256;
constexpr int CB = 64;   // channels per block tile on each side

struct WgArgs {
    float* slabs;
    const float* g;
    const float* x;
    int B, Co, Ci, H, W;     // low-res (cell grid) size
    int Hg, Wg, Hx, Wx;      // spatial size of g and x
    int S;                   // chunks per sample
    int TH, TW, lgTW, NC;    // cell tile (NC = TH*TW cells per stage)
    int tiles_x, tiles_y;
    int QH, QW, QS;          // shifted-operand tile: rows, cols, channel stride (odd)
    int PS;                  // plain-operand channel stride (odd)
};

template <int KIND> struct WK;
template <> struct WK<TE_CONV_3X3> { static constexpr int NT = 9, NCELL = 64, NP = 16, NQ = 36; };
template <> struct WK<TE_CONV_1X1> { static constexpr int NT = 1, NCELL = 64, NP = 16, NQ = 16; };
template <> struct WK<TE_CONV_T2>  { static constexpr int NT = 9, NCELL = 32, NP = 8,  NQ = 52; };

template <int KIND>
__global__ __launch_bounds__(NTHREADS) void wgrad_mfma_kernel(const WgArgs p) {
    constexpr int NT = WK<KIND>::NT;
    constexpr bool GSHIFT = (KIND == TE_CONV_T2);   // which operand carries the tap shift
    constexpr int NP = WK<KIND>::NP;                // plain-tile elements per thread   (CB*NC / 256)
    constexpr int NQ = WK<KIND>::NQ;                // shifted-tile elements per thread (>= CB*QH*QW / 256)

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* pl = smem;                     // plain operand   [CB][PS]
    float* ql = smem + CB * p.PS;         // shifted operand [CB][QS]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wco = wid >> 1, wci = wid & 1;       // wave -> (co block, ci block)

    const int s_chunk = blockIdx.x % p.S, b = blockIdx.x / p.S;
    const int co0 = blockIdx.y * CB, ci0 = blockIdx.z * CB;

    const float* gP = p.g + (size_t)b * p.Co * p.Hg * p.Wg;
    const float* xP = p.x + (size_t)b * p.Ci * p.Hx * p.Wx;
    // plain / shifted operand descriptors
    const float* pbase = GSHIFT ? xP : gP;
    const float* qbase = GSHIFT ? gP : xP;
    const int pC = GSHIFT ? p.Ci : p.Co, qC = GSHIFT ? p.Co : p.Ci;
    const int pc0 = GSHIFT ? ci0 : co0, qc0 = GSHIFT ? co0 : ci0;
    const int pH = GSHIFT ? p.Hx : p.Hg, pW = GSHIFT ? p.Wx : p.Wg;
    const int qH = GSHIFT ? p.Hg : p.Hx, qW = GSHIFT ? p.Wg : p.Wx;

    // per-thread staging descriptors that do not depend on the tile
    // (packed: channel << 20 | row << 10 | col; channel 127 = no element for this thread)
    int pd[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        const int e = tid + NTHREADS * r;
        int ch = e / p.NC;
        const int cell = e - ch * p.NC;
        if (ch >= CB) ch = 127;
        pd[r] = (ch << 20) | ((cell >> p.lgTW) << 10) | (cell & (p.TW - 1));
    }
    const int q_tile = p.QH * p.QW;
    int qd[NQ];
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
        const int e = tid + NTHREADS * r;
        int ch = e / q_tile;
        const int rem = e - ch * q_tile;
        if (ch >= CB) ch = 127;
        const int ry = rem / p.QW;
        qd[r] = (ch << 20) | (ry << 10) | (rem - ry * p.QW);
    }
#define DCH(d) ((d) >> 20)
#define DY(d) (((d) >> 10) & 1023)
#define DX(d) ((d) & 1023)

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // lane bases inside the LDS tiles
    const int a_ch = (GSHIFT ? wco : wco) * 32 + l31;   // co channel of this lane (A operand rows)
    const int b_ch = wci * 32 + l31;                    // ci channel of this lane (B operand cols)
    // g is the A operand, x the B operand.  plain operand position = cell; shifted = f(cell) + tap offset
    const int g_lane = a_ch * (GSHIFT ? p.QS : p.PS) + (GSHIFT ? 2 * half : half);
    const int x_lane = b_ch * (GSHIFT ? p.PS : p.QS) + half;
    const float* g_l = GSHIFT ? ql : pl;
    const float* x_l = GSHIFT ? pl : ql;

    const int n_tiles = p.tiles_x * p.tiles_y;
    const int t_begin = (int)((int64_t)n_tiles * s_chunk / p.S), t_end = (int)((int64_t)n_tiles * (s_chunk + 1) / p.S);

    float preg[NP], qreg[NQ];
    for (int tl = t_begin - 1; tl < t_end; ++tl) {
        if (tl >= t_begin) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < NP; ++r)
                if (DCH(pd[r]) < CB) pl[DCH(pd[r]) * p.PS + (DY(pd[r]) << p.lgTW) + DX(pd[r])] = preg[r];
#pragma unroll
            for (int r = 0; r < NQ; ++r)
                if (DCH(qd[r]) < CB) ql[DCH(qd[r]) * p.QS + DY(qd[r]) * p.QW + DX(qd[r])] = qreg[r];
            __syncthreads();
        }
        const int tn = tl + 1;
        if (tn < t_end) {
            const int ty0 = (tn / p.tiles_x) * p.TH, tx0 = (tn % p.tiles_x) * p.TW;   // first cell of the tile
            int qy0, qx0;
            if (KIND == TE_CONV_3X3) { qy0 = ty0 - 1; qx0 = tx0 - 1; }
            else if (KIND == TE_CONV_T2) { qy0 = 2 * ty0; qx0 = 2 * tx0; }
            else { qy0 = ty0; qx0 = tx0; }
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int y = ty0 + DY(pd[r]), xx = tx0 + DX(pd[r]), ch = pc0 + DCH(pd[r]);
                const bool ok = DCH(pd[r]) < CB && y < p.H && xx < p.W && ch < pC;    // plain operand lives on the cell grid
                const size_t off = ((size_t)(ok ? ch : 0) * pH + (ok ? y : 0)) * pW + (ok ? xx : 0);
                const float v = pbase[off];
                preg[r] = ok ? v : 0.f;
            }
#pragma unroll
            for (int r = 0; r < NQ; ++r) {
                const int y = qy0 + DY(qd[r]), xx = qx0 + DX(qd[r]), ch = qc0 + DCH(qd[r]);
                const bool ok = DCH(qd[r]) < CB && y >= 0 && y < qH && xx >= 0 && xx < qW && ch < qC;
                const size_t off = ((size_t)(ok ? ch : 0) * qH + (ok ? y : 0)) * qW + (ok ? xx : 0);
                const float v = qbase[off];
                qreg[r] = ok ? v : 0.f;
            }
        }
        if (tl < t_begin) continue;

        // ---- MFMAs over the cells of the staged tile (2 cells per k-step: lane half h takes cell 2*ks + h)
#pragma unroll 2
        for (int ks = 0; ks < p.NC / 2; ++ks) {
            const int c0 = 2 * ks;
            const int cy = c0 >> p.lgTW, cx = c0 & (p.TW - 1);
            int sh;   // position of cell c0 inside the shifted tile (tap 0,0)
            if (KIND == TE_CONV_T2) sh = 2 * cy * p.QW + 2 * cx;
            else sh = cy * p.QW + cx;
            if (!GSHIFT) {
                const float av = g_l[g_lane + c0];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int toff = (NT == 1) ? 0 : (t / 3) * p.QW + (t % 3);
                    const float bv = x_l[x_lane + sh + toff];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            } else {
                const float bv = x_l[x_lane + c0];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int toff = (t / 3) * p.QW + (t % 3);
                    const float av = g_l[g_lane + sh + toff];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // ---- write the slab tile: rows = co, cols = ci;  slab[b][s][co][ci][tap]
    float* sl = p.slabs + ((size_t)b * p.S + s_chunk) * p.Co * p.Ci * NT;
    const int ci = ci0 + wci * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < p.Co && ci < p.Ci) {
            float* dst = sl + ((size_t)co * p.Ci + ci) * NT;
#pragma unroll
            for (int t = 0; t < NT; ++t) dst[t] = acc[t][r];
        }
    }
}

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline int pow2ceil(int v) { return 1 << ilog2(v); }

template <int KIND>
bool fill_geometry(WgArgs& a) {
    constexpr int NCELL = WK<KIND>::NCELL;
    a.TW = std::min(32, pow2ceil(a.W));
    if (a.TW < 2) return false;
    a.TH = std::max(1, std::min(pow2ceil(a.H), NCELL / a.TW));
    a.NC = NCELL;                                  // cells per stage (positions beyond TH*TW are never valid)
    if (a.TH * a.TW < NCELL) a.NC = a.TH * a.TW;
    a.lgTW = ilog2(a.TW);
    a.tiles_x = (a.W + a.TW - 1) / a.TW;
    a.tiles_y = (a.H + a.TH - 1) / a.TH;
    if (KIND == TE_CONV_3X3) { a.QH = a.TH + 2; a.QW = a.TW + 2; }
    else if (KIND == TE_CONV_T2) { a.QH = 2 * a.TH + 1; a.QW = 2 * a.TW + 1; }
    else { a.QH = a.TH; a.QW = a.TW; }
    a.QS = (a.QH * a.QW) | 1;
    a.PS = a.NC | 1;
    return CB * a.QH * a.QW <= WK<KIND>::NQ * NTHREADS && CB * a.NC <= WK<KIND>::NP * NTHREADS;
}

template <int KIND>
int launch_wgrad(WgArgs a, hipStream_t s) {
    if (!fill_geometry<KIND>(a)) return te::fail(TE_ERR_UNSUPPORTED, "te_wgrad_f32: unsupported image size %dx%d", a.H, a.W);
    // the staging loops cover exactly NP*256 plain elements: NC must make CB*NC == NP*256 or be guarded
    const size_t lds = sizeof(float) * ((size_t)CB * a.PS + (size_t)CB * a.QS);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)wgrad_mfma_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_done = true;
    }
    dim3 grid((unsigned)(a.B * a.S), (unsigned)te::cdiv(a.Co, CB), (unsigned)te::cdiv(a.Ci, CB));
    wgrad_mfma_kernel<KIND><<<grid, NTHREADS, lds, s>>>(a);
    return 0;
}

inline int n_cell_tiles(int kind, int H, int W) {
    const int ncell = (kind == TE_CONV_T2) ? 32 : 64;
    const int TW = std::min(32, pow2ceil(W));
    const int TH = std::max(1, std::min(pow2ceil(H), ncell / TW));
    return ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
}

// ------------------------------------------------------------------------------------------------ reduce
constexpr int COB = 4;     // output channels per block
constexpr int BMAX = 16;   // samples per pass (register array)

template <int NT>
__global__ __launch_bounds__(NTHREADS) void wgrad_reduce_kernel(float* __restrict__ gw, float* __restrict__ gisc,
                                                                float* __restrict__ gosc, const float* __restrict__ slabs,
                                                                const float* __restrict__ w, float wscale,
                                                                const float* __restrict__ isc, const float* __restrict__ osc,
                                                                int B, int S, int Co, int Ci) {
    const int ci = blockIdx.x * NTHREADS + threadIdx.x;
    const bool live = ci < Ci;
    const int cic = live ? ci : Ci - 1;
    const int lane = threadIdx.x & 63;
    const size_t slab_sz = (size_t)Co * Ci * NT;
    for (int bb = 0; bb < B; bb += BMAX) {
        const int nb = min(BMAX, B - bb);
        float pisc[BMAX];
#pragma unroll
        for (int i = 0; i < BMAX; ++i) pisc[i] = 0.f;
        for (int co = blockIdx.y * COB; co < min(Co, (int)(blockIdx.y + 1) * COB); ++co) {
            float wv[NT], gacc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { wv[t] = w[((size_t)co * Ci + cic) * NT + t] * wscale; gacc[t] = 0.f; }
#pragma unroll
            for (int i = 0; i < BMAX; ++i) {
                if (i < nb) {
                    const int b = bb + i;
                    float sl[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) sl[t] = 0.f;
                    for (int s = 0; s < S; ++s) {
                        const float* src = slabs + ((size_t)b * S + s) * slab_sz + ((size_t)co * Ci + cic) * NT;
#pragma unroll
                        for (int t = 0; t < NT; ++t) sl[t] += src[t];
                    }
                    const float os = osc ? osc[(size_t)b * Co + co] : 1.f;
                    const float is = isc ? isc[(size_t)b * Ci + cic] : 1.f;
                    float dot = 0.f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) { gacc[t] += os * is * sl[t]; dot += wv[t] * sl[t]; }
                    pisc[i] += os * dot;
                    if (gosc) {
                        float po = live ? is * dot : 0.f;
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) po += __shfl_down(po, off, 64);
                        if (lane == 0) atomicAdd(gosc + (size_t)b * Co + co, po);
                    }
                }
            }
            if (gw && live) {
                // several passes over b (B > BMAX) accumulate; the first pass overwrites
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float* dst = gw + ((size_t)co * Ci + ci) * NT + t;
                    *dst = (bb == 0 ? 0.f : *dst) + gacc[t] * wscale;
                }
            }
        }
        if (gisc && live) {
#pragma unroll
            for (int i = 0; i < BMAX; ++i)
                if (i < nb) atomicAdd(gisc + (size_t)(bb + i) * Ci + ci, pisc[i]);
        }
    }
}

}  // namespace

extern "C" int te_wgrad_slab_count(int kind, int B, int Co, int Ci, int H, int W) {
    if (B <= 0 || Co <= 0 || Ci <= 0 || H <= 0 || W <= 0) return TE_ERR_SHAPE;
    const int tiles = n_cell_tiles(kind, H, W);
    const int64_t mn = te::cdiv(Co, CB) * te::cdiv(Ci, CB) * (int64_t)B;
    int64_t S = te::cdiv(4 * te::kNumCU, mn);     // aim at >= 4 blocks per CU
    S = std::max<int64_t>(1, std::min<int64_t>(S, tiles));
    return (int)S;
}

extern "C" int te_wgrad_f32(float* slabs, const float* g, const float* x, int kind, int B, int Co, int Ci, int H, int W,
                            int S, te_stream_t stream_) {
    TE_REQUIRE(slabs && g && x, TE_ERR_NULL, "te_wgrad_f32: NULL pointer");
    TE_REQUIRE(B > 0 && Co > 0 && Ci > 0 && H > 0 && W > 0 && S > 0, TE_ERR_SHAPE, "te_wgrad_f32: bad dims");
    TE_REQUIRE((int64_t)B * S <= 0x7FFFFFFF && te::cdiv(Co, CB) <= 65535 && te::cdiv(Ci, CB) <= 65535, TE_ERR_SHAPE,
               "te_wgrad_f32: grid too large");
    WgArgs a{};
    a.slabs = slabs; a.g = g; a.x = x; a.B = B; a.Co = Co; a.Ci = Ci; a.H = H; a.W = W; a.S = S;
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

extern "C" int te_wgrad_reduce_f32(float* gw, float* gisc, float* gosc, const float* slabs, const float* w, float wscale,
                                   const float* isc, const float* osc, int B, int S, int Co, int Ci, int taps,
                                   te_stream_t stream_) {
    TE_REQUIRE(slabs && w, TE_ERR_NULL, "te_wgrad_reduce_f32: NULL pointer");
    TE_REQUIRE(B > 0 && S > 0 && Co > 0 && Ci > 0 && (taps == 1 || taps == 9), TE_ERR_SHAPE, "te_wgrad_reduce_f32: bad dims");
    dim3 grid((unsigned)te::cdiv(Ci, NTHREADS), (unsigned)te::cdiv(Co, COB));
    hipStream_t s = (hipStream_t)stream_;
    if (taps == 9) wgrad_reduce_kernel<9><<<grid, NTHREADS, 0, s>>>(gw, gisc, gosc, slabs, w, wscale, isc, osc, B, S, Co, Ci);
    else wgrad_reduce_kernel<1><<<grid, NTHREADS, 0, s>>>(gw, gisc, gosc, slabs, w, wscale, isc, osc, B, S, Co, Ci);
    return te::launch_status("te_wgrad_reduce_f32");
}
