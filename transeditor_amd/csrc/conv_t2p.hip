// F1 (upsampling layers): 3x3 transposed convolution, stride 2, split by OUTPUT ROW PARITY into two implicit GEMMs.
//
// Reference: ModulatedConv2d.forward upsample branch, model_spatial_query.py:310-321 (F.conv_transpose2d(stride=2) on B
// materialised weight copies).  Output pixel (2i + a, 2j + b) of "cell" (i, j) only receives the taps with ky = a and
// kx = b (mod 2):
//     out[2i + a, 2j + b] = sum_{ky in T(a), kx in T(b)} W[ky][kx] * x[i - (ky >> 1), j - (kx >> 1)],   T(0) = {0, 2}, T(1) = {1}
// conv.hip's TE_CONV_T2 kernel keeps all four phase accumulators of a cell block in one wave (8 accumulator tiles = 128
// VGPRs -> 256 VGPRs, 2 waves per SIMD, 70 % MFMA utilisation).  Here a BLOCK owns ONE ROW PARITY a: 64 output channels
// x 128 cells, 4 waves as 2 x 2, per wave 2 cell blocks x 2 column phases = 4 accumulator tiles (64 VGPRs) -> 3 waves per
// SIMD, and only the 6 (a = 0) or 3 (a = 1) taps of that parity are staged (same FLOPs in total).  The two column phases
// of a cell are adjacent output pixels, so every lane stores 8 contiguous bytes and a wave writes full 256-byte row
// segments (a single-phase-per-block variant with stride-2 4-byte stores measured 55 TFLOP/s against 98: partial-line
// writes).  The two parity blocks of a tile sit next to each other in the grid so the input tile both read stays on chip.
// Cells of the last output column (j = W) and row (i = H, a = 0 only) form thin extra regions of the same launch.
// Used for images with more than 16 x 16 cells and >= 96 output channels; everything else stays on conv.hip's kernel.
#include "conv_common.h"

namespace {

constexpr int BM = 64, NBW = 2, WN = 2;       // block tile: 64 output channels (2 waves x 32) x 128 cells (2 waves x 2 x 32)
constexpr int MAXREG = 5;

struct T2pArgs {
    float* out;
    const float* in;
    const float* wp;
    const float* isc;
    const float* osc;
    const float* bias;
    int B, K, M, Kp, Mp;
    int Hi, Wi, Ho, Wo;
    int act;
    int main_tiles;          // tiles of one main region (both have the same grid); blocks [0, 2 * main_tiles) interleave the parities
    struct Region {
        int phase;               // row parity a
        int ri0, rj0, rh, rw;    // cell region
        int TH, TW, lgTW;
        int tiles_x, tiles_y;
        int TIH, TIW, CS;        // input tile rows / cols, per-channel LDS stride
        int first_block;
    } reg[MAXREG];
    int nreg;
};

#ifndef T2P_KC0
#define T2P_KC0 16
#define T2P_KC1 16
#endif
template <int A> struct Phase {
    static constexpr int NY = A ? 1 : 2, NT = NY * 3;                  // taps of this row parity
    static constexpr int KC = A ? T2P_KC1 : T2P_KC0;                   // channels per stage
    static constexpr int ky(int t) { return A ? 1 : 2 * (t / 3); }
    static constexpr int kx(int t) { return t % 3; }
};

template <int A, bool HAS_ISC>
__device__ __forceinline__ void t2p_body(const T2pArgs& p, const T2pArgs::Region& g, int t, float* smem) {
    using P = Phase<A>;
    constexpr int NT = P::NT, KC = P::KC;
    constexpr int WSTAGE = NT * KC * BM;
    constexpr int WLDR = WSTAGE / 4 / NTHREADS;
    static_assert(WSTAGE % (4 * NTHREADS) == 0, "weight stage must be a whole number of 16-byte loads per thread");
    float* wl = smem;            // [NT][KC][BM]
    float* xl = smem + WSTAGE;   // [KC][CS]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;

    const int tx_i = t % g.tiles_x; t /= g.tiles_x;
    const int ty_i = t % g.tiles_y;
    const int b = t / g.tiles_y;
    const int m0 = blockIdx.y * BM;
    const int ci0 = g.ri0 + ty_i * g.TH, cj0 = g.rj0 + tx_i * g.TW;
    const int oy = ci0 - 1, ox = cj0 - 1;

    // ---- staging descriptor: one input-tile element per thread (tiles hold <= 256 elements)
    const unsigned plane4 = (unsigned)p.Hi * p.Wi * 4u;
    unsigned goff = OOBH;
    int loff = -1;
    if (tid < g.TIH * g.TIW) {
        const int ry = tid / g.TIW, rx = tid - ry * g.TIW;
        const int gy = oy + ry, gx = ox + rx;
        loff = tid;
        if (gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi) goff = (unsigned)(gy * p.Wi + gx) * 4u;
    }
    const __amdgpu_buffer_rsrc_t irs = make_rsrc(p.in + (size_t)b * p.K * p.Hi * p.Wi, (unsigned)p.K * plane4);

    int boff[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int c = wn * (NBW * 32) + nb * 32 + l31;
        const int ty = c >> g.lgTW, tx = c & (g.TW - 1);
        boff[nb] = (ty < g.TH ? (ty + 1) * g.TIW + tx + 1 : g.TIW + 1) + half * g.CS;     // cells beyond the tile: any valid address
    }
    const int aoff = half * BM + wm * 32 + l31;

    f32x16 acc[NBW][2];          // [cell block][column phase b]
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 wreg[WLDR];
    float xreg[KC];
    const float* iscb = HAS_ISC ? p.isc + (size_t)b * p.K : nullptr;

    for (int k0 = -KC; k0 < p.Kp; k0 += KC) {
        if (k0 >= 0) {
            // style scales of the stage being committed: block-uniform -> scalar loads, in flight across the barrier
            float sreg[HAS_ISC ? KC : 1];
            if (HAS_ISC) {
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) sreg[kk] = iscb[min(k0 + kk, p.K - 1)];
            }
            __syncthreads();              // every wave finished reading the previous stage
#pragma unroll
            for (int r = 0; r < WLDR; ++r) *reinterpret_cast<f32x4*>(wl + (tid + NTHREADS * r) * 4) = wreg[r];
            if (loff >= 0) {
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) xl[kk * g.CS + loff] = HAS_ISC ? xreg[kk] * sreg[kk] : xreg[kk];
            }
            __syncthreads();
        }
        const int kn = k0 + KC;
        if (kn < p.Kp) {
#pragma unroll
            for (int r = 0; r < WLDR; ++r) {
                const int idx = tid + NTHREADS * r;                       // float4 index inside the stage
                const int row = idx / (BM / 4), c4 = idx % (BM / 4);      // row = t * KC + kk
                const int tt = row / KC, kk = row - tt * KC;
                const int tap = P::ky(tt) * 3 + P::kx(tt);
                const int k = min(kn + kk, p.Kp - 1);                     // (rows past Kp meet zero input: any finite value)
                wreg[r] = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)(tap * p.Kp + k) * p.Mp + m0 + c4 * 4));
            }
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                const int k = kn + kk;
                xreg[kk] = buf_load(irs, goff + (k < p.K ? (unsigned)k * plane4 : OOBH));
            }
        }
        if (k0 < 0) continue;

#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int toff = -(P::ky(tt) == 2 ? g.TIW : 0) - (P::kx(tt) == 2 ? 1 : 0);
                const float a = wl[(tt * KC + kk) * BM + aoff];
                constexpr int dummy = 0;
                (void)dummy;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    const float bv = xl[kk * g.CS + boff[nb] + toff];
                    const int j = P::kx(tt) == 1 ? 1 : 0;
                    acc[nb][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[nb][j], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue.  C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const size_t oplane = (size_t)p.Ho * p.Wo;
    const int mbase = m0 + wm * 32 + 4 * half;
    float sc[16], bi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] = 1.f; bi[r] = 0.f; }
    if (p.osc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2);
            sc[r] = p.osc[(size_t)b * p.M + (m < p.M ? m : p.M - 1)];
        }
    }
    if (p.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2);
            bi[r] = p.bias[m < p.M ? m : p.M - 1];
        }
    }
    const float gain = p.act == 3 ? 1.4142135623730951f : 1.f;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int c = wn * (NBW * 32) + nb * 32 + l31;
        const int ty = c >> g.lgTW, tx = c & (g.TW - 1);
        const int ci = ci0 + ty, cj = cj0 + tx;
        const bool cell_ok = ty < g.TH && ci < g.ri0 + g.rh && cj < g.rj0 + g.rw;
        const int X = 2 * cj;
        const size_t opix = (size_t)(2 * ci + A) * p.Wo + X;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2);
            float v0 = acc[nb][0][r] * sc[r] + bi[r];
            float v1 = acc[nb][1][r] * sc[r] + bi[r];
            if (p.act >= 3) {
                v0 = (v0 > 0.f ? v0 : v0 * 0.2f) * gain;
                v1 = (v1 > 0.f ? v1 : v1 * 0.2f) * gain;
            }
            if (cell_ok && m < p.M) {
                // the two column phases of a cell are adjacent outputs: one 8-byte store (rows are 2W+1 wide, so the pair
                // is only 4-byte aligned, which global stores accept)
                float* dst = p.out + ((size_t)b * p.M + m) * oplane + opix;
                if (X + 1 < p.Wo) {
                    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
                    *reinterpret_cast<f32x2u*>(dst) = f32x2u{v0, v1};
                } else {
                    dst[0] = v0;
                }
            }
        }
    }
}

template <bool HAS_ISC>
__global__ __launch_bounds__(NTHREADS, 3) void t2p_kernel(const T2pArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int ridx, t;
    if ((int)blockIdx.x < 2 * p.main_tiles) {          // the two parities of a tile are neighbours in the grid
        ridx = blockIdx.x & 1;
        t = blockIdx.x >> 1;
    } else {
        ridx = 2;
        for (int r = 3; r < p.nreg; ++r)
            if ((int)blockIdx.x >= p.reg[r].first_block) ridx = r;
        t = blockIdx.x - p.reg[ridx].first_block;
    }
    const T2pArgs::Region g = p.reg[ridx];
    if (g.phase == 0) t2p_body<0, HAS_ISC>(p, g, t, smem);
    else t2p_body<1, HAS_ISC>(p, g, t, smem);
}

inline int ilog2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// returns the number of tiles (per sample) of the region
int fill_region(T2pArgs::Region& g, int phase, int ri0, int rj0, int rh, int rw, size_t& lds_floats) {
    g.phase = phase; g.ri0 = ri0; g.rj0 = rj0; g.rh = rh; g.rw = rw;
    g.TW = std::min(32, 1 << ilog2i(rw));
    g.TH = std::min(1 << ilog2i(rh), 128 / g.TW);
    while (g.TH > 1 && (g.TH + 1) * (g.TW + 1) > NTHREADS) g.TH >>= 1;      // one staged element per thread
    g.lgTW = ilog2i(g.TW);
    g.tiles_x = (rw + g.TW - 1) / g.TW;
    g.tiles_y = (rh + g.TH - 1) / g.TH;
    g.TIH = g.TH + 1; g.TIW = g.TW + 1;
    g.CS = g.TIH * g.TIW;
    const int nt = phase ? 3 : 6, kc = phase ? T2P_KC1 : T2P_KC0;
    lds_floats = std::max(lds_floats, (size_t)nt * kc * BM + (size_t)kc * g.CS);
    return g.tiles_x * g.tiles_y;
}

}  // namespace

// in [B,K,H,W] -> out [B,M,2H+1,2W+1]; requires H, W >= 16 (see conv.hip for smaller images)
int te_launch_t2p(float* out, const float* in, const float* wp, const float* isc, const float* osc, const float* bias, int act,
                  int B, int K, int M, int H, int W, hipStream_t s) {
    T2pArgs a{};
    a.out = out; a.in = in; a.wp = wp; a.isc = isc; a.osc = osc; a.bias = bias; a.act = act;
    a.B = B; a.K = K; a.M = M; a.Kp = (K + KPAD - 1) / KPAD * KPAD; a.Mp = (M + MPAD - 1) / MPAD * MPAD;
    a.Hi = H; a.Wi = W; a.Ho = 2 * H + 1; a.Wo = 2 * W + 1;
    if ((int64_t)K * H * W * 4 >= (int64_t)OOBH) return te::fail(TE_ERR_UNSUPPORTED, "te_conv_f32: sample of %d x %dx%d exceeds 1 GiB", K, H, W);
    size_t lds_floats = 0;
    int n = 0;
    for (int ph = 0; ph < 2; ++ph) {                      // main regions: cells [0,H) x [0,W), identical tile grids
        a.main_tiles = fill_region(a.reg[n], ph, 0, 0, H, W, lds_floats) * B;
        a.reg[n].first_block = 0;
        ++n;
    }
    int nblocks = 2 * a.main_tiles;
    // last cell column j = W (rows i <= H - a), and for a = 0 the last cell row i = H (the corner belongs to the column)
    const int thin[3][5] = {{0, 0, W, H + 1, 1}, {1, 0, W, H, 1}, {0, H, 0, 1, W}};
    for (int i = 0; i < 3; ++i) {
        const int tiles = fill_region(a.reg[n], thin[i][0], thin[i][1], thin[i][2], thin[i][3], thin[i][4], lds_floats) * B;
        a.reg[n].first_block = nblocks;
        nblocks += tiles;
        ++n;
    }
    a.nreg = n;
    dim3 grid((unsigned)nblocks, (unsigned)te::cdiv(M, BM));
    const size_t lds = lds_floats * sizeof(float);
    if (isc) {
        static std::atomic<uint64_t> done{0};
        te::allow_big_lds(done, (const void*)t2p_kernel<true>, 64 * 1024);
        t2p_kernel<true><<<grid, NTHREADS, lds, s>>>(a);
    } else {
        static std::atomic<uint64_t> done{0};
        te::allow_big_lds(done, (const void*)t2p_kernel<false>, 64 * 1024);
        t2p_kernel<false><<<grid, NTHREADS, lds, s>>>(a);
    }
    return 0;
}
