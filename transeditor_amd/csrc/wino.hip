// F1w: 3x3 / stride 1 / pad 1 convolution through a 1-D Winograd F(2,3) transform along x, on the fp32 matrix pipe (round 4).
//
// Two adjacent outputs of a row come from FOUR multiplications per (input channel, tap row) instead of six:
//
//   d0..d3 = s[b,ci] * in[b, ci, y, 2j-1 .. 2j+2]      t0 = d0 - d2   t1 = d1 + d2   t2 = d2 - d1   t3 = d1 - d3     (B^T d)
//   U[ky][c][ci][m] = sum_kx G[c][kx] * wscale * w[m][ci][ky][kx]     G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]   (packing)
//   m_c = sum_{ci, ky} U[ky][c][ci][m] * t_c[ci][y + ky - 1]                                                          (12 GEMMs)
//   out[2j] = m0 + m1 + m2      out[2j+1] = m1 - m2 - m3                                                              (A^T m)
//
// so the implicit GEMM issues 12/18 = 2/3 of the MFMAs of the direct form (conv.hip) for the same result (fp32 throughout; the
// transform constants are 0, +-1, +-1/2: relative error ~1e-6, measured against the direct kernel in tests/test_gpu_winograd.py).
// The price: 4 vector-ALU adds per staged pair, 4/3 of the weight bytes per stage, twice the accumulator registers.  The
// convolution launches that carry the FLOPs of the model (>= 32x32 images, channel counts that are multiples of 8 / 128) were
// at 128 - 142 TFLOP/s = 86 % MFMA utilisation at the sustained clock with the direct kernel, i.e. at its ceiling; this
// kernel reaches 145 - 162 "algorithmic" TFLOP/s at 69 % utilisation (profiles/experiments/r04_winograd_ab.log).
//
// Block: 512 threads = 8 waves, output tile 128 (M) x 4 rows x 32 columns (= 64 pairs); wave (wm, wr): 32 output channels x rows
// {2 wr, 2 wr + 1} x 16 pairs x 4 components = 4 accumulator tiles of v_mfma_f32_32x32x2_f32 (64 registers), so FOUR waves
// share a SIMD (two blocks per CU): measured, a SIMD needs several waves issuing MFMAs at the same time - one wave alone streams
// them at ~58 % of the pipe's rate however its operands are prefetched (the ping-pong design of tools/exp/wino_pp.hip: 88 TFLOP/s
// MFMA-equivalent), two waves with 128 accumulator registers each reach 69 %, four waves with 64 each 75 %
// (profiles/experiments/r04_winograd_ab.log).  8 input channels per stage; the loads of stage s + 1 are issued before the MFMAs
// of stage s (register prefetch), transformed and written to the single LDS image after them (two barriers per stage); inside
// the MFMA loop the LDS operands of step i + 1 are read before the MFMA of step i is issued (sched_barrier-pinned).  Block ->
// (tile, M block) is XCD-aware like the direct kernel's: the M blocks of a tile run on one XCD and share the input tile through
// its L2, and an XCD walks a contiguous band of the tile list.  Epilogue = the direct kernel's: out = (act(osc * conv + bias) + res) * slope(mask_ref).
#include "conv_common.h"

namespace {

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WT = 512;                                   // threads per block (8 waves)
constexpr int KC = 8, TW = 32, NP = TW / 2;
// Block tile BM (M) x TH rows x 32 columns; the 8 waves split as (BM / 32 along M) x (TH / 2 row pairs):
//   BM = 128: 4 x 2, TH = 4      BM = 64: 2 x 4, TH = 8      BM = 32: 1 x 8, TH = 16
// (64- and 32-channel layers - the FFHQ-1024 tail - keep eight waves busy by covering more rows per block)
template <int BM> struct Geo {
    static constexpr int NMW = BM / 32, WNR = 8 / NMW, TH = 2 * WNR;
    static constexpr int RS = (TH + 2) * NP;               // floats per (channel, component) plane
    static constexpr int T_FLOATS = KC * 4 * RS;           // transformed input tile  [KC][4][TH + 2][NP]
    static constexpr int U_FLOATS = 3 * 4 * KC * BM;       // transformed weights     [3][4][KC][BM]
    static constexpr int N_ITEMS = KC * RS;                // input items (channel, row, pair) of a stage
    static constexpr int N_IN = (N_ITEMS + WT - 1) / WT;   // ... per thread (the last one bounds-checked)
    static constexpr int N_W4T = U_FLOATS / 4;             // weight float4 of a stage
    static constexpr int N_W4 = (N_W4T + WT - 1) / WT;     // ... per thread (bounds-checked for BM = 32)
    static constexpr int C4 = BM / 4;                      // float4 per weight row
};

struct WinoArgs {
    float* out; const float* in; const float* U; const float* isc; const float* osc; const float* bias; const float* res;
    const float* mref; float mgain; int act;
    int B, K, M, H, W, ntiles, mblocks, tiles_x, tiles_y, nt8;
};

template <int BM>
__global__ __launch_bounds__(WT, 4) void wino3x3_kernel(const WinoArgs p) {
    typedef Geo<BM> G;
    constexpr int TH = G::TH, RS = G::RS, T_FLOATS = G::T_FLOATS, N_ITEMS = G::N_ITEMS, N_IN = G::N_IN, N_W4 = G::N_W4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm = wid / G::WNR, wr = wid % G::WNR;            // wm: 32 output channels each; wr: a pair of rows
    // block -> (cell tile, M block): the j-th block of XCD x takes tile (j / mblocks) * 8 + x and M block j % mblocks
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    // (banded, te::xcd_banded(): XCD x walks tiles [x n / 8, (x + 1) n / 8) so that neighbouring tiles share their halo lines in
    //  one L2: 1671 -> 650 MB read at 128 -> 128 @256^2, profiles/experiments/r04_xcd_band_ab.log; else tile q * 8 + x)
    const int tq = jx / p.mblocks, mb = jx % p.mblocks;
    const int tile = p.nt8 ? (int)(((int64_t)xcd * p.ntiles) >> 3) + tq : tq * 8 + xcd;
    if (tile >= (p.nt8 ? (int)(((int64_t)(xcd + 1) * p.ntiles) >> 3) : p.ntiles)) return;      // (grid padded; block-uniform)
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

    // staging geometry: item e = tid + 256 i -> (channel, row, pair); one element offset and one LDS offset per item
    int g_off[N_IN], l_off[N_IN];
#pragma unroll
    for (int i = 0; i < N_IN; ++i) {
        const int e = tid + WT * i;
        const int pr = e % NP, row = (e / NP) % (TH + 2), ch = e / RS;
        g_off[i] = (ch * p.H + (y0 - 1 + row)) * p.W + x0 + 2 * pr - 1;
        l_off[i] = e < N_ITEMS ? ch * 4 * RS + row * NP + pr : -1;
    }
    // weights: float4 idx = tid + 512 i -> row idx / C4 (of 96), column 4 (idx % C4)
    const size_t stage_in = (size_t)KC * p.H * p.W, stage_w = (size_t)3 * 4 * KC * p.M;
    f32x4 rin[N_IN];
    float rsc[N_IN];
    f32x4 rw[N_W4];
    const int nstage = p.K / KC;
    auto issue = [&](int s) {
        const float* base = inb + s * stage_in;
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            if (l_off[i] < 0) continue;
            const int e = tid + WT * i, ch = e / RS;
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
        const float* us = p.U + s * stage_w + mb * BM;
#pragma unroll
        for (int i = 0; i < N_W4; ++i) {
            const int idx = tid + WT * i;
            if (G::N_W4T % WT == 0 || idx < G::N_W4T)
                rw[i] = *reinterpret_cast<const f32x4*>(us + (size_t)(idx / G::C4) * p.M + 4 * (idx % G::C4));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < N_IN; ++i) {
            if (l_off[i] < 0) continue;
            const float sc = rsc[i];
            const float d0 = rin[i][0] * sc, d1 = rin[i][1] * sc, d2 = rin[i][2] * sc, d3 = rin[i][3] * sc;
            float* dst = smem + l_off[i];
            dst[0] = d0 - d2;
            dst[RS] = d1 + d2;
            dst[2 * RS] = d2 - d1;
            dst[3 * RS] = d1 - d3;
        }
#pragma unroll
        for (int i = 0; i < N_W4; ++i) {
            const int idx = tid + WT * i;
            if (G::N_W4T % WT == 0 || idx < G::N_W4T) *reinterpret_cast<f32x4*>(smem + T_FLOATS + 4 * idx) = rw[i];
        }
    };
    const int rr = l31 >> 4, jj = l31 & 15;
    const int b_off = half * 4 * RS + (2 * wr + rr) * NP + jj;          // + (2 ks * 4 + c) * RS + ky * NP
    const int a_off = T_FLOATS + half * BM + wm * 32 + l31;             // + ((ky * 4 + c) * KC + 2 ks) * BM

    issue(0);
    commit();
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        if (s + 1 < nstage) issue(s + 1);
        // 48 steps (ks, ky, c), one MFMA each; the operands of step i + 1 are read before the MFMA of step i is issued
        {
            constexpr int NST = (KC / 2) * 12;
            float av[2], bv[2];
            auto rd = [&](int st, int slot) {
                const int ks = st / 12, ky = (st % 12) / 4, c = st % 4;
                bv[slot] = smem[b_off + (2 * ks * 4 + c) * RS + ky * NP];
                av[slot] = smem[a_off + ((ky * 4 + c) * KC + 2 * ks) * BM];
            };
            rd(0, 0);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                if (st + 1 < NST) rd(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);       // keep those reads in front of the MFMA of step st
                acc[st % 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st & 1], bv[st & 1], acc[st % 4], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        if (s + 1 < nstage) commit();
        __syncthreads();
    }
    // epilogue: output transform (two adjacent columns per accumulator element), then the direct kernel's epilogue stages
    const int mbase = mb * BM + wm * 32;
    const size_t plane = (size_t)p.H * p.W;
    const size_t off0 = ((size_t)b * p.M + mbase) * plane + (size_t)(y0 + 2 * wr + rr) * p.W + x0 + 2 * jj;
    const float g_pos = p.act == 3 ? 1.4142135623730951f : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dm = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int m = mbase + dm;
        float v0 = acc[0][r] + acc[1][r] + acc[2][r];
        float v1 = acc[1][r] - acc[2][r] - acc[3][r];
        const float sc = p.osc ? p.osc[(size_t)b * p.M + m] : 1.f, bi = p.bias ? p.bias[m] : 0.f;
        v0 = v0 * sc + bi;
        v1 = v1 * sc + bi;
        if (p.act >= 3) {
            v0 = (v0 > 0.f ? v0 : v0 * 0.2f) * g_pos;
            v1 = (v1 > 0.f ? v1 : v1 * 0.2f) * g_pos;
        }
        const size_t o = off0 + (size_t)dm * plane;
        if (p.res) {
            const f32x2 rv = *reinterpret_cast<const f32x2*>(p.res + o);
            v0 += rv[0]; v1 += rv[1];
        }
        if (p.mref) {
            const f32x2 q = *reinterpret_cast<const f32x2*>(p.mref + o);
            v0 *= q[0] > 0.f ? p.mgain : 0.2f * p.mgain;
            v1 *= q[1] > 0.f ? p.mgain : 0.2f * p.mgain;
        }
        f32x2 v; v[0] = v0; v[1] = v1;
        *reinterpret_cast<f32x2*>(p.out + o) = v;
    }
}

}  // namespace

static int wino_bm(int M) { return M % 128 == 0 ? 128 : (M % 64 == 0 ? 64 : (M % 32 == 0 ? 32 : 0)); }

extern "C" int te_conv_wino_supported(int B, int K, int M, int H, int W) {
    if (!(B > 0 && K >= KC && K % KC == 0 && M > 0 && H > 0 && W >= TW && W % TW == 0)) return 0;
    const int bm = wino_bm(M);
    if (!bm) return 0;
    const int th = 2 * (8 / (bm / 32));
    return (H % th == 0 && (int64_t)K * H * W * 4 < 0x7FFFFFFF && (int64_t)B * (H / th) * (W / TW) * (M / bm) < 0x7FFFFFF0) ? 1 : 0;
}

template <int BM>
static void wino_launch_t(WinoArgs a, hipStream_t s) {
    typedef Geo<BM> G;
    a.tiles_x = a.W / TW; a.tiles_y = a.H / G::TH; a.mblocks = a.M / BM;
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    a.nt8 = te::xcd_banded() ? (int)te::cdiv(a.ntiles, 8) : 0;
    const size_t lds = sizeof(float) * (G::T_FLOATS + G::U_FLOATS);
    static std::atomic<uint64_t> attr_done{0};
    te::allow_big_lds(attr_done, (const void*)wino3x3_kernel<BM>, 96 * 1024);
    const int64_t blocks = te::cdiv(a.ntiles, 8) * 8 * a.mblocks;
    wino3x3_kernel<BM><<<dim3((unsigned)blocks), WT, lds, s>>>(a);
}

int te_wino_launch(float* out, const float* in, const float* U, const float* isc, const float* osc, const float* bias, const float* res,
                   const float* mask_ref, float mask_gain, int act, int B, int K, int M, int H, int W, hipStream_t s) {
    TE_REQUIRE(te_conv_wino_supported(B, K, M, H, W), TE_ERR_UNSUPPORTED,
               "te_conv_f32(TE_CONV_3X3W): needs K %% 8 == 0, M %% 32 == 0, W %% 32 == 0 and H a multiple of the tile height "
               "(te_conv_wino_supported)");
    TE_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(res) |
                 reinterpret_cast<uintptr_t>(mask_ref)) & 15) == 0 && (reinterpret_cast<uintptr_t>(in) & 3) == 0, TE_ERR_UNSUPPORTED,
               "te_conv_f32(TE_CONV_3X3W): 16-byte aligned tensors required");
    WinoArgs a{};
    a.out = out; a.in = in; a.U = U; a.isc = isc; a.osc = osc; a.bias = bias; a.res = res; a.mref = mask_ref; a.mgain = mask_gain; a.act = act;
    a.B = B; a.K = K; a.M = M; a.H = H; a.W = W;
    switch (wino_bm(M)) {
        case 128: wino_launch_t<128>(a, s); break;
        case 64: wino_launch_t<64>(a, s); break;
        default: wino_launch_t<32>(a, s); break;
    }
    return te::launch_status("te_conv_f32(TE_CONV_3X3W)");
}
