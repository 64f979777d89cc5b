// F1 (upsampling layers): 3x3 transposed convolution, stride 2, split by OUTPUT ROW PARITY into two implicit GEMMs,
// with 16-byte staging of the input tile.
//
// Reference: ModulatedConv2d.forward upsample branch, model_spatial_query.py:310-321 (F.conv_transpose2d(stride=2) on B
// materialised weight copies).  Output pixel (2i + a, 2j + b) of "cell" (i, j) only receives the taps with ky = a and
// kx = b (mod 2):
//     out[2i + a, 2j + b] = sum_{ky in T(a), kx in T(b)} W[ky][kx] * x[i - (ky >> 1), j - (kx >> 1)],   T(0) = {0, 2}, T(1) = {1}
// conv.hip's TE_CONV_T2 kernel keeps all four phase accumulators of a cell block in one wave (8 accumulator tiles = 128
// VGPRs -> 256 VGPRs, 2 waves per SIMD, 70 % MFMA utilisation).  Here a BLOCK owns ONE ROW PARITY a: 64 output channels
// x 128 cells (4 rows x 32 columns), 4 waves as 2 x 2, per wave 2 cell blocks x 2 column phases = 4 accumulator tiles
// (64 VGPRs) -> 3 waves per SIMD, and only the 6 (a = 0) or 3 (a = 1) taps of that parity are staged (same FLOPs in
// total).  The two column phases of a cell are adjacent output pixels: every lane stores 8 contiguous bytes, a wave writes
// full 256-byte row segments.  Measured dead ends on the way (profiles/README.md): one phase per block with stride-2
// 4-byte stores (55 TFLOP/s: partial-line writes); this split with the 4-byte-per-lane input staging of conv.hip
// (72 TFLOP/s: at 48-96 MFMAs per stage the 16 dword loads per thread and stage saturate the texture addresser, 12 waves
// per CU).  So the input tile (5 rows x [1 + 32] columns per channel) is fetched as 8 aligned 16-byte loads + one 4-byte
// load per row: 3 vector loads per thread and stage instead of 16, committed to LDS with ds_write_b128.
// The two parity blocks of a tile are neighbours in the grid so the input tile both read stays on chip.
// Covers the body cells [0,H) x [0,W) of images with W % 32 == 0 and >= 96 output channels; the last output row / column
// (cells i = H, j = W) and every other shape stay on conv.hip's kernel.
#include "conv_common.h"

namespace {

constexpr int BM = 64, NBW = 2, WN = 2;       // block tile: 64 output channels (2 waves x 32) x 128 cells (2 waves x 2 x 32)
constexpr int TW = 32, TH = 4, TIH = TH + 1;
constexpr int TIWP = 36;                       // LDS row: [3 unused][x(cj0 - 1)][x(cj0) ... x(cj0 + 31)] -> the 32 body columns are 16-byte aligned
constexpr int CS = TIH * TIWP;                 // per-channel stride (180 floats)

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
    int tiles_x, tiles_y;
};

#ifndef T2P_KC0
#define T2P_KC0 16
#define T2P_KC1 16
#endif
#ifndef T2P_OCC
#define T2P_OCC 3
#endif
template <int A> struct Phase {
    static constexpr int NY = A ? 1 : 2, NT = NY * 3;                  // taps of this row parity
    static constexpr int KC = A ? T2P_KC1 : T2P_KC0;                   // channels per stage
    static constexpr int ky(int t) { return A ? 1 : 2 * (t / 3); }
    static constexpr int kx(int t) { return t % 3; }
};

template <int A, bool HAS_ISC>
__device__ __forceinline__ void t2p_body(const T2pArgs& p, int t, float* smem) {
    using P = Phase<A>;
    constexpr int NT = P::NT, KC = P::KC;
    constexpr int WSTAGE = NT * KC * BM;
    constexpr int WLDR = WSTAGE / 4 / NTHREADS;
    static_assert(WSTAGE % (4 * NTHREADS) == 0, "weight stage must be a whole number of 16-byte loads per thread");
    constexpr int NV4 = KC * TIH * 8;                  // 16-byte items of the input tile per stage
    constexpr int NV = (NV4 + NTHREADS - 1) / NTHREADS;
    static_assert(KC * TIH <= NTHREADS, "one 4-byte item (the column left of the tile) per thread");
    float* wl = smem;                    // [NT][KC][BM]
    float* xl = smem + WSTAGE;           // [KC][TIH][TIWP]
    float* sl = xl + KC * CS;            // [Kp + 32] style scales of this sample (zero tail: a last stage may run past Kp)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;

    const int tx_i = t % p.tiles_x; t /= p.tiles_x;
    const int ty_i = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int m0 = blockIdx.y * BM;
    const int ci0 = ty_i * TH, cj0 = tx_i * TW;

    // ---- staging descriptors (constant over the K loop): item -> (channel of the stage, byte offset inside a channel plane)
    const unsigned plane4 = (unsigned)p.Hi * p.Wi * 4u;
    unsigned gv[NV];
    int lv[NV], kv[NV];
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const int e = tid + NTHREADS * r;
        const int kk = e / (TIH * 8), rem = e - kk * (TIH * 8);
        const int ry = rem >> 3, q = rem & 7;
        const int gy = ci0 - 1 + ry;
        kv[r] = e < NV4 ? kk : -1;
        lv[r] = kk * CS + ry * TIWP + 4 + 4 * q;
        gv[r] = (e < NV4 && gy >= 0 && gy < p.Hi) ? (unsigned)(gy * p.Wi + cj0 + 4 * q) * 4u : OOBH;
    }
    int ks = -1, ls = 0;
    unsigned gs = OOBH;
    if (tid < KC * TIH) {                // the column left of the tile (x[., cj0 - 1]): one 4-byte item
        const int kk = tid / TIH, ry = tid - kk * TIH;
        const int gy = ci0 - 1 + ry;
        ks = kk;
        ls = kk * CS + ry * TIWP + 3;
        if (gy >= 0 && gy < p.Hi && cj0 > 0) gs = (unsigned)(gy * p.Wi + cj0 - 1) * 4u;
    }
    const __amdgpu_buffer_rsrc_t irs = make_rsrc(p.in + (size_t)b * p.K * p.Hi * p.Wi, (unsigned)p.K * plane4);
    if (HAS_ISC) {
        for (int k = tid; k < p.Kp + 32; k += NTHREADS) sl[k] = k < p.K ? p.isc[(size_t)b * p.K + k] : 0.f;
    }

    int boff[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int c = wn * (NBW * 32) + nb * 32 + l31;     // cell of the tile: row c >> 5, column c & 31
        boff[nb] = ((c >> 5) + 1) * TIWP + (c & 31) + 4 + half * CS;
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
    f32x4 xv[NV];
    float xs = 0.f;

    for (int k0 = -KC; k0 < p.Kp; k0 += KC) {
        if (k0 >= 0) {
            __syncthreads();              // every wave finished reading the previous stage (first pass: sl is complete)
#pragma unroll
            for (int r = 0; r < WLDR; ++r) *reinterpret_cast<f32x4*>(wl + (tid + NTHREADS * r) * 4) = wreg[r];
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                if (kv[r] >= 0) {
                    const float s = HAS_ISC ? sl[k0 + kv[r]] : 1.f;
                    *reinterpret_cast<f32x4*>(xl + lv[r]) = HAS_ISC ? xv[r] * s : xv[r];
                }
            }
            if (ks >= 0) xl[ls] = HAS_ISC ? xs * sl[k0 + ks] : xs;
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
            // unconditional buffer loads: rows above / below the image and the channel tail come back as 0 from the
            // hardware range check (an offset containing OOBH exceeds num_records)
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int k = kn + max(kv[r], 0);
                xv[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, gv[r] + (k < p.K ? (unsigned)k * plane4 : OOBH), 0, 0));
            }
            {
                const int k = kn + max(ks, 0);
                xs = buf_load(irs, gs + (k < p.K ? (unsigned)k * plane4 : OOBH));
            }
        }
        if (k0 < 0) continue;

#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                const int toff = -(P::ky(tt) == 2 ? TIWP : 0) - (P::kx(tt) == 2 ? 1 : 0);
                const float a = wl[(tt * KC + kk) * BM + aoff];
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    const float bv = xl[kk * CS + boff[nb] + toff];
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
        const int ci = ci0 + (c >> 5), cj = cj0 + (c & 31);
        const bool cell_ok = ci < p.Hi;
        const size_t opix = (size_t)(2 * ci + A) * p.Wo + 2 * cj;
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
                typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
                *reinterpret_cast<f32x2u*>(p.out + ((size_t)b * p.M + m) * oplane + opix) = f32x2u{v0, v1};
            }
        }
    }
}

template <bool HAS_ISC>
__global__ __launch_bounds__(NTHREADS, T2P_OCC) void t2p_kernel(const T2pArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // the two row parities of a tile are neighbours in the grid (the heavier a = 0 block first)
    if ((blockIdx.x & 1) == 0) t2p_body<0, HAS_ISC>(p, blockIdx.x >> 1, smem);
    else t2p_body<1, HAS_ISC>(p, blockIdx.x >> 1, smem);
}

}  // namespace

bool te_t2p_supported(int M, int H, int W) { return M >= 96 && W % TW == 0 && H > 16 && W > 16; }

// body cells [0,H) x [0,W) of the transposed convolution: in [B,K,H,W] -> out [B,M,2H+1,2W+1] rows < 2H, columns < 2W
int te_launch_t2p(float* out, const float* in, const float* wp, const float* isc, const float* osc, const float* bias, int act,
                  int B, int K, int M, int H, int W, hipStream_t s) {
    T2pArgs a{};
    a.out = out; a.in = in; a.wp = wp; a.isc = isc; a.osc = osc; a.bias = bias; a.act = act;
    a.B = B; a.K = K; a.M = M; a.Kp = (K + KPAD - 1) / KPAD * KPAD; a.Mp = (M + MPAD - 1) / MPAD * MPAD;
    a.Hi = H; a.Wi = W; a.Ho = 2 * H + 1; a.Wo = 2 * W + 1;
    if ((int64_t)K * H * W * 4 >= (int64_t)OOBH) return te::fail(TE_ERR_UNSUPPORTED, "te_conv_f32: sample of %d x %dx%d exceeds 1 GiB", K, H, W);
    a.tiles_x = W / TW;
    a.tiles_y = (H + TH - 1) / TH;
    constexpr int kc = T2P_KC0 > T2P_KC1 ? T2P_KC0 : T2P_KC1;
    const size_t lds = sizeof(float) * ((size_t)std::max(6 * T2P_KC0, 3 * T2P_KC1) * BM + (size_t)kc * CS + a.Kp + 32);
    dim3 grid((unsigned)(2 * a.tiles_x * a.tiles_y * B), (unsigned)te::cdiv(M, BM));
    if (isc) t2p_kernel<true><<<grid, NTHREADS, lds, s>>>(a);
    else t2p_kernel<false><<<grid, NTHREADS, lds, s>>>(a);
    return 0;
}
