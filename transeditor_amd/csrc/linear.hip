// Small dense layers of the mapping / attention stack (reference: EqualLinear.forward, model_spatial_query.py:213-221,
// used ~60 times per generator pass on [256 | 16, 512]-sized activations).  These GEMMs are far too small for a
// library GEMM's tile heuristics (hipBLASLt picks one 128x256 macro-tile for [256,512]x[512,128]: 61 us) and each
// costs 3-8 separate launches (weight * scale, bias * lr_mul, GEMM, bias add, activation, residual).  One kernel:
//
//     C[i,j] = act( alpha * sum_k A(i,k) * B(k,j) + beta * bias[j] ) + residual[i,j]
//
// with arbitrary element strides for A and B, so the same kernel computes  y = x W^T  (forward),  dx = g W  and
// dW = g^T x.  Exact fp32 on v_mfma_f32_32x32x2_f32.  No LDS staging: everything is L2-resident (<= 1 MB operands); a
// block of 1-16 waves owns one 32x32 output tile, the waves split K and combine through LDS.  Operands whose
// reduction index is contiguous are read 16 bytes per lane (one load feeds 4 MFMAs).
#include "te_common.h"
#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LinArgs {
    float* c;
    float* pre;              // optional: pre-activation output (for the GELU backward)
    const float* a;
    const float* b;
    const float* bias;
    const float* residual;
    float* arowsum;          // optional: arowsum[i] = rs_scale * sum_k A(i,k)  (bias gradient when A = g^T)
    float rs_scale;
    int I, J, K;
    int64_t sai, sak, sbk, sbj;   // element strides: A(i,k) = a[i*sai + k*sak], B(k,j) = b[k*sbk + j*sbj]
    int64_t sci, scj;             // C(i,j) = c[i*sci + j*scj]  (pre / residual use the same addressing)
    float alpha, beta;
    int act;                 // 0 none, 1 GELU (erf), 3 leaky-ReLU(0.2) * sqrt(2)
    // batch over blockIdx.z (the 16 per-token layers of a mapping network in one launch): uniform element strides for
    // a / c, and either a uniform stride or a per-z offset table (separately allocated parameters) for b / bias
    int64_t za, zc, zb, zbias, zrs;      // zrs: stride of arowsum per z
    int use_tab;
    int64_t b_tab[16], bias_tab[16];
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// AV / BV: the operand's reduction stride is 1 and rows are 16-byte aligned -> float4 loads.
// NW waves share one 32x32 tile and split K (host picks NW so a wave's slice is <= 64: every load of the slice is in
// flight at once, the kernel costs one memory latency + <= 32 MFMAs); all waves then reduce the NW partial tiles through
// LDS, one output element per thread, so the epilogue's loads and stores are coalesced rows.
template <int NW, bool AV, bool BV>
__global__ __launch_bounds__(NW * 64) void small_gemm_kernel(const LinArgs p) {
    __shared__ float red[NW][32 * 32];     // 64 KB at NW = 16
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int ia = min(i0 + l31, p.I - 1), jb = min(j0 + l31, p.J - 1);     // clamped: padded rows / columns are never stored
    const int z = blockIdx.z;
    const float* ap = p.a + z * p.za + (int64_t)ia * p.sai;
    const float* bp = p.b + (p.use_tab ? p.b_tab[z] : z * p.zb) + (int64_t)jb * p.sbj;
    const float* biasp = p.bias ? p.bias + (p.use_tab ? p.bias_tab[z] : z * p.zbias) : nullptr;
    float* cp = p.c + z * p.zc;
    // this wave's K range (multiples of 8 so the float4 path stays aligned)
    const int kq = ((p.K + 8 * NW - 1) / (8 * NW)) * 8;
    const int kb = wid * kq, ke = min(p.K, kb + kq);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float asum = 0.f;

#pragma unroll 8
    for (int k = kb; k < ke; k += 8) {
        // lane (row, half) supplies k = k + 4*half + {0,1,2,3} over four MFMAs (A and B use the same assignment)
        const int kk = k + 4 * half;
        float av[4], bv[4];
        if (AV && kk + 3 < ke) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(ap + kk);
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = t[q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = (kk + q < ke) ? ap[(int64_t)(kk + q) * p.sak] : 0.f;
        }
        if (BV && kk + 3 < ke) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(bp + kk);
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = t[q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = (kk + q < ke) ? bp[(int64_t)(kk + q) * p.sbk] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
        asum += (av[0] + av[1]) + (av[2] + av[3]);
    }

    // C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid][((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
    __syncthreads();
#pragma unroll
    for (int e = tid; e < 1024; e += NW * 64) {
        const int row = e >> 5, col = e & 31, i = i0 + row, j = j0 + col;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][e];
        if (i < p.I && j < p.J) {
            v *= p.alpha;
            if (biasp) v += biasp[j] * p.beta;
            const int64_t o = (int64_t)i * p.sci + (int64_t)j * p.scj;
            if (p.pre) p.pre[o] = v;
            if (p.act == 1) v = gelu_erf(v);
            else if (p.act == 3) v = (v > 0.f ? v : v * 0.2f) * 1.4142135623730951f;
            if (p.residual) v += p.residual[o];
            cp[o] = v;
        }
    }
    if (p.arowsum && blockIdx.x == 0) {          // uniform branch; the tile partials are consumed, reuse the LDS
        __syncthreads();
        red[0][tid] = asum;
        __syncthreads();
        if (tid < 32 && i0 + tid < p.I) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[0][w * 64 + tid] + red[0][w * 64 + 32 + tid];
            p.arowsum[z * p.zrs + i0 + tid] = t * p.rs_scale;
        }
    }
}

// second pass of the split-K form: c = act(alpha * sum_s ws[s] + beta * bias) + residual  (fixed summation order)
__global__ __launch_bounds__(256) void splitk_finish_kernel(float* __restrict__ c, float* __restrict__ pre, const float* __restrict__ ws,
                                                            const float* __restrict__ bias, const float* __restrict__ residual, int S,
                                                            int I, int J, float alpha, float beta, int act) {
    const int64_t total = (int64_t)I * J;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int q = 0; q < S; ++q) v += ws[(int64_t)q * total + e];
        v *= alpha;
        if (bias) v += bias[e % J] * beta;
        if (pre) pre[e] = v;
        if (act == 1) v = gelu_erf(v);
        else if (act == 3) v = (v > 0.f ? v : v * 0.2f) * 1.4142135623730951f;
        if (residual) v += residual[e];
        c[e] = v;
    }
}

template <int NW>
void launch_nw(const LinArgs& p, bool av, bool bv, dim3 grid, hipStream_t s) {
    if (av && bv) small_gemm_kernel<NW, true, true><<<grid, NW * 64, 0, s>>>(p);
    else if (av) small_gemm_kernel<NW, true, false><<<grid, NW * 64, 0, s>>>(p);
    else if (bv) small_gemm_kernel<NW, false, true><<<grid, NW * 64, 0, s>>>(p);
    else small_gemm_kernel<NW, false, false><<<grid, NW * 64, 0, s>>>(p);
}

}  // namespace

static int launch_small_gemm(LinArgs& p, int nz, hipStream_t s) {
    const bool av = p.sak == 1 && p.sai % 4 == 0 && p.za % 4 == 0 && (reinterpret_cast<uintptr_t>(p.a) & 15) == 0;
    bool bv = p.sbk == 1 && p.sbj % 4 == 0 && p.zb % 4 == 0 && (reinterpret_cast<uintptr_t>(p.b) & 15) == 0;
    if (p.use_tab)
        for (int z = 0; z < nz; ++z) bv = bv && p.b_tab[z] % 4 == 0;
    dim3 grid((unsigned)te::cdiv(p.J, 32), (unsigned)te::cdiv(p.I, 32), (unsigned)nz);
    // waves per tile: a wave's K slice <= 64 where possible, and more waves when the grid alone cannot fill the chip
    const int64_t tiles = (int64_t)grid.x * grid.y * grid.z;
    const int K = p.K;
    int nw = K > 256 ? 8 : (K > 64 ? 4 : (K > 16 ? 2 : 1));
    if (K >= 512 && (K > 512 || tiles < 2 * te::kNumCU)) nw = 16;
    if (nw == 16) launch_nw<16>(p, av, bv, grid, s);
    else if (nw == 8) launch_nw<8>(p, av, bv, grid, s);
    else if (nw == 4) launch_nw<4>(p, av, bv, grid, s);
    else if (nw == 2) launch_nw<2>(p, av, bv, grid, s);
    else launch_nw<1>(p, av, bv, grid, s);
    return 0;
}

extern "C" int te_small_gemm_f32(float* c, float* pre, const float* a, const float* b, const float* bias,
                                 const float* residual, float* arowsum, float rs_scale, int I, int J, int K,
                                 int64_t sai, int64_t sak, int64_t sbk, int64_t sbj, float alpha, float beta, int act,
                                 te_stream_t stream_) {
    TE_REQUIRE(c && a && b, TE_ERR_NULL, "te_small_gemm_f32: NULL pointer");
    TE_REQUIRE(I > 0 && J > 0 && K > 0, TE_ERR_SHAPE, "te_small_gemm_f32: bad dims");
    TE_REQUIRE(act == 0 || act == 1 || act == 3, TE_ERR_UNSUPPORTED, "te_small_gemm_f32: act must be 0, 1 or 3");
    LinArgs p{};
    p.c = c; p.pre = pre; p.a = a; p.b = b; p.bias = bias; p.residual = residual; p.arowsum = arowsum; p.rs_scale = rs_scale;
    p.I = I; p.J = J; p.K = K; p.sai = sai; p.sak = sak; p.sbk = sbk; p.sbj = sbj; p.sci = J; p.scj = 1;
    p.alpha = alpha; p.beta = beta; p.act = act;
    launch_small_gemm(p, 1, (hipStream_t)stream_);
    return te::launch_status("te_small_gemm_f32");
}

extern "C" int te_small_gemm_batched_f32(float* c, const float* a, const float* b, const float* bias, int nz, int64_t za,
                                         int64_t zc, int64_t zb, int64_t zbias, const int64_t* b_tab,
                                         const int64_t* bias_tab, int I, int J, int K, int64_t sai, int64_t sak, int64_t sbk,
                                         int64_t sbj, int64_t sci, int64_t scj, float alpha, float beta, int act,
                                         te_stream_t stream_) {
    TE_REQUIRE(c && a && b, TE_ERR_NULL, "te_small_gemm_batched_f32: NULL pointer");
    TE_REQUIRE(I > 0 && J > 0 && K > 0 && nz > 0, TE_ERR_SHAPE, "te_small_gemm_batched_f32: bad dims");
    TE_REQUIRE(act == 0 || act == 1 || act == 3, TE_ERR_UNSUPPORTED, "te_small_gemm_batched_f32: act must be 0, 1 or 3");
    TE_REQUIRE(!b_tab || nz <= 16, TE_ERR_UNSUPPORTED, "te_small_gemm_batched_f32: at most 16 table entries per launch");
    TE_REQUIRE(!b_tab || !bias || bias_tab, TE_ERR_NULL, "te_small_gemm_batched_f32: bias_tab missing");
    LinArgs p{};
    p.c = c; p.a = a; p.b = b; p.bias = bias;
    p.I = I; p.J = J; p.K = K; p.sai = sai; p.sak = sak; p.sbk = sbk; p.sbj = sbj; p.sci = sci; p.scj = scj;
    p.alpha = alpha; p.beta = beta; p.act = act;
    p.za = za; p.zc = zc; p.zb = zb; p.zbias = zbias;
    if (b_tab) {
        p.use_tab = 1;
        for (int z = 0; z < nz; ++z) { p.b_tab[z] = b_tab[z]; p.bias_tab[z] = bias_tab ? bias_tab[z] : 0; }
    }
    launch_small_gemm(p, nz, (hipStream_t)stream_);
    return te::launch_status("te_small_gemm_batched_f32");
}

/* wide reductions (the discriminator's 8192 -> 512 linear, model_spatial_query.py:831-834): K is cut into S chunks that run
 * as the z dimension of the same kernel (S x tiles blocks instead of `tiles`), partial tiles go to the caller's workspace
 * ws[S][I][J], a second tiny kernel sums them in fixed order and applies the epilogue.  Deterministic, no atomics. */
extern "C" int te_small_gemm_splitk_f32(float* c, float* pre, float* ws, int S, const float* a, const float* b, const float* bias,
                                        const float* residual, int I, int J, int K, int64_t sai, int64_t sak, int64_t sbk,
                                        int64_t sbj, float alpha, float beta, int act, te_stream_t stream_) {
    TE_REQUIRE(c && ws && a && b, TE_ERR_NULL, "te_small_gemm_splitk_f32: NULL pointer");
    TE_REQUIRE(I > 0 && J > 0 && K > 0 && S > 0, TE_ERR_SHAPE, "te_small_gemm_splitk_f32: bad dims");
    TE_REQUIRE(K % S == 0 && (K / S) % 8 == 0, TE_ERR_SHAPE, "te_small_gemm_splitk_f32: K must split into S chunks of a multiple of 8");
    TE_REQUIRE(act == 0 || act == 1 || act == 3, TE_ERR_UNSUPPORTED, "te_small_gemm_splitk_f32: act must be 0, 1 or 3");
    const int Kc = K / S;
    LinArgs p{};
    p.c = ws; p.a = a; p.b = b;
    p.I = I; p.J = J; p.K = Kc; p.sai = sai; p.sak = sak; p.sbk = sbk; p.sbj = sbj; p.sci = J; p.scj = 1;
    p.alpha = 1.f; p.beta = 0.f; p.act = 0;
    p.za = (int64_t)Kc * sak; p.zb = (int64_t)Kc * sbk; p.zc = (int64_t)I * J;
    hipStream_t s = (hipStream_t)stream_;
    launch_small_gemm(p, S, s);
    const int64_t total = (int64_t)I * J;
    splitk_finish_kernel<<<(int)std::min<int64_t>(te::cdiv(total, 256), te::kNumCU * 4), 256, 0, s>>>(c, pre, ws, bias, residual, S, I, J,
                                                                                                     alpha, beta, act);
    return te::launch_status("te_small_gemm_splitk_f32");
}
