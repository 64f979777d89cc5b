// T2: multi-tensor Adam and EMA — ONE launch over all parameter tensors of a module.
//
// Reference: train_spatial_query.py:458-473 (two torch.optim.Adam over 257 / 38 tensors, betas (0, 0.99 ** c)) and
// accumulate(), :56-61 (g_ema <- decay * g_ema + (1 - decay) * g, one mul_ + add_ pair per tensor): hundreds of tiny
// launches per iteration.  Here the tensors are described by a DEVICE table (pointers + sizes) and split into fixed-size
// chunks; block c processes chunk c of tensor chunk_tensor[c].  Pure HBM streaming: Adam reads p, g, v (+ m when
// beta1 != 0) and writes p, m, v once; EMA reads dst, src and writes dst.  16-byte accesses whenever every pointer of
// the tensor is 16-byte aligned (gradients that live inside a GradSync bucket may only be 4-byte aligned: scalar path).
#include "te_common.h"

namespace {

constexpr int THREADS = 256;

struct AdamK {       // scalar prefactors, formed in double on the host (as torch.optim.Adam does) and applied in fp32
    float lr_step, beta1, beta2, w1, w2, eps, bc2_sqrt;      // w1 = 1 - beta1, w2 = 1 - beta2
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, bool has_m, const AdamK& k) {
    // torch.optim.Adam (non-amsgrad, no weight decay), same operation order:
    //   m = lerp(m, g, 1 - beta1);  v = v * beta2 + (1 - beta2) * g * g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
    float mm;
    if (!has_m) mm = g;                                   // beta1 == 0: lerp weight 1 returns g exactly
    else mm = (k.w1 < 0.5f) ? m + k.w1 * (g - m) : g - (g - m) * (1.f - k.w1);
    float vv = v * k.beta2;
    vv = vv + k.w2 * g * g;
    const float denom = sqrtf(vv) / k.bc2_sqrt + k.eps;
    p = p - k.lr_step * (mm / denom);
    m = mm;
    v = vv;
}

__global__ __launch_bounds__(THREADS) void mt_adam_kernel(const int64_t* __restrict__ table, const int32_t* __restrict__ chunks,
                                                           int n, int n_chunks, int chunk_elems, const AdamK k) {
    const int c = blockIdx.x;
    const int t = chunks[c];
    const int64_t off = (int64_t)chunks[n_chunks + c] * chunk_elems;
    float* p = reinterpret_cast<float*>(table[t]);
    const float* g = reinterpret_cast<const float*>(table[n + t]);
    float* m = reinterpret_cast<float*>(table[2 * n + t]);
    float* v = reinterpret_cast<float*>(table[3 * n + t]);
    const int64_t numel = table[4 * n + t];
    if (g == nullptr) return;                              // parameter without gradient this step: skipped, as torch does
    const bool has_m = k.beta1 != 0.f;
    const int64_t len = min((int64_t)chunk_elems, numel - off);
    p += off; g += off; m += off; v += off;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (aligned) {
        const int64_t n4 = len >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += THREADS) {
            float4 P = reinterpret_cast<float4*>(p)[i];
            const float4 G = reinterpret_cast<const float4*>(g)[i];
            float4 M = has_m ? reinterpret_cast<float4*>(m)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 V = reinterpret_cast<float4*>(v)[i];
            adam_elem(P.x, G.x, M.x, V.x, has_m, k);
            adam_elem(P.y, G.y, M.y, V.y, has_m, k);
            adam_elem(P.z, G.z, M.z, V.z, has_m, k);
            adam_elem(P.w, G.w, M.w, V.w, has_m, k);
            reinterpret_cast<float4*>(p)[i] = P;
            reinterpret_cast<float4*>(m)[i] = M;
            reinterpret_cast<float4*>(v)[i] = V;
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < len; i += THREADS) {
            float P = p[i], M = has_m ? m[i] : 0.f, V = v[i];
            adam_elem(P, g[i], M, V, has_m, k);
            p[i] = P; m[i] = M; v[i] = V;
        }
    } else {
        for (int64_t i = threadIdx.x; i < len; i += THREADS) {
            float P = p[i], M = has_m ? m[i] : 0.f, V = v[i];
            adam_elem(P, g[i], M, V, has_m, k);
            p[i] = P; m[i] = M; v[i] = V;
        }
    }
}

__global__ __launch_bounds__(THREADS) void mt_ema_kernel(const int64_t* __restrict__ table, const int32_t* __restrict__ chunks,
                                                          int n, int n_chunks, int chunk_elems, float decay, float one_minus) {
    const int c = blockIdx.x;
    const int t = chunks[c];
    const int64_t off = (int64_t)chunks[n_chunks + c] * chunk_elems;
    float* d = reinterpret_cast<float*>(table[t]) + off;
    const float* s = reinterpret_cast<const float*>(table[n + t]) + off;
    const int64_t len = min((int64_t)chunk_elems, table[2 * n + t] - off);
    // accumulate(): par1.mul_(decay).add_(1 - decay, par2)  (train_spatial_query.py:60-61)
    if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
        const int64_t n4 = len >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += THREADS) {
            float4 D = reinterpret_cast<float4*>(d)[i];
            const float4 S = reinterpret_cast<const float4*>(s)[i];
            D.x = D.x * decay + one_minus * S.x;
            D.y = D.y * decay + one_minus * S.y;
            D.z = D.z * decay + one_minus * S.z;
            D.w = D.w * decay + one_minus * S.w;
            reinterpret_cast<float4*>(d)[i] = D;
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < len; i += THREADS) d[i] = d[i] * decay + one_minus * s[i];
    } else {
        for (int64_t i = threadIdx.x; i < len; i += THREADS) d[i] = d[i] * decay + one_minus * s[i];
    }
}

}  // namespace

extern "C" int te_mt_adam_f32(const int64_t* table, const int32_t* chunks, int n_tensors, int n_chunks, int chunk_elems,
                              double lr, double beta1, double beta2, double eps, int step, te_stream_t stream_) {
    TE_REQUIRE(table && chunks, TE_ERR_NULL, "te_mt_adam_f32: NULL table");
    TE_REQUIRE(n_tensors > 0 && n_chunks > 0 && chunk_elems > 0 && chunk_elems % 4 == 0 && step > 0, TE_ERR_SHAPE,
               "te_mt_adam_f32: bad table dims / step");
    // scalar prefactors in double, as torch.optim.Adam computes them on the host
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    AdamK k;
    k.lr_step = (float)(lr / bc1);
    k.bc2_sqrt = (float)sqrt(bc2);
    k.beta1 = (float)beta1; k.beta2 = (float)beta2; k.w1 = (float)(1.0 - beta1); k.w2 = (float)(1.0 - beta2); k.eps = (float)eps;
    mt_adam_kernel<<<n_chunks, THREADS, 0, (hipStream_t)stream_>>>(table, chunks, n_tensors, n_chunks, chunk_elems, k);
    return te::launch_status("te_mt_adam_f32");
}

extern "C" int te_mt_ema_f32(const int64_t* table, const int32_t* chunks, int n_tensors, int n_chunks, int chunk_elems,
                             double decay, te_stream_t stream_) {
    TE_REQUIRE(table && chunks, TE_ERR_NULL, "te_mt_ema_f32: NULL table");
    TE_REQUIRE(n_tensors > 0 && n_chunks > 0 && chunk_elems > 0 && chunk_elems % 4 == 0, TE_ERR_SHAPE, "te_mt_ema_f32: bad table dims");
    mt_ema_kernel<<<n_chunks, THREADS, 0, (hipStream_t)stream_>>>(table, chunks, n_tensors, n_chunks, chunk_elems, (float)decay,
                                                                  (float)(1.0 - decay));
    return te::launch_status("te_mt_ema_f32");
}
