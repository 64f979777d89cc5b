// Probe for a round-5 design question (NOT product code): can the fp32 convolutions run on the bf16 matrix pipe with fp32-equivalent
// results?  Every fp32 operand is split into three bf16 pieces x = h + m + l (8 + 8 + 8 mantissa bits, exact up to 2^-25 relative);
// products of bf16 pieces are exact in fp32; 6 of the 9 cross products (all but m*l, l*m, l*l: below 2^-24) accumulated in fp32.
//   part 1: accuracy on the hardware (v_mfma_f32_32x32x16_bf16) against double, next to the native fp32 MFMA chain
//   part 2: the rate the matrix pipe sustains with its operands coming from LDS the way a convolution kernel would feed it
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/bf16x_probe.hip -o gpurun_out/bf16x_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short to_bf16(float x) {          // round to nearest even
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float from_bf16(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = to_bf16(x); const float r = x - from_bf16(h);
    m = to_bf16(r); const float r2 = r - from_bf16(m);
    l = to_bf16(r2);
}

// ---- part 1: C[M][N] = A[M][K] B[K][N], one wave per 32 x 32 tile, operands straight from global memory
// mode 0: native fp32 MFMA (32x32x2); 3 / 6 / 9: number of bf16 cross products
__global__ void gemm_probe(float* C, const float* A, const float* B, int M, int N, int K, int mode) {
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(m0 + l31) * K + k + half], B[(size_t)(k + half) * N + n0 + l31], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a[3], b[3];
            for (int j = 0; j < 8; ++j) {
                unsigned short h, m, l;
                split3(A[(size_t)(m0 + l31) * K + k + 8 * half + j], h, m, l);
                a[0][j] = (short)h; a[1][j] = (short)m; a[2][j] = (short)l;
                split3(B[(size_t)(k + 8 * half + j) * N + n0 + l31], h, m, l);
                b[0][j] = (short)h; b[1][j] = (short)m; b[2][j] = (short)l;
            }
            // small terms first
            if (mode >= 9) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0);
            }
            if (mode >= 6) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            }
            if (mode >= 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) C[(size_t)(m0 + (r & 3) + 8 * (r >> 2) + 4 * half) * N + n0 + l31] = acc[r];
}

// ---- part 2: rate.  Block = 512 threads; LDS holds NSLOT operand slots of 64 lanes x 16 bytes; a wave reads RA "A" slots and RB "B" slots
// per group (ds_read_b128 each) and issues NM MFMAs on 4 accumulators per group.  FROM_LDS = 0: operands stay in registers.
template <int RA, int RB, int NM, int FROM_LDS, int OCC>
__global__ __launch_bounds__(512, OCC) void rate_probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16x8* slots = reinterpret_cast<bf16x8*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NSLOT = 48;
    for (int i = tid; i < NSLOT * 64; i += 512) { bf16x8 v; for (int j = 0; j < 8; ++j) v[j] = (short)(0x3C00 + ((i * 7 + j) & 63)); slots[i] = v; }
    __syncthreads();
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a[RA], b[RB];
    for (int i = 0; i < RA; ++i) a[i] = slots[(i + wid) % NSLOT * 64 + lane];
    for (int i = 0; i < RB; ++i) b[i] = slots[(i + 7 + wid) % NSLOT * 64 + lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            if (FROM_LDS) {
#pragma unroll
                for (int i = 0; i < RA; ++i) a[i] = slots[((g * RA + i + wid + it) % NSLOT) * 64 + lane];
#pragma unroll
                for (int i = 0; i < RB; ++i) b[i] = slots[((g * RB + i + 5 + wid + it) % NSLOT) * 64 + lane];
            }
#pragma unroll
            for (int q = 0; q < NM; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q % RA], b[(q / RA) % RB], acc[q & 3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int RA, int RB, int NM, int FROM_LDS, int OCC>
static void run_rate(const char* name, float* out) {
    const int blocks = 256 * (OCC / 2), iters = 400;
    hipFuncSetAttribute((const void*)rate_probe<RA, RB, NM, FROM_LDS, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    rate_probe<RA, RB, NM, FROM_LDS, OCC><<<blocks, 512, 48 * 1024, 0>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(s, 0);
    rate_probe<RA, RB, NM, FROM_LDS, OCC><<<blocks, 512, 48 * 1024, 0>>>(out, iters);
    hipEventRecord(e, 0); hipEventSynchronize(e);
    float ms = 0; hipEventElapsedTime(&ms, s, e);
    const double mfma = (double)blocks * 8 * iters * 12 * NM;
    const double tf = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("  %-58s %7.3f ms  %7.1f TFLOP/s bf16 = %5.1f %% of 2516;  fp32-equivalent at 6 products: %6.1f, at 3: %6.1f   (err %s)\n", name, ms, tf,
           100.0 * tf / 2516.0, tf / 6, tf / 3, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int M = 64, N = 64, K = 1152;
    std::vector<float> A((size_t)M * K), B((size_t)K * N), C((size_t)M * N);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    auto gauss = [&]() { float s = 0; for (int i = 0; i < 6; ++i) s += rnd(); return s; };     // roughly normal
    for (auto& v : A) v = gauss();
    for (auto& v : B) v = gauss();
    std::vector<double> ref((size_t)M * N, 0.0);
    double refmax = 0, refn = 0;
    for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) {
        double s = 0; for (int k = 0; k < K; ++k) s += (double)A[(size_t)i * K + k] * (double)B[(size_t)k * N + j];
        ref[(size_t)i * N + j] = s; refmax = fmax(refmax, fabs(s)); refn += s * s;
    }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    printf("part 1: C = A B, %d x %d x %d, fp32 inputs ~N(0, 2); error against double\n", M, N, K);
    const int modes[4] = {0, 3, 6, 9};
    const char* names[4] = {"native fp32 MFMA (v_mfma_f32_32x32x2_f32)", "bf16 x 3 products (hh, hm, mh)", "bf16 x 6 products", "bf16 x 9 products"};
    for (int t = 0; t < 4; ++t) {
        gemm_probe<<<dim3(N / 32, M / 32), 64>>>(dC, dA, dB, M, N, K, modes[t]);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, en = 0;
        for (size_t i = 0; i < C.size(); ++i) { const double d = (double)C[i] - ref[i]; emax = fmax(emax, fabs(d)); en += d * d; }
        printf("  %-44s max |err| / max |ref| = %.3e   L2 = %.3e\n", names[t], emax / refmax, sqrt(en / refn));
    }
    float* out; hipMalloc(&out, 1024 * 512 * 4);
    printf("part 2: sustained rate of v_mfma_f32_32x32x16_bf16, 8-wave blocks, 12 groups per iteration\n");
    run_rate<1, 1, 4, 0, 4>("registers only, 4 waves/SIMD", out);
    run_rate<1, 1, 4, 0, 2>("registers only, 2 waves/SIMD", out);
    run_rate<3, 3, 6, 1, 4>("6 reads b128 per 6 MFMAs (1 tile/wave), 4 waves/SIMD", out);
    run_rate<3, 3, 6, 1, 2>("6 reads b128 per 6 MFMAs (1 tile/wave), 2 waves/SIMD", out);
    run_rate<6, 3, 12, 1, 2>("9 reads per 12 MFMAs (2 M tiles/wave), 2 waves/SIMD", out);
    run_rate<2, 2, 3, 1, 4>("4 reads per 3 MFMAs (3-product form), 4 waves/SIMD", out);
    return 0;
}
