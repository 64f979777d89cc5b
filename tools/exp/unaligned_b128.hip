// does raw_buffer_load_b128 accept 4-byte aligned offsets on gfx950 (and how fast is it)?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, float* out, int n, int shift) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned off = (unsigned)(i * 4 + shift) * 4u;
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    out[i] = v[0] + 10.f * v[1] + 100.f * v[2] + 1000.f * v[3];
}
int main() {
    const int n = 1 << 24;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)(i % 7);
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> ho(n / 4);
    for (int shift = 0; shift < 4; ++shift) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        k<<<n / 4 / 256, 256>>>(d, o, n, shift);
        hipEventRecord(a);
        for (int it = 0; it < 10; ++it) k<<<n / 4 / 256, 256>>>(d, o, n, shift);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(ho.data(), o, n, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n / 4 - 1; ++i) {
            int e = i * 4 + shift;
            float want = h[e] + 10.f * h[e + 1] + 100.f * h[e + 2] + 1000.f * h[e + 3];
            if (ho[i] != want) ++bad;
        }
        printf("shift %d floats: bad %d of %d, %.3f ms per pass (%.1f GB/s)\n", shift, bad, n / 4 - 1, ms / 10, n * 4.0 / (ms / 10) / 1e6);
    }
    return 0;
}
