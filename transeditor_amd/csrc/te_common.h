// Shared host-side helpers for libte_hip.so (gfx950 only; no CUDA / multi-arch paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/te_hip.h"

namespace te {

// thread-local description of the most recent failure on this host thread
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE property: do it once per (kernel, device), safely from
// any host thread (the autograd engine's backward threads call the ABI concurrently; setting it twice is harmless).
inline void allow_big_lds(std::atomic<uint64_t>& done, const void* fn, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
}

constexpr int kNumCU = 256;  // MI355X
constexpr int kNumXCD = 8;

}  // namespace te

#define TE_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return te::fail((code), __VA_ARGS__); } while (0)
