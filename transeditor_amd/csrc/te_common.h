// Shared host-side helpers for libte_hip.so (gfx950 only; no CUDA / multi-arch paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <stdlib.h>
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

// Tile order of the convolution launches over the 8 XCDs (workgroup ids go round-robin over them, each XCD has its own L2):
// banded (default) = XCD x walks the x-th EIGHTH of the tile list, so horizontally and vertically adjacent tiles - which share
// halo columns / rows, i.e. whole 128-byte lines - run on the same L2; interleaved (TE_XCD_INTERLEAVED=1, the order of rounds
// 2-3) = XCD x takes tiles x, x + 8, ...  A/B with counters: profiles/experiments/r04_xcd_band_ab.log.
inline bool xcd_banded() {
    static const bool interleaved = getenv("TE_XCD_INTERLEAVED") && atoi(getenv("TE_XCD_INTERLEAVED"));
    return !interleaved;
}

}  // namespace te

#define TE_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return te::fail((code), __VA_ARGS__); } while (0)
