// Shared host-side helpers for libte_hip.so (gfx950 only; no CUDA / multi-arch paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

constexpr int kNumCU = 256;  // MI355X
constexpr int kNumXCD = 8;

}  // namespace te

#define TE_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return te::fail((code), __VA_ARGS__); } while (0)
