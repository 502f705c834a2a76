// Shared host/device helpers for libcmvae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/cmvae_abi.h"

namespace cmvae {

constexpr int D = CMVAE_D;
constexpr int M1 = CMVAE_M1;
constexpr int HID = D * M1;                      // 2048 hidden units of the NOTEARS-MLP
constexpr float LOG_NORM = -0.91893853320467274178f;   // -0.5*log(2*pi)

int set_error(const char* fmt, ...);

// hipGetLastError reports the last error of ANY earlier runtime call of this thread (the host framework's device
// probing included); every entry point drops that stale state first so check_launch speaks for its own launches.
inline void clear_stale_error() { (void)hipGetLastError(); }

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error("%s: %s", what, hipGetErrorString(e));
    return 0;
}

// Workgroups of `kernel` the device keeps resident at once (CUs x occupancy): the grid of a persistent kernel
// that loops over its work items, so that no second, partly filled round of workgroups trails the first.
inline int resident_blocks(const void* kernel, int block_threads, int fallback) {
    int dev = 0, cus = 0, occ = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, block_threads, 0) != hipSuccess || cus < 1 ||
        occ < 1) {
        (void)hipGetLastError();
        return fallback;
    }
    return cus * occ;
}

// Compute units of the current device (256 on MI355X, also the answer when no device is present: the workspace-size
// queries are pure host functions).  Grids of persistent kernels whose occupancy is fixed by their launch bounds are
// `compute_units() * workgroups per CU`.
inline int compute_units() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            return 256;
        }
        return n;
    }();
    return cus;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// expm1 accurate to ~2e-7 relative without the cost of ocml's expm1f.
__device__ __forceinline__ float expm1_fast(float x) {
    // Horner, degree 8 on [-0.35, 0]; exp(x)-1 below that (|result| >= 0.29)
    float p = 1.f / 40320.f;
    p = fmaf(p, x, 1.f / 5040.f);
    p = fmaf(p, x, 1.f / 720.f);
    p = fmaf(p, x, 1.f / 120.f);
    p = fmaf(p, x, 1.f / 24.f);
    p = fmaf(p, x, 1.f / 6.f);
    p = fmaf(p, x, 0.5f);
    p = fmaf(p, x, 1.f);
    p = p * x;
    float e = __expf(x) - 1.f;
    return x > -0.35f ? p : e;
}
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1_fast(x); }

}  // namespace cmvae

#define CMVAE_REQUIRE(cond, ...) \
    do { if (!(cond)) return cmvae::set_error(__VA_ARGS__); } while (0)
