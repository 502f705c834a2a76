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

// 16 bytes of a tensor that one launch reads (writes) exactly once and that is far larger than the 32 MB of L2 — the
// [pixels, 64] activations and gradients of the decoder's last layers, the NOTEARS-MLP hidden — as a non-temporal
// access: the line does not displace what other kernels and other waves of this one will reuse.  Measured on the last
// decoder layer's apply pass (reads 1.9 GB, writes 1.9 GB): 751 -> 722 us; all streaming kernels together (that pass,
// BatchNorm-backward apply, activation backward, the NOTEARS-MLP tail): same-box step 11.00 -> 10.94 ms over six pairs.
// CMVAE_STREAM_NT (default 1): 0 = plain accesses everywhere (comparison builds).
#ifndef CMVAE_STREAM_NT
#define CMVAE_STREAM_NT 1
#endif
typedef float cmvae_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_once(const float* p) {
#if CMVAE_STREAM_NT
    const cmvae_f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const cmvae_f32x4*>(p));
    return make_float4(q[0], q[1], q[2], q[3]);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void store_once(float* p, const float4& v) {
#if CMVAE_STREAM_NT
    __builtin_nontemporal_store(cmvae_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<cmvae_f32x4*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

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
