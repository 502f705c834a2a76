// Sustained rate of v_mfma_f32_16x16x4_f32 (the weight-gradient stream kernel's instruction) next to 32x32x2: NACC
// independent accumulators per wave, issued back to back, 1..2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, const float* in, int iters) {
    float a[4], b[4];
    for (int j = 0; j < 4; ++j) { a[j] = in[threadIdx.x + 64 * j]; b[j] = in[threadIdx.x + 64 * j + 256]; }
    floatx4 c[NACC];
    for (int n = 0; n < NACC; ++n) c[n] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[n & 3], b[(n >> 2) & 3], c[n], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) s += c[n][0] + c[n][1] + c[n][2] + c[n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(float* out, const float* in, int iters) {
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    floatx16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static float time_ms(F&& launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float *in, *out;
    hipMalloc(&in, 1024 * 4); hipMalloc(&out, 4096 * 256 * 4);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = 256 * wps;
        float ms = time_ms([&] { hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, in, iters); });
        printf("32x32x2  4 acc   waves/SIMD %d: %.1f TFLOP/s\n", wps, (double)blocks * 4 * iters * 4 * 4096.0 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, in, iters); });
        printf("16x16x4  4 acc   waves/SIMD %d: %.1f TFLOP/s\n", wps, (double)blocks * 4 * iters * 4 * 2048.0 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k16<16>, dim3(blocks), dim3(256), 0, 0, out, in, iters); });
        printf("16x16x4 16 acc   waves/SIMD %d: %.1f TFLOP/s\n", wps, (double)blocks * 4 * iters * 16 * 2048.0 / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k16<32>, dim3(blocks), dim3(256), 0, 0, out, in, iters / 2); });
        printf("16x16x4 32 acc   waves/SIMD %d: %.1f TFLOP/s\n", wps, (double)blocks * 4 * (iters / 2) * 32 * 2048.0 / ms / 1e9);
    }
    return 0;
}
