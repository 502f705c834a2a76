// Sustained fp32 MFMA rate of this device: every SIMD issues v_mfma_f32_32x32x2_f32 back to back on random operands.
// Prints TFLOP/s and the shader clock seen by s_memtime, for 1..3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, long long* clk) {
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    floatx16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
    float *in, *out; long long* clk;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 8);
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 3; ++wps) {
        const int blocks = 256 * wps, iters = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        double flops = (double)blocks * 4 * iters * 4 * 4096.0;
        printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s   cycle counter: %lld ticks -> %.3f GHz if shader clock\n", wps, ms,
               flops / ms / 1e9, c, c / (ms * 1e6));
    }
    return 0;
}
