// Does the sustained fp32 MFMA rate depend on the operand data?  32 accumulators of v_mfma_f32_16x16x4_f32 per wave, two
// waves per SIMD, operands (a) the same two registers for every MFMA of a lane, (b) 16 + 16 different random registers,
// (c) as (b) but with operands that change every iteration (xor with the loop counter).  Reports TFLOP/s and the ratio of
// shader cycles (clock64) to the 100 MHz wall clock (wall_clock64) seen by one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, long long* clk) {
    float a[16], b[16];
    for (int j = 0; j < 16; ++j) {
        a[j] = in[(threadIdx.x * 16 + j) & 4095];
        b[j] = in[(threadIdx.x * 16 + j + 2048) & 4095];
    }
    floatx4 c[32];
    float d[32];
    for (int j = 0; j < 32; ++j) d[j] = in[(threadIdx.x + 64 * j) & 4095];
    for (int n = 0; n < 32; ++n) c[n] = floatx4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { a[j] = __int_as_float(__float_as_int(a[j]) ^ (i & 0xffff)); b[j] = __int_as_float(__float_as_int(b[j]) ^ ((i * 7) & 0xffff)); }
        }
        if (MODE == 3) {        // 32 VALU ops on registers no MFMA reads
#pragma unroll
            for (int j = 0; j < 16; ++j) { d[j] = fmaf(d[j], 1.0001f, 0.5f); d[j + 16] = fmaxf(d[j + 16] * 0.999f, 0.1f); }
        }
        if (MODE == 4) {        // 8 VALU ops on registers no MFMA reads
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = fmaf(d[j], 1.0001f, 0.5f);
        }
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            const int ja = MODE == 0 ? 0 : (n & 15), jb = MODE == 0 ? 0 : ((n * 5 + 3) & 15);
            c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ja], b[jb], c[n], 0, 0, 0);
            if (MODE == 5 && (n & 3) == 0) d[n >> 2] = fmaf(d[n >> 2], 1.0001f, 0.5f);     // 8 VALU ops spread between the MFMAs
        }
    }
    for (int j = 0; j < 32; ++j) c[0][0] += d[j];
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int n = 0; n < 32; ++n) s += c[n][0] + c[n][1] + c[n][2] + c[n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <int MODE>
static void run(float* out, const float* in, long long* clk) {
    const int blocks = 512, iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c[2]; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    printf("mode %d: %.3f ms  %.1f TFLOP/s   clock64/wall_clock64 = %.2f (x100 MHz = %.2f GHz)\n", MODE, ms,
           (double)blocks * 4 * iters * 32 * 2048.0 / ms / 1e9, (double)c[0] / c[1], (double)c[0] / c[1] / 10.0);
}
int main() {
    float *in, *out; long long* clk;
    (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&clk, 16);
    float h[4096]; unsigned x = 12345;
    for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) / 8388608.f - 1.f; }
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(out, in, clk); run<1>(out, in, clk); run<2>(out, in, clk); run<3>(out, in, clk); run<4>(out, in, clk); run<5>(out, in, clk);
    return 0;
}
