// v_mfma_f32_16x16x32_bf16 on ONE wave per SIMD: cycles per MFMA when an accumulator takes six dependent MFMAs in a row
// (the order mfma16_split6 is written in) against the same 96 MFMAs with the four accumulators of a group interleaved,
// and against 16 accumulators round-robin.  Shader cycles (clock64) per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, const uint4* in, int iters, long long* cyc) {
    bf16x8 a[12], b[12];
    for (int j = 0; j < 12; ++j) {
        a[j] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 64 * j) & 1023]);
        b[j] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 64 * j + 512) & 1023]);
    }
    floatx4 c[16];
    for (int n = 0; n < 16; ++n) c[n] = floatx4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {            // per (ja, jb): six dependent MFMAs in a row
#pragma unroll
            for (int ja = 0; ja < 4; ++ja)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                    for (int p = 0; p < 6; ++p)
                        c[ja * 4 + jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[3 * ja + p % 3], b[3 * jb + p / 2], c[ja * 4 + jb], 0, 0, 0);
        } else if (MODE == 1) {     // per ja: the six products outer, the four jb accumulators inner
#pragma unroll
            for (int ja = 0; ja < 4; ++ja)
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb)
                        c[ja * 4 + jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[3 * ja + p % 3], b[3 * jb + p / 2], c[ja * 4 + jb], 0, 0, 0);
        } else {                    // 16 accumulators round-robin
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int n = 0; n < 16; ++n)
                    c[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[3 * (n >> 2) + p % 3], b[3 * (n & 3) + p / 2], c[n], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int n = 0; n < 16; ++n) s += c[n][0] + c[n][1] + c[n][2] + c[n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void run(float* out, const uint4* in, long long* cyc, const char* what) {
    const int iters = 2000, blocks = 256;
    long long h[256];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, in, iters, cyc);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double)h[i];
    printf("%-58s %.1f cycles per MFMA\n", what, s / blocks / iters / 96.0);
}
int main() {
    float* out; uint4* in; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&in, 1024 * 16); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(in, 0x3c, 1024 * 16);
    run<0>(out, in, cyc, "six dependent MFMAs per accumulator, 16 accumulators:");
    run<1>(out, in, cyc, "four accumulators interleaved per product:");
    run<2>(out, in, cyc, "16 accumulators round-robin:");
    return 0;
}
