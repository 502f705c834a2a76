// What does a VALU instruction cost next to fp32 MFMAs on gfx950?  Per loop iteration a burst of V VALU instructions
// (inline asm, so the count and the position are what is written here) and then 64 x v_mfma_f32_32x32x2_f32 on two
// alternating accumulators — the shape of one tap step of the weight-stationary conv GEMM (csrc/conv_gemm.hip).
// KIND 0: v_max_f32, 1: v_pk_fma_f32, 2: v_mov_b32, 3: v_cndmask_b32; SPREAD = 1: one VALU after every 64 / V MFMAs
// instead of a burst.  One and two workgroups per CU (1 / 2 waves per SIMD).  Prints shader cycles per iteration per
// SIMD and the cycles each VALU instruction added.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void valu(float& d0, float& d1, floatx2& p, const floatx2& q) {
    if (KIND == 0) asm volatile("v_max_f32 %0, 0, %0" : "+v"(d0));
    if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(q));
    if (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(d0) : "v"(d1));
    if (KIND == 3) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(d0));
}

template <int V, int KIND, int SPREAD>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* in, int iters) {
    float a[8], b[8], d[32];
    floatx2 p[16], q = {1.0001f, 0.9999f};
    for (int j = 0; j < 8; ++j) { a[j] = in[(threadIdx.x * 8 + j) & 4095]; b[j] = in[(threadIdx.x * 8 + j + 2048) & 4095]; }
    for (int j = 0; j < 32; ++j) d[j] = in[(threadIdx.x + 64 * j) & 4095];
    for (int j = 0; j < 16; ++j) p[j] = floatx2{d[j], d[j + 16]};
    floatx16 c0 = {}, c1 = {};
    for (int i = 0; i < iters; ++i) {
        if (!SPREAD) {
#pragma unroll
            for (int j = 0; j < V; ++j) valu<KIND>(d[j & 31], d[(j + 1) & 31], p[j & 15], q);
        }
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[n & 7], b[(n * 3) & 7], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(n + 1) & 7], b[(n * 5 + 1) & 7], c1, 0, 0, 0);
            if (SPREAD && V > 0 && (n % (32 / (V < 32 ? V : 32))) == 0) {
#pragma unroll
                for (int j = 0; j < (V + 31) / 32; ++j) valu<KIND>(d[(n + j) & 31], d[(n + j + 1) & 31], p[(n + j) & 15], q);
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + p[j].x + p[j].y + d[j] + d[j + 16];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double base[2];
template <int V, int KIND, int SPREAD>
static void run(float* out, const float* in, double ghz) {
    const int iters = 4000;
    for (int occ = 1; occ <= 2; ++occ) {
        const int blocks = 256 * occ;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k<V, KIND, SPREAD>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        // cycles per iteration of ONE wave's work, per SIMD: occ waves share a SIMD
        const double cyc = ms * 1e-3 * ghz * 1e9 / iters / occ;
        if (V == 0) base[occ - 1] = cyc;
        printf("V=%3d kind=%d spread=%d  %d wave(s)/SIMD: %7.0f cycles per 64 MFMAs (ideal 4096)  -> %5.1f TFLOP/s", V, KIND,
               SPREAD, occ, cyc, 256.0 * 4 * 64 * 4096 / (cyc / (ghz * 1e9)) / 1e12);
        if (V > 0) printf("   +%.1f cycles per VALU instruction", (cyc - base[occ - 1]) / V);
        printf("\n");
    }
}
int main() {
    float *in, *out;
    (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 4096 * 256 * 4);
    float h[4096]; unsigned x = 12345;
    for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) / 8388608.f - 1.f; }
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const double ghz = 2.4;
    run<0, 0, 0>(out, in, ghz);
    run<16, 0, 0>(out, in, ghz); run<48, 0, 0>(out, in, ghz); run<80, 0, 0>(out, in, ghz); run<112, 0, 0>(out, in, ghz);
    run<16, 1, 0>(out, in, ghz); run<48, 1, 0>(out, in, ghz);
    run<32, 2, 0>(out, in, ghz); run<32, 3, 0>(out, in, ghz);
    run<16, 0, 1>(out, in, ghz); run<32, 0, 1>(out, in, ghz); run<64, 0, 1>(out, in, ghz);
    return 0;
}
