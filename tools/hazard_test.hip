// How many wait states does gfx950 need between a v_pk_fma_f32 (inline asm: the compiler's hazard recognizer does not
// see a VALU write there) and a v_mfma that reads the result as SrcA?  Both back to back in ONE asm block, N nops apart.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int N>
__global__ void k(const float* x, float* out) {
    floatx2 a = {x[threadIdx.x], x[64 + threadIdx.x]};
    floatx2 s = {2.f, 3.f}, t = {0.f, 0.f};
    float b = 1.f;
    floatx4 c = {0.f, 0.f, 0.f, 0.f};
    if (N == 0) asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n s_nop 7\n s_nop 7\n v_pk_fma_f32 v[10:11], %1, %2, %3\n v_mfma_f32_16x16x4_f32 %0, v10, %4, %0\n s_nop 7\n s_nop 7" : "+v"(c) : "v"(a), "v"(s), "v"(t), "v"(b) : "v10", "v11");
    if (N == 1) asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n s_nop 7\n s_nop 7\n v_pk_fma_f32 v[10:11], %1, %2, %3\n s_nop 0\n v_mfma_f32_16x16x4_f32 %0, v10, %4, %0\n s_nop 7\n s_nop 7" : "+v"(c) : "v"(a), "v"(s), "v"(t), "v"(b) : "v10", "v11");
    if (N == 2) asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n s_nop 7\n s_nop 7\n v_pk_fma_f32 v[10:11], %1, %2, %3\n s_nop 1\n v_mfma_f32_16x16x4_f32 %0, v10, %4, %0\n s_nop 7\n s_nop 7" : "+v"(c) : "v"(a), "v"(s), "v"(t), "v"(b) : "v10", "v11");
    if (N == 3) asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n s_nop 7\n s_nop 7\n v_pk_fma_f32 v[10:11], %1, %2, %3\n s_nop 2\n v_mfma_f32_16x16x4_f32 %0, v10, %4, %0\n s_nop 7\n s_nop 7" : "+v"(c) : "v"(a), "v"(s), "v"(t), "v"(b) : "v10", "v11");
    if (N == 4) asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n s_nop 7\n s_nop 7\n v_pk_fma_f32 v[10:11], %1, %2, %3\n s_nop 3\n v_mfma_f32_16x16x4_f32 %0, v10, %4, %0\n s_nop 7\n s_nop 7" : "+v"(c) : "v"(a), "v"(s), "v"(t), "v"(b) : "v10", "v11");
    if (N == 6) asm volatile("v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n s_nop 7\n s_nop 7\n v_pk_fma_f32 v[10:11], %1, %2, %3\n s_nop 5\n v_mfma_f32_16x16x4_f32 %0, v10, %4, %0\n s_nop 7\n s_nop 7" : "+v"(c) : "v"(a), "v"(s), "v"(t), "v"(b) : "v10", "v11");
    out[threadIdx.x * 4 + 0] = c[0]; out[threadIdx.x * 4 + 1] = c[1]; out[threadIdx.x * 4 + 2] = c[2]; out[threadIdx.x * 4 + 3] = c[3];
}
template <int N> void run(const float* x, float* o, const float* hx) {
    k<N><<<1, 64>>>(x, o);
    float ho[256]; hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    // D[m][n] = sum_k A[m][k] B[k][n], A[m=lane%16][k=lane/16] = 2*x[lane] (d.x), B = 1: D[m][*] = 2*sum_k x[m+16k]
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
        const int m = 4 * (lane / 16) + r;
        float ref = 0; for (int kk = 0; kk < 4; ++kk) ref += 2.f * hx[m + 16 * kk];
        if (fabsf(ho[lane * 4 + r] - ref) > 1e-4f * fabsf(ref) + 1e-5f) ++bad;
    }
    printf("wait states %d between v_pk_fma_f32 and the MFMA that reads it: %d wrong of 256\n", N, bad);
}
int main() {
    float hx[128]; for (int i = 0; i < 128; ++i) hx[i] = 1.f + 0.01f * i;
    float *x, *o; hipMalloc(&x, sizeof hx); hipMalloc(&o, 1024); hipMemcpy(x, hx, sizeof hx, hipMemcpyHostToDevice);
    run<0>(x, o, hx); run<1>(x, o, hx); run<2>(x, o, hx); run<3>(x, o, hx); run<4>(x, o, hx); run<6>(x, o, hx);
    return 0;
}
