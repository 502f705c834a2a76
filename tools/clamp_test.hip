#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ floatx2 pk_fma_clamp(floatx2 a, floatx2 b, floatx2 c) {
    floatx2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__global__ void k(const float* x, const float* sc, const float* sh, float* out, float invS) {
    int i = threadIdx.x;
    floatx2 a = {x[2 * i], x[2 * i + 1]};
    floatx2 s = {sc[0] * invS, sc[1] * invS}, t = {sh[0] * invS, sh[1] * invS};
    floatx2 d = pk_fma_clamp(a, s, t);
    out[2 * i] = d.x; out[2 * i + 1] = d.y;
}
int main() {
    const int n = 64;
    float hx[2 * n], hs[2] = {1.5f, 0.75f}, ht[2] = {0.1f, -0.2f}, ho[2 * n];
    for (int i = 0; i < 2 * n; ++i) hx[i] = (i - 64) * 0.37f;
    float *x, *s, *t, *o;
    hipMalloc(&x, sizeof hx); hipMalloc(&s, 8); hipMalloc(&t, 8); hipMalloc(&o, sizeof ho);
    hipMemcpy(x, hx, sizeof hx, hipMemcpyHostToDevice); hipMemcpy(s, hs, 8, hipMemcpyHostToDevice); hipMemcpy(t, ht, 8, hipMemcpyHostToDevice);
    const float S = 1152921504606846976.0f;   // 2^60
    k<<<1, n>>>(x, s, t, o, 1.0f / S);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2 * n; ++i) {
        float ref = fmaf(hs[i & 1], hx[i], ht[i & 1]);
        ref = ref > 0.f ? ref : 0.f;
        float got = ho[i] * S;
        if (got != ref) { if (bad < 5) printf("mismatch %d: x %g ref %g got %g\n", i, hx[i], ref, got); ++bad; }
    }
    printf("pk_fma clamp relu trick: %d mismatches of %d\n", bad, 2 * n);
    return bad != 0;
}
