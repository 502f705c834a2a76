// Reparametrisation + log q(z|x) (fwd/bwd) and the NOTEARS acyclicity term
// h = tr(expm(A∘A)) - 64 with its gradient.  gfx950 / wave64.
#include "common.h"

namespace cmvae {

// ---------------------------------------------------------------- reparam + log q
// 16 lanes x float4 cover one 64-float latent row; a wave covers 4 (row,m) pairs per step.
__global__ __launch_bounds__(256) void reparam_logq_fwd_kernel(const float* __restrict__ mean,
                                                               const float* __restrict__ logvar,
                                                               const float* __restrict__ eps, long total_rows, int M,
                                                               float* __restrict__ z, float* __restrict__ logq) {
    const int q = threadIdx.x & 15;
    const long stride = (long)gridDim.x * 16;
    for (long rm = (long)blockIdx.x * 16 + (threadIdx.x >> 4); rm < total_rows; rm += stride) {
        const long r = rm / M;
        const float4 mu = reinterpret_cast<const float4*>(mean + r * D)[q];
        const float4 lv = reinterpret_cast<const float4*>(logvar + r * D)[q];
        const float4 e = reinterpret_cast<const float4*>(eps + rm * D)[q];
        float4 zz;
        zz.x = fmaf(e.x, expf(0.5f * lv.x), mu.x);
        zz.y = fmaf(e.y, expf(0.5f * lv.y), mu.y);
        zz.z = fmaf(e.z, expf(0.5f * lv.z), mu.z);
        zz.w = fmaf(e.w, expf(0.5f * lv.w), mu.w);
        reinterpret_cast<float4*>(z + rm * D)[q] = zz;
        if (logq) {
            float dx = zz.x - mu.x, dy = zz.y - mu.y, dz = zz.z - mu.z, dw = zz.w - mu.w;
            float p = -0.5f * (lv.x + dx * dx / expf(lv.x)) + LOG_NORM;
            p += -0.5f * (lv.y + dy * dy / expf(lv.y)) + LOG_NORM;
            p += -0.5f * (lv.z + dz * dz / expf(lv.z)) + LOG_NORM;
            p += -0.5f * (lv.w + dw * dw / expf(lv.w)) + LOG_NORM;
            p += __shfl_xor(p, 1);
            p += __shfl_xor(p, 2);
            p += __shfl_xor(p, 4);
            p += __shfl_xor(p, 8);
            if (q == 0) logq[rm] = p;
        }
    }
}

// one wave per row r: lane = latent dim
__global__ __launch_bounds__(256) void reparam_logq_bwd_kernel(const float* __restrict__ logvar,
                                                               const float* __restrict__ eps,
                                                               const float* __restrict__ gz,
                                                               const float* __restrict__ glogq, int rows, int M,
                                                               float* __restrict__ dmean,
                                                               float* __restrict__ dlogvar) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float hs = 0.5f * expf(0.5f * logvar[(size_t)r * D + lane]);
    float am = 0.f, av = 0.f, ag = 0.f;
    for (int m = 0; m < M; ++m) {
        const size_t o = ((size_t)r * M + m) * D + lane;
        const float g = gz ? gz[o] : 0.f;
        am += g;
        av = fmaf(g * eps[o], hs, av);
        if (glogq) ag += glogq[(size_t)r * M + m];
    }
    dmean[(size_t)r * D + lane] = am;
    dlogvar[(size_t)r * D + lane] = av - 0.5f * ag;
}

// ---------------------------------------------------------------- expm trace
constexpr int XL = 68;

// C = A * B (64x64, LDS, leading dim XL); 4x4 register tile per thread; `post` applied per element
template <typename F>
__device__ __forceinline__ void mm64(const float* A, const float* B, float* C, F post) {
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    float acc[4][4] = {};
    for (int k = 0; k < D; ++k) {
        const float4 b = *reinterpret_cast<const float4*>(&B[k * XL + 4 * tj]);
        const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float a = A[(4 * ti + p) * XL + k];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[p][q] = fmaf(a, bv[q], acc[p][q]);
        }
    }
    __syncthreads();   // C may alias an operand of a previous step only after everyone has read
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 4 * ti + p, j = 4 * tj + q;
            C[i * XL + j] = post(acc[p][q], i, j);
        }
    __syncthreads();
}

__global__ __launch_bounds__(256) void expm_trace_kernel(const float* __restrict__ Ain, float* __restrict__ h,
                                                         float* __restrict__ grad) {
    __shared__ __align__(16) float X[D * XL];
    __shared__ __align__(16) float P[D * XL];
    __shared__ __align__(16) float T[D * XL];
    __shared__ float colsum[D];
    __shared__ float snorm;
    const int tid = threadIdx.x;
    for (int e = tid; e < D * D; e += 256) {
        const float a = Ain[e];
        X[(e >> 6) * XL + (e & 63)] = a * a;
    }
    __syncthreads();
    if (tid < D) {
        float s = 0.f;
        for (int i = 0; i < D; ++i) s += fabsf(X[i * XL + tid]);
        colsum[tid] = s;
        float m = s;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (tid == 0) snorm = m;
    }
    __syncthreads();
    int s = 0;
    {
        float n = snorm;
        while (n > 0.25f && s < 60) { n *= 0.5f; ++s; }
    }
    const float scale = ldexpf(1.f, -s);
    for (int e = tid; e < D * D; e += 256) {
        const int i = e >> 6, j = e & 63;
        const float x = X[i * XL + j] * scale;
        X[i * XL + j] = x;
        P[i * XL + j] = (i == j ? 1.f : 0.f) + x * 0.125f;   // I + X/8
    }
    __syncthreads();
    // Horner: P <- I + (X P)/k, k = 7..2 ; E = X P
    float* cur = P;
    float* nxt = T;
    for (int k = 7; k >= 2; --k) {
        const float ik = 1.f / (float)k;
        mm64(X, cur, nxt, [ik](float v, int i, int j) { return (i == j ? 1.f : 0.f) + v * ik; });
        float* t = cur; cur = nxt; nxt = t;
    }
    mm64(X, cur, nxt, [](float v, int, int) { return v; });   // E = expm(X) - I
    { float* t = cur; cur = nxt; nxt = t; }
    // undo the scaling: (I+E)^2 - I = 2E + E E
    for (int it = 0; it < s; ++it) {
        const float* E = cur;
        mm64(E, E, nxt, [E](float v, int i, int j) { return fmaf(2.f, E[i * XL + j], v); });
        float* t = cur; cur = nxt; nxt = t;
    }
    if (tid < D) {
        float d = cur[tid * XL + tid];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
        if (tid == 0) h[0] = d;
    }
    if (grad) {
        for (int e = tid; e < D * D; e += 256) {
            const int i = e >> 6, j = e & 63;
            grad[e] = ((i == j ? 1.f : 0.f) + cur[j * XL + i]) * (2.f * Ain[e]);
        }
    }
}

}  // namespace cmvae

using namespace cmvae;

extern "C" int cmvae_reparam_logq_fwd(const float* mean, const float* logvar, const float* eps, int rows, int M,
                                      float* z, float* logq, void* stream) {
    CMVAE_REQUIRE(mean && logvar && eps && z, "cmvae_reparam_logq_fwd: NULL tensor");
    CMVAE_REQUIRE(rows >= 0 && M > 0, "cmvae_reparam_logq_fwd: bad shape");
    const long total = (long)rows * M;
    if (total == 0) return 0;
    long blocks = (total + 15) / 16;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reparam_logq_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mean, logvar,
                       eps, total, M, z, logq);
    return check_launch("reparam_logq_fwd_kernel");
}

extern "C" int cmvae_reparam_logq_bwd(const float* logvar, const float* eps, const float* gz, const float* glogq,
                                      int rows, int M, float* dmean, float* dlogvar, void* stream) {
    CMVAE_REQUIRE(logvar && eps && dmean && dlogvar, "cmvae_reparam_logq_bwd: NULL tensor");
    CMVAE_REQUIRE(rows >= 0 && M > 0, "cmvae_reparam_logq_bwd: bad shape");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(reparam_logq_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logvar, eps,
                       gz, glogq, rows, M, dmean, dlogvar);
    return check_launch("reparam_logq_bwd_kernel");
}

extern "C" int cmvae_expm_trace(const float* A, float* h, float* grad, void* stream) {
    CMVAE_REQUIRE(A && h, "cmvae_expm_trace: NULL tensor");
    hipLaunchKernelGGL(expm_trace_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, A, h, grad);
    return check_launch("expm_trace_kernel");
}
