// Softmax attention core of the Set-Transformer MAB (omniglot/module.py:92-99), heads of 64 channels:
//   O[b,i,h,:] = Q[b,i,h,:] + sum_j softmax_j(scale * Q[b,i,h,:].K[b,j,h,:]) V[b,j,h,:]
// Q, K, V, O are [B, S, H*64] row-major (the head is a 64-column slice of the row).  One workgroup owns 16
// query rows (forward, dQ) or 16 key rows (dK/dV) of one (episode, head); the S x 16 score strip lives in LDS.
// Sets are small (S = 5 ... a few hundred), so these are latency-bound VALU kernels.
#include "common.h"

namespace cmvae {

constexpr int AT = 16;            // rows per workgroup
constexpr int ATT_MAX_S = 1024;   // 2 strips * 16 * 1024 * 4 B = 128 KB of LDS in the backward kernels

__device__ __forceinline__ float dot64(const float4* __restrict__ a_lds, const float4 (&b)[16]) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
        const float4 a0 = a_lds[c], a1 = a_lds[c + 1];
        s0 = fmaf(a0.x, b[c].x, fmaf(a0.y, b[c].y, fmaf(a0.z, b[c].z, fmaf(a0.w, b[c].w, s0))));
        s1 = fmaf(a1.x, b[c + 1].x, fmaf(a1.y, b[c + 1].y, fmaf(a1.z, b[c + 1].z, fmaf(a1.w, b[c + 1].w, s1))));
    }
    return s0 + s1;
}

__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

__device__ __forceinline__ void load_row16(const float* __restrict__ p, float4 (&r)[16]) {
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = reinterpret_cast<const float4*>(p)[c];
}

// ---------------------------------------------------------------------------- forward
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                       const float* __restrict__ V, int S, int C, float scale,
                                                       float* __restrict__ O, float* __restrict__ lse) {
    extern __shared__ __align__(16) float smem[];
    float* qs = smem;                    // [16][64] scaled queries
    float* sc = smem + AT * 64;          // [16][SP]
    const int SP = (S + 3) & ~3;
    const int t = threadIdx.x, r = t >> 4, c4 = t & 15;
    const int s0 = blockIdx.x * AT, h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
    const long base = (long)b * S * C + h * 64;
    const bool rv = s0 + r < S;
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rv) qv = *reinterpret_cast<const float4*>(Q + base + (long)(s0 + r) * C + 4 * c4);
    *reinterpret_cast<float4*>(qs + r * 64 + 4 * c4) = make_float4(qv.x * scale, qv.y * scale, qv.z * scale, qv.w * scale);
    __syncthreads();
    for (int j = t; j < S; j += 256) {
        float4 kr[16];
        load_row16(K + base + (long)j * C, kr);
#pragma unroll
        for (int rr = 0; rr < AT; ++rr) sc[rr * SP + j] = dot64(reinterpret_cast<const float4*>(qs + rr * 64), kr);
    }
    __syncthreads();
    {   // softmax of row r by its 16 lanes
        float m = -INFINITY;
        for (int j = c4; j < S; j += 16) m = fmaxf(m, sc[r * SP + j]);
        m = group16_max(m);
        float sum = 0.f;
        for (int j = c4; j < S; j += 16) {
            const float e = expf(sc[r * SP + j] - m);
            sc[r * SP + j] = e;
            sum += e;
        }
        sum = group16_sum(sum);
        const float inv = 1.f / sum;
        for (int j = c4; j < S; j += 16) sc[r * SP + j] *= inv;
        if (c4 == 0 && rv) lse[((long)b * H + h) * S + s0 + r] = m + logf(sum);
    }
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < S; ++j) {
        const float p = sc[r * SP + j];
        const float4 vv = *reinterpret_cast<const float4*>(V + base + (long)j * C + 4 * c4);
        acc.x = fmaf(p, vv.x, acc.x); acc.y = fmaf(p, vv.y, acc.y);
        acc.z = fmaf(p, vv.z, acc.z); acc.w = fmaf(p, vv.w, acc.w);
    }
    if (rv)
        *reinterpret_cast<float4*>(O + base + (long)(s0 + r) * C + 4 * c4) =
            make_float4(qv.x + acc.x, qv.y + acc.y, qv.z + acc.z, qv.w + acc.w);
}

// ---------------------------------------------------------------------------- backward, query side
// dQ = dO + scale * dS K,  dS = P * (dP - delta),  dP = dO V^T,  delta_i = sum_j P_ij dP_ij (written out)
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                          const float* __restrict__ V,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ dO, int S, int C, float scale,
                                                          float* __restrict__ dQ, float* __restrict__ delta) {
    extern __shared__ __align__(16) float smem[];
    float* qs = smem;                    // [16][64] scaled queries
    float* ds = smem + AT * 64;          // [16][64] dO rows
    float* sp = smem + 2 * AT * 64;      // [16][SP] P, then dS
    const int SP = (S + 3) & ~3;
    float* sd = sp + AT * SP;            // [16][SP] dP
    const int t = threadIdx.x, r = t >> 4, c4 = t & 15;
    const int s0 = blockIdx.x * AT, h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
    const long base = (long)b * S * C + h * 64;
    const bool rv = s0 + r < S;
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv;
    if (rv) {
        qv = *reinterpret_cast<const float4*>(Q + base + (long)(s0 + r) * C + 4 * c4);
        gv = *reinterpret_cast<const float4*>(dO + base + (long)(s0 + r) * C + 4 * c4);
    }
    *reinterpret_cast<float4*>(qs + r * 64 + 4 * c4) = make_float4(qv.x * scale, qv.y * scale, qv.z * scale, qv.w * scale);
    *reinterpret_cast<float4*>(ds + r * 64 + 4 * c4) = gv;
    __shared__ float row_lse[AT];
    if (t < AT) row_lse[t] = s0 + t < S ? lse[((long)b * H + h) * S + s0 + t] : 0.f;
    __syncthreads();
    for (int j = t; j < S; j += 256) {
        float4 kr[16];
        load_row16(K + base + (long)j * C, kr);
#pragma unroll
        for (int rr = 0; rr < AT; ++rr)
            sp[rr * SP + j] = expf(dot64(reinterpret_cast<const float4*>(qs + rr * 64), kr) - row_lse[rr]);
        load_row16(V + base + (long)j * C, kr);
#pragma unroll
        for (int rr = 0; rr < AT; ++rr) sd[rr * SP + j] = dot64(reinterpret_cast<const float4*>(ds + rr * 64), kr);
    }
    __syncthreads();
    {
        float d = 0.f;
        for (int j = c4; j < S; j += 16) d = fmaf(sp[r * SP + j], sd[r * SP + j], d);
        d = group16_sum(d);
        for (int j = c4; j < S; j += 16) sp[r * SP + j] *= sd[r * SP + j] - d;
        if (c4 == 0 && rv) delta[((long)b * H + h) * S + s0 + r] = d;
    }
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < S; ++j) {
        const float p = sp[r * SP + j];
        const float4 kk = *reinterpret_cast<const float4*>(K + base + (long)j * C + 4 * c4);
        acc.x = fmaf(p, kk.x, acc.x); acc.y = fmaf(p, kk.y, acc.y);
        acc.z = fmaf(p, kk.z, acc.z); acc.w = fmaf(p, kk.w, acc.w);
    }
    if (rv)
        *reinterpret_cast<float4*>(dQ + base + (long)(s0 + r) * C + 4 * c4) =
            make_float4(fmaf(scale, acc.x, gv.x), fmaf(scale, acc.y, gv.y), fmaf(scale, acc.z, gv.z),
                        fmaf(scale, acc.w, gv.w));
}

// ---------------------------------------------------------------------------- backward, key side
// dV_j = sum_i P_ij dO_i,  dK_j = scale * sum_i dS_ij Q_i      (16 key rows per workgroup)
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta,
                                                           const float* __restrict__ dO, int S, int C, float scale,
                                                           float* __restrict__ dK, float* __restrict__ dV) {
    extern __shared__ __align__(16) float smem[];
    float* ks = smem;                    // [16][64] keys, pre-scaled
    float* vs = smem + AT * 64;          // [16][64] values
    float* pt = smem + 2 * AT * 64;      // [16][SP] P^T
    const int SP = (S + 3) & ~3;
    float* st = pt + AT * SP;            // [16][SP] dS^T
    const int t = threadIdx.x, r = t >> 4, c4 = t & 15;
    const int j0 = blockIdx.x * AT, h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
    const long base = (long)b * S * C + h * 64;
    const bool rv = j0 + r < S;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (rv) {
        kv = *reinterpret_cast<const float4*>(K + base + (long)(j0 + r) * C + 4 * c4);
        vv = *reinterpret_cast<const float4*>(V + base + (long)(j0 + r) * C + 4 * c4);
    }
    *reinterpret_cast<float4*>(ks + r * 64 + 4 * c4) = make_float4(kv.x * scale, kv.y * scale, kv.z * scale, kv.w * scale);
    *reinterpret_cast<float4*>(vs + r * 64 + 4 * c4) = vv;
    __syncthreads();
    const long sbase = ((long)b * H + h) * S;
    for (int i = t; i < S; i += 256) {
        float4 row[16];
        float p[AT];
        const float l = lse[sbase + i], d = delta[sbase + i];
        load_row16(Q + base + (long)i * C, row);
#pragma unroll
        for (int jj = 0; jj < AT; ++jj) p[jj] = expf(dot64(reinterpret_cast<const float4*>(ks + jj * 64), row) - l);
        load_row16(dO + base + (long)i * C, row);
#pragma unroll
        for (int jj = 0; jj < AT; ++jj) {
            const float dp = dot64(reinterpret_cast<const float4*>(vs + jj * 64), row);
            pt[jj * SP + i] = p[jj];
            st[jj * SP + i] = p[jj] * (dp - d);
        }
    }
    __syncthreads();
    float4 av = make_float4(0.f, 0.f, 0.f, 0.f), ak = av;
    for (int i = 0; i < S; ++i) {
        const float p = pt[r * SP + i], s = st[r * SP + i];
        const float4 g = *reinterpret_cast<const float4*>(dO + base + (long)i * C + 4 * c4);
        const float4 q = *reinterpret_cast<const float4*>(Q + base + (long)i * C + 4 * c4);
        av.x = fmaf(p, g.x, av.x); av.y = fmaf(p, g.y, av.y); av.z = fmaf(p, g.z, av.z); av.w = fmaf(p, g.w, av.w);
        ak.x = fmaf(s, q.x, ak.x); ak.y = fmaf(s, q.y, ak.y); ak.z = fmaf(s, q.z, ak.z); ak.w = fmaf(s, q.w, ak.w);
    }
    if (rv) {
        *reinterpret_cast<float4*>(dV + base + (long)(j0 + r) * C + 4 * c4) = av;
        *reinterpret_cast<float4*>(dK + base + (long)(j0 + r) * C + 4 * c4) =
            make_float4(scale * ak.x, scale * ak.y, scale * ak.z, scale * ak.w);
    }
}

static int check_attn(const char* who, int B, int S, int H) {
    CMVAE_REQUIRE(B >= 1 && S >= 1 && H >= 1 && S <= ATT_MAX_S && B <= 65535 && H <= 65535,
                  "%s: B=%d S=%d heads=%d (set size must be 1..%d)", who, B, S, H, ATT_MAX_S);
    return 0;
}

// score strips above 64 KB (S > ~450) need the dynamic-LDS opt-in, once per process
static void allow_large_lds() {
    static bool done = false;
    if (done) return;
    const int cap = 144 * 1024;
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipGetLastError();
    done = true;
}

}  // namespace cmvae

using namespace cmvae;

extern "C" int cmvae_attention_fwd(const float* Q, const float* K, const float* V, int B, int S, int heads,
                                   float scale, float* O, float* lse, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(Q && K && V && O && lse, "cmvae_attention_fwd: NULL argument");
    if (int e = check_attn("cmvae_attention_fwd", B, S, heads)) return e;
    const int SP = (S + 3) & ~3;
    const size_t lds = (size_t)(AT * 64 + AT * SP) * sizeof(float);
    allow_large_lds();
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((S + AT - 1) / AT, heads, B), dim3(256), lds, (hipStream_t)stream, Q, K, V,
                       S, heads * 64, scale, O, lse);
    return check_launch("attn_fwd_kernel");
}

extern "C" int cmvae_attention_bwd(const float* Q, const float* K, const float* V, const float* lse, const float* dO,
                                   int B, int S, int heads, float scale, float* dQ, float* dK, float* dV,
                                   float* delta, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(Q && K && V && lse && dO && dQ && dK && dV && delta, "cmvae_attention_bwd: NULL argument");
    if (int e = check_attn("cmvae_attention_bwd", B, S, heads)) return e;
    const int SP = (S + 3) & ~3;
    const size_t lds = (size_t)(2 * AT * 64 + 2 * AT * SP) * sizeof(float);
    allow_large_lds();
    const dim3 grid((S + AT - 1) / AT, heads, B);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), lds, (hipStream_t)stream, Q, K, V, lse, dO, S,
                       heads * 64, scale, dQ, delta);
    if (int e = check_launch("attn_bwd_dq_kernel")) return e;
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), lds, (hipStream_t)stream, Q, K, V, lse, delta, dO, S,
                       heads * 64, scale, dK, dV);
    return check_launch("attn_bwd_dkv_kernel");
}
