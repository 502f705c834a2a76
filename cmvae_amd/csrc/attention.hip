// Softmax attention core of the Set-Transformer MAB (omniglot/module.py:92-99), heads of 64 channels:
//   O[b,i,h,:] = Q[b,i,h,:] + sum_j softmax_j(scale * Q[b,i,h,:].K[b,j,h,:]) V[b,j,h,:]
// Q, K, V, O are [B, S, H*64] row-major (the head is a 64-column slice of the row); Q, K, V (and dQ, dK, dV) may be
// column slices of wider rows — the three thirds of a packed [B, S, 3*H*64] projection — with row pitch P.  One workgroup owns 16
// query rows (forward, dQ) or 16 key rows (dK/dV) of one (episode, head); the S x 16 score strip lives in LDS.
// Both contractions of every kernel (scores = tile . M^T into the strip, output = strip . M) run on
// v_mfma_f32_16x16x4_f32; softmax and the dS algebra stay on the VALU, 16 lanes per row.
#include "common.h"

namespace cmvae {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int AT = 16;            // rows per workgroup
constexpr int ATT_MAX_S = 1024;   // 2 strips * 16 * 1028 * 4 B + 2 tiles = 137 KB of LDS in the backward kernels

// pitch of a 16 x S strip in LDS: S rounded up to whole 16-column MFMA tiles, + 4 so that the 16 rows of an
// A-fragment read fall into different banks
__host__ __device__ inline int strip_pitch(int S) { return ((S + 15) / 16) * 16 + 4; }

__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

// ---------------------------------------------------------------------------- the two MFMA building blocks
// (v_mfma_f32_16x16x4_f32: A[m = lane&15][k = lane>>4], B[k = lane>>4][n = lane&15], C[row = 4*(lane>>4) + reg]
//  [col = lane&15]; the contraction index may be permuted freely as long as A and B agree, so lane group g takes
//  channels 16g .. 16g+15: its B operand is 64 contiguous bytes of a global row.)
//
// strip_mfma: T1[m][j] = tile1[m][:] . M1[j][:]  (and T2 from tile2, M2) for all j < S, 16 columns per wave step;
// tile* are [16][64] in LDS, M* global matrices with row strides C1, C2.  emit(m, j, t1, t2) is called for every element
// of the 16-column tiles (j may be >= S in the last tile: the caller stores zeros there).
template <bool DUAL, typename F>
__device__ __forceinline__ void strip_mfma(const float* tile1, const float* __restrict__ M1, const float* tile2,
                                           const float* __restrict__ M2, int S, int C1, int C2, F&& emit) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    float a1[16], a2[DUAL ? 16 : 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(tile1 + n * 64 + 16 * g + 4 * c);
        a1[4 * c] = v.x; a1[4 * c + 1] = v.y; a1[4 * c + 2] = v.z; a1[4 * c + 3] = v.w;
        if (DUAL) {
            const float4 u = *reinterpret_cast<const float4*>(tile2 + n * 64 + 16 * g + 4 * c);
            a2[4 * c] = u.x; a2[4 * c + 1] = u.y; a2[4 * c + 2] = u.z; a2[4 * c + 3] = u.w;
        }
    }
    const int ntile = (S + 15) / 16;
    for (int jt = w; jt < ntile; jt += 4) {
        const int j = 16 * jt + n, jc = j < S ? j : S - 1;
        float4 b1[4], b2[DUAL ? 4 : 1];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            b1[c] = *reinterpret_cast<const float4*>(M1 + (long)jc * C1 + 16 * g + 4 * c);
            if (DUAL) b2[c] = *reinterpret_cast<const float4*>(M2 + (long)jc * C2 + 16 * g + 4 * c);
        }
        floatx4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0, r0 = p0, r1 = p0;     // two chains per product
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * c], b1[c].x, p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * c + 1], b1[c].y, p1, 0, 0, 0);
            if (DUAL) {
                r0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[4 * c], b2[c].x, r0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[4 * c + 1], b2[c].y, r1, 0, 0, 0);
            }
            p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * c + 2], b1[c].z, p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * c + 3], b1[c].w, p1, 0, 0, 0);
            if (DUAL) {
                r0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[4 * c + 2], b2[c].z, r0, 0, 0, 0);
                r1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[4 * c + 3], b2[c].w, r1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) emit(4 * g + reg, j, p0[reg] + p1[reg], DUAL ? r0[reg] + r1[reg] : 0.f);
    }
}

// out_mfma: O1[m][c] = sum_{j < S} strip1[m][j] * M1[j][c]  (and O2 from strip2, M2); wave w owns channels
// 16w .. 16w+15.  Strip columns S .. 16*ceil(S/16)-1 must hold zeros.  Returns the accumulators:
// element reg <-> (row 4*(lane>>4) + reg, channel 16w + (lane&15)).
template <bool DUAL>
__device__ __forceinline__ void out_mfma(const float* strip1, const float* __restrict__ M1, const float* strip2,
                                         const float* __restrict__ M2, int S, int C1, int C2, floatx4& o1,
                                         floatx4& o2) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int SPP = strip_pitch(S), ntile = (S + 15) / 16;
    floatx4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0, r0 = p0, r1 = p0;
    const float* s1 = strip1 + n * SPP + g;
    const float* s2 = DUAL ? strip2 + n * SPP + g : nullptr;
    const float* m1 = M1 + 16 * w + n;
    const float* m2 = DUAL ? M2 + 16 * w + n : nullptr;
    for (int jt = 0; jt < ntile; ++jt) {
        float a1[4], b1[4], a2[4], b2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                     // four k-steps of four rows: all operands requested first
            const int j = 16 * jt + 4 * q + g;
            const long jr = j < S ? j : S - 1;
            a1[q] = s1[16 * jt + 4 * q];
            b1[q] = m1[jr * C1];
            if (DUAL) { a2[q] = s2[16 * jt + 4 * q]; b2[q] = m2[jr * C2]; }
        }
        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0], b1[0], p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1], b1[1], p1, 0, 0, 0);
        if (DUAL) {
            r0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[0], b2[0], r0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[1], b2[1], r1, 0, 0, 0);
        }
        p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[2], b1[2], p0, 0, 0, 0);
        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[3], b1[3], p1, 0, 0, 0);
        if (DUAL) {
            r0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[2], b2[2], r0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[3], b2[3], r1, 0, 0, 0);
        }
    }
    o1 = p0 + p1;
    if (DUAL) o2 = r0 + r1;
}

// ---------------------------------------------------------------------------- forward
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                       const float* __restrict__ V, int S, int C, int P,
                                                       float scale, float* __restrict__ O,
                                                       float* __restrict__ lse) {
    extern __shared__ __align__(16) float smem[];
    float* qs = smem;                    // [16][64] scaled queries
    float* sc = smem + AT * 64;          // [16][SPP] scores, then probabilities
    const int SPP = strip_pitch(S);
    const int t = threadIdx.x, r = t >> 4, c4 = t & 15;
    const int s0 = blockIdx.x * AT, h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
    const long base = (long)b * S * C + h * 64, pbase = (long)b * S * P + h * 64;
    const bool rv = s0 + r < S;
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rv) qv = *reinterpret_cast<const float4*>(Q + pbase + (long)(s0 + r) * P + 4 * c4);
    *reinterpret_cast<float4*>(qs + r * 64 + 4 * c4) = make_float4(qv.x * scale, qv.y * scale, qv.z * scale, qv.w * scale);
    __syncthreads();
    strip_mfma<false>(qs, K + pbase, nullptr, nullptr, S, P, P,
                      [&](int m, int j, float v, float) { sc[m * SPP + j] = j < S ? v : 0.f; });
    __syncthreads();
    {   // softmax of row r by its 16 lanes
        float m = -INFINITY;
        for (int j = c4; j < S; j += 16) m = fmaxf(m, sc[r * SPP + j]);
        m = group16_max(m);
        float sum = 0.f;
        for (int j = c4; j < S; j += 16) {
            const float e = expf(sc[r * SPP + j] - m);
            sc[r * SPP + j] = e;
            sum += e;
        }
        sum = group16_sum(sum);
        const float inv = 1.f / sum;
        for (int j = c4; j < S; j += 16) sc[r * SPP + j] *= inv;
        if (c4 == 0 && rv) lse[((long)b * H + h) * S + s0 + r] = m + logf(sum);
    }
    __syncthreads();
    floatx4 acc, unused;
    out_mfma<false>(sc, V + pbase, nullptr, nullptr, S, P, P, acc, unused);
    const int lane = t & 63, w = t >> 6, n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = s0 + 4 * g + reg;
        if (row < S) {
            O[base + (long)row * C + 16 * w + n] = Q[pbase + (long)row * P + 16 * w + n] + acc[reg];
        }
    }
}

// ---------------------------------------------------------------------------- backward, query side
// dQ = dO + scale * dS K,  dS = P * (dP - delta),  dP = dO V^T,  delta_i = sum_j P_ij dP_ij (written out)
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                          const float* __restrict__ V,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ dO, int S, int C, int P,
                                                          float scale, float* __restrict__ dQ,
                                                          float* __restrict__ delta) {
    extern __shared__ __align__(16) float smem[];
    float* qs = smem;                    // [16][64] scaled queries
    float* ds = smem + AT * 64;          // [16][64] dO rows
    float* sp = smem + 2 * AT * 64;      // [16][SPP] P, then dS
    const int SPP = strip_pitch(S);
    float* sd = sp + AT * SPP;           // [16][SPP] dP
    const int t = threadIdx.x, r = t >> 4, c4 = t & 15;
    const int s0 = blockIdx.x * AT, h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
    const long base = (long)b * S * C + h * 64, pbase = (long)b * S * P + h * 64;
    const bool rv = s0 + r < S;
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), gv = qv;
    if (rv) {
        qv = *reinterpret_cast<const float4*>(Q + pbase + (long)(s0 + r) * P + 4 * c4);
        gv = *reinterpret_cast<const float4*>(dO + base + (long)(s0 + r) * C + 4 * c4);
    }
    *reinterpret_cast<float4*>(qs + r * 64 + 4 * c4) = make_float4(qv.x * scale, qv.y * scale, qv.z * scale, qv.w * scale);
    *reinterpret_cast<float4*>(ds + r * 64 + 4 * c4) = gv;
    __shared__ float row_lse[AT];
    if (t < AT) row_lse[t] = s0 + t < S ? lse[((long)b * H + h) * S + s0 + t] : 0.f;
    __syncthreads();
    strip_mfma<true>(qs, K + pbase, ds, V + pbase, S, P, P, [&](int m, int j, float v1, float v2) {
        const bool in = j < S;
        sp[m * SPP + j] = in ? expf(v1 - row_lse[m]) : 0.f;
        sd[m * SPP + j] = in ? v2 : 0.f;
    });
    __syncthreads();
    {
        float d = 0.f;
        for (int j = c4; j < S; j += 16) d = fmaf(sp[r * SPP + j], sd[r * SPP + j], d);
        d = group16_sum(d);
        for (int j = c4; j < S; j += 16) sp[r * SPP + j] *= sd[r * SPP + j] - d;
        if (c4 == 0 && rv) delta[((long)b * H + h) * S + s0 + r] = d;
    }
    __syncthreads();
    floatx4 acc, unused;
    out_mfma<false>(sp, K + pbase, nullptr, nullptr, S, P, P, acc, unused);
    const int lane = t & 63, w = t >> 6, n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = s0 + 4 * g + reg;
        if (row < S) {
            dQ[pbase + (long)row * P + 16 * w + n] = fmaf(scale, acc[reg], dO[base + (long)row * C + 16 * w + n]);
        }
    }
}

// ---------------------------------------------------------------------------- backward, key side
// dV_j = sum_i P_ij dO_i,  dK_j = scale * sum_i dS_ij Q_i      (16 key rows per workgroup)
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta,
                                                           const float* __restrict__ dO, int S, int C, int P,
                                                           float scale, float* __restrict__ dK,
                                                           float* __restrict__ dV) {
    extern __shared__ __align__(16) float smem[];
    float* ks = smem;                    // [16][64] keys, pre-scaled
    float* vs = smem + AT * 64;          // [16][64] values
    float* pt = smem + 2 * AT * 64;      // [16][SPP] P^T
    const int SPP = strip_pitch(S);
    float* st = pt + AT * SPP;           // [16][SPP] dS^T
    const int t = threadIdx.x, r = t >> 4, c4 = t & 15;
    const int j0 = blockIdx.x * AT, h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
    const long base = (long)b * S * C + h * 64, pbase = (long)b * S * P + h * 64;
    const bool rv = j0 + r < S;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (rv) {
        kv = *reinterpret_cast<const float4*>(K + pbase + (long)(j0 + r) * P + 4 * c4);
        vv = *reinterpret_cast<const float4*>(V + pbase + (long)(j0 + r) * P + 4 * c4);
    }
    *reinterpret_cast<float4*>(ks + r * 64 + 4 * c4) = make_float4(kv.x * scale, kv.y * scale, kv.z * scale, kv.w * scale);
    *reinterpret_cast<float4*>(vs + r * 64 + 4 * c4) = vv;
    __syncthreads();
    const long sbase = ((long)b * H + h) * S;
    // transposed strips: row = key of this workgroup, column = query i
    strip_mfma<true>(ks, Q + pbase, vs, dO + base, S, P, C, [&](int m, int i, float v1, float v2) {
        const bool in = i < S;
        const int ic = in ? i : S - 1;
        const float p = in ? expf(v1 - lse[sbase + ic]) : 0.f;
        pt[m * SPP + i] = p;
        st[m * SPP + i] = in ? p * (v2 - delta[sbase + ic]) : 0.f;
    });
    __syncthreads();
    floatx4 av, ak;
    out_mfma<true>(pt, dO + base, st, Q + pbase, S, C, P, av, ak);
    const int lane = t & 63, w = t >> 6, n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = j0 + 4 * g + reg;
        if (row < S) {
            const long o = pbase + (long)row * P + 16 * w + n;
            dV[o] = av[reg];
            dK[o] = scale * ak[reg];
        }
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

static int attention_fwd(const float* Q, const float* K, const float* V, int pitch, int B, int S, int heads, float scale,
                         float* O, float* lse, void* stream) {
    const size_t lds = (size_t)(AT * 64 + AT * strip_pitch(S)) * sizeof(float);
    allow_large_lds();
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((S + AT - 1) / AT, heads, B), dim3(256), lds, (hipStream_t)stream, Q, K, V,
                       S, heads * 64, pitch, scale, O, lse);
    return check_launch("attn_fwd_kernel");
}

static int attention_bwd(const float* Q, const float* K, const float* V, int pitch, const float* lse, const float* dO,
                         int B, int S, int heads, float scale, float* dQ, float* dK, float* dV, float* delta,
                         void* stream) {
    const size_t lds = (size_t)(2 * AT * 64 + 2 * AT * strip_pitch(S)) * sizeof(float);
    allow_large_lds();
    const dim3 grid((S + AT - 1) / AT, heads, B);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), lds, (hipStream_t)stream, Q, K, V, lse, dO, S,
                       heads * 64, pitch, scale, dQ, delta);
    if (int e = check_launch("attn_bwd_dq_kernel")) return e;
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), lds, (hipStream_t)stream, Q, K, V, lse, delta, dO, S,
                       heads * 64, pitch, scale, dK, dV);
    return check_launch("attn_bwd_dkv_kernel");
}

extern "C" int cmvae_attention_fwd(const float* Q, const float* K, const float* V, int B, int S, int heads,
                                   float scale, float* O, float* lse, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(Q && K && V && O && lse, "cmvae_attention_fwd: NULL argument");
    if (int e = check_attn("cmvae_attention_fwd", B, S, heads)) return e;
    return attention_fwd(Q, K, V, heads * 64, B, S, heads, scale, O, lse, stream);
}

extern "C" int cmvae_attention_bwd(const float* Q, const float* K, const float* V, const float* lse, const float* dO,
                                   int B, int S, int heads, float scale, float* dQ, float* dK, float* dV,
                                   float* delta, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(Q && K && V && lse && dO && dQ && dK && dV && delta, "cmvae_attention_bwd: NULL argument");
    if (int e = check_attn("cmvae_attention_bwd", B, S, heads)) return e;
    return attention_bwd(Q, K, V, heads * 64, lse, dO, B, S, heads, scale, dQ, dK, dV, delta, stream);
}

// The same over a packed projection QKV [B, S, 3*heads*64] = [q | k | v] per row (one GEMM instead of three for the
// self-attention blocks, where fc_q, fc_k and fc_v read the same rows); dQKV has the same layout.
extern "C" int cmvae_attention_qkv_fwd(const float* QKV, int B, int S, int heads, float scale, float* O, float* lse,
                                       void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(QKV && O && lse, "cmvae_attention_qkv_fwd: NULL argument");
    if (int e = check_attn("cmvae_attention_qkv_fwd", B, S, heads)) return e;
    const int C = heads * 64;
    return attention_fwd(QKV, QKV + C, QKV + 2 * C, 3 * C, B, S, heads, scale, O, lse, stream);
}

extern "C" int cmvae_attention_qkv_bwd(const float* QKV, const float* lse, const float* dO, int B, int S, int heads,
                                       float scale, float* dQKV, float* delta, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(QKV && lse && dO && dQKV && delta, "cmvae_attention_qkv_bwd: NULL argument");
    if (int e = check_attn("cmvae_attention_qkv_bwd", B, S, heads)) return e;
    const int C = heads * 64;
    return attention_bwd(QKV, QKV + C, QKV + 2 * C, 3 * C, lse, dO, B, S, heads, scale, dQKV, dQKV + C, dQKV + 2 * C,
                         delta, stream);
}
