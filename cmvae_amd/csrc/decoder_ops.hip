// HBM-bound companions of the conv GEMMs (gfx950): BatchNorm statistics / finalize / backward on
// channels-last [rows,64] activations, and the fused last decoder layer
// ConvTranspose2d(64,1,4,2,1) + Sigmoid + BCE (omniglot/module.py:62-63, model.py:463-466) with its
// backward (which also folds the ReLU mask and the BatchNorm-backward reductions of layer 3).
#include "common.h"

namespace cmvae {

constexpr int C = 64;

// ------------------------------------------------------------------ out[L] = sum_b part[b][L]
// 64 columns x 16 row groups per block; fixed summation order (reproducible)
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ part, int nblk, int L,
                                                             float* __restrict__ out) {
    __shared__ float red[16 * 64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + c;
    float s0 = 0.f, s1 = 0.f;
    if (e < L) {
        int b = g;
        // eight loads in flight per thread: the kernel is pure latency (two workgroups for the 128 BatchNorm sums)
        for (; b + 112 < nblk; b += 128) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = part[(long)(b + 16 * i) * L + e];
#pragma unroll
            for (int i = 0; i < 8; i += 2) { s0 += v[i]; s1 += v[i + 1]; }
        }
        for (; b + 16 < nblk; b += 32) {
            s0 += part[(long)b * L + e];
            s1 += part[(long)(b + 16) * L + e];
        }
        if (b < nblk) s0 += part[(long)b * L + e];
    }
    red[g * 64 + c] = s0 + s1;
    __syncthreads();
    if (g == 0 && e < L) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r * 64 + c];
        out[e] = s;
    }
}

// ------------------------------------------------------------------ BN forward finalize
// sums = {sum x[64], sum x^2[64]} over `count` values per channel (already all-reduced over ranks).
__global__ void bn_finalize_kernel(const float* __restrict__ sums, float count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_o,
                                   float* __restrict__ invstd_o) {
    const int c = threadIdx.x;
    if (c >= C) return;
    const float mean = sums[c] / count;
    float var = sums[C + c] / count - mean * mean;
    var = fmaxf(var, 0.f);
    const float invstd = rsqrtf(var + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    mean_o[c] = mean;
    invstd_o[c] = invstd;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count / (count - 1.f));
    }
}

// The two finalize steps straight from the per-workgroup partials [nblk][128] (single process: no all-reduce sits
// between the column sums and the finalize): one launch instead of colsum_reduce + finalize (+ the counter's add).
// 64 column pairs x 16 row groups, fixed summation order; sums[] is left in LDS for the 64 finalizing threads.
__device__ __forceinline__ void part_colsum_128(const float* __restrict__ part, int nblk, float* red, float* sums) {
    // thread = (column pair, one of 16 row groups); twelve 8-byte loads in flight per thread: the kernel is pure latency
    const int c2 = threadIdx.x & 63, g = threadIdx.x >> 6;
    const float2* p2 = reinterpret_cast<const float2*>(part) + c2;
    float2 s0 = make_float2(0.f, 0.f), s1 = s0;
    int b = g;
    for (; b + 16 * 11 < nblk; b += 16 * 12) {
        float2 v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = p2[(long)(b + 16 * i) * 64];
#pragma unroll
        for (int i = 0; i < 12; i += 2) { s0.x += v[i].x; s0.y += v[i].y; s1.x += v[i + 1].x; s1.y += v[i + 1].y; }
    }
    for (; b < nblk; b += 16) { const float2 v = p2[(long)b * 64]; s0.x += v.x; s0.y += v.y; }
    red[g * 128 + 2 * c2] = s0.x + s1.x;
    red[g * 128 + 2 * c2 + 1] = s0.y + s1.y;
    __syncthreads();
    if (threadIdx.x < 128) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r * 128 + threadIdx.x];
        sums[threadIdx.x] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void bn_finalize_part_kernel(const float* __restrict__ part, int nblk, float count,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps,
                                                                float momentum, float* __restrict__ running_mean,
                                                                float* __restrict__ running_var,
                                                                long long* __restrict__ num_batches_tracked,
                                                                float* __restrict__ scale, float* __restrict__ shift,
                                                                float* __restrict__ mean_o,
                                                                float* __restrict__ invstd_o) {
    __shared__ float red[16 * 128];
    __shared__ float sums[128];
    part_colsum_128(part, nblk, red, sums);
    const int c = threadIdx.x;
    if (c == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
    if (c >= C) return;
    const float mean = sums[c] / count;
    float var = sums[C + c] / count - mean * mean;
    var = fmaxf(var, 0.f);
    const float invstd = rsqrtf(var + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    mean_o[c] = mean;
    invstd_o[c] = invstd;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count / (count - 1.f));
    }
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_part_kernel(const float* __restrict__ part, int nblk,
                                                                    float count, float* __restrict__ c1,
                                                                    float* __restrict__ c2,
                                                                    float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta) {
    __shared__ float red[16 * 128];
    __shared__ float sums[128];
    part_colsum_128(part, nblk, red, sums);
    const int c = threadIdx.x;
    if (c >= C) return;
    c1[c] = sums[c] / count;
    c2[c] = sums[C + c] / count;
    dbeta[c] = sums[c];
    dgamma[c] = sums[C + c];
}

// eval mode: the fold from the running statistics (nn.BatchNorm2d.eval())
__global__ void bn_eval_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                    float eps, float* __restrict__ scale, float* __restrict__ shift,
                                    float* __restrict__ invstd_o) {
    const int c = threadIdx.x;
    if (c >= C) return;
    const float invstd = rsqrtf(running_var[c] + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - running_mean[c] * sc;
    invstd_o[c] = invstd;
}

// ------------------------------------------------------------------ BN + ReLU backward
// dybn = d_a * (y*scale+shift > 0);  part[blk] = {sum dybn[64], sum dybn*xhat[64]}
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const float* __restrict__ d_a,
                                                                 const float* __restrict__ y,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, long rows,
                                                                 int masked, float* __restrict__ part) {
    __shared__ float red[16 * 128];
    const int q = threadIdx.x & 15, rb = threadIdx.x >> 4;
    const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
    const float4 mu = reinterpret_cast<const float4*>(mean)[q], is = reinterpret_cast<const float4*>(invstd)[q];
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (long r = (long)blockIdx.x * 16 + rb; r < rows; r += (long)gridDim.x * 16) {
        float4 g = load_once(d_a + r * C + 4 * q);
        const float4 v = load_once(y + r * C + 4 * q);
        if (!masked) {
            g.x = fmaf(v.x, sc.x, sh.x) > 0.f ? g.x : 0.f;
            g.y = fmaf(v.y, sc.y, sh.y) > 0.f ? g.y : 0.f;
            g.z = fmaf(v.z, sc.z, sh.z) > 0.f ? g.z : 0.f;
            g.w = fmaf(v.w, sc.w, sh.w) > 0.f ? g.w : 0.f;
        }
        s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
        s2.x = fmaf(g.x, (v.x - mu.x) * is.x, s2.x);
        s2.y = fmaf(g.y, (v.y - mu.y) * is.y, s2.y);
        s2.z = fmaf(g.z, (v.z - mu.z) * is.z, s2.z);
        s2.w = fmaf(g.w, (v.w - mu.w) * is.w, s2.w);
    }
    *reinterpret_cast<float4*>(&red[rb * 128 + 4 * q]) = s1;
    *reinterpret_cast<float4*>(&red[rb * 128 + 64 + 4 * q]) = s2;
    __syncthreads();
    if (threadIdx.x < 128) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r * 128 + threadIdx.x];
        part[(long)blockIdx.x * 128 + threadIdx.x] = s;
    }
}

// sums_global (all-reduced) -> c1 = sum dybn / n, c2 = sum dybn*xhat / n ; local sums -> dbeta, dgamma
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ sums_local, const float* __restrict__ sums_global,
                                       float count, float* __restrict__ c1, float* __restrict__ c2,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = threadIdx.x;
    if (c >= C) return;
    c1[c] = sums_global[c] / count;
    c2[c] = sums_global[C + c] / count;
    dbeta[c] = sums_local[c];
    dgamma[c] = sums_local[C + c];
}

// dy = scale * (dybn - c1 - xhat*c2)      (scale = gamma*invstd)
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const float* __restrict__ d_a,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ c1,
                                                                const float* __restrict__ c2, long rows, int masked,
                                                                float* __restrict__ dy) {
    const int q = threadIdx.x & 15, rb = threadIdx.x >> 4;
    const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
    const float4 mu = reinterpret_cast<const float4*>(mean)[q], is = reinterpret_cast<const float4*>(invstd)[q];
    const float4 a1 = reinterpret_cast<const float4*>(c1)[q], a2 = reinterpret_cast<const float4*>(c2)[q];
    for (long r = (long)blockIdx.x * 16 + rb; r < rows; r += (long)gridDim.x * 16) {
        float4 g = reinterpret_cast<const float4*>(d_a + r * C)[q];
        const float4 v = reinterpret_cast<const float4*>(y + r * C)[q];
        if (!masked) {
            g.x = fmaf(v.x, sc.x, sh.x) > 0.f ? g.x : 0.f;
            g.y = fmaf(v.y, sc.y, sh.y) > 0.f ? g.y : 0.f;
            g.z = fmaf(v.z, sc.z, sh.z) > 0.f ? g.z : 0.f;
            g.w = fmaf(v.w, sc.w, sh.w) > 0.f ? g.w : 0.f;
        }
        float4 o;
        o.x = sc.x * (g.x - a1.x - (v.x - mu.x) * is.x * a2.x);
        o.y = sc.y * (g.y - a1.y - (v.y - mu.y) * is.y * a2.y);
        o.z = sc.z * (g.z - a1.z - (v.z - mu.z) * is.z * a2.z);
        o.w = sc.w * (g.w - a1.w - (v.w - mu.w) * is.w * a2.w);
        store_once(dy + r * C + 4 * q, o);
    }
}

// ------------------------------------------------------------------ last decoder layer, forward
// y3 [n,14,14,64] --relu(bn3)--> convT(64->1,k4,s2,p1) + b --sigmoid--> p [n,28,28];
// loss_part[img] = -sum_pix ( x*max(log p,-100) + (1-x)*max(log1p(-p),-100) ),  x = target[img / M]
//
// Per image: Y[tap][pix] = w4^T[tap][0:64] . a3^T[0:64][pix] on v_mfma_f32_16x16x4_f32 (16 taps x 16 pixels per
// accumulator, 13 blocks of 16 pixels per image, operand fragments straight from global memory); a lane then holds
// the four kx taps of one (pixel, ky) and stores them as 16 bytes into the pixel's slot of a zero-bordered 16 x 16
// grid in LDS, from which every output pixel reads its four (pixel, tap) terms at constant offsets -- no bounds
// tests, no index arithmetic per image.  One block barrier per image (Y is double-buffered).
// The kernel is bound by its instruction count (with every load an L2 hit it is 15 % faster, without the MFMAs'
// 130 us it would still take 300): everything that does not depend on the image is computed once.
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int HI = 14, HO = 28, NPIX_I = HI * HI, NPIX_O = HO * HO;
constexpr int RB = (NPIX_I + 15) / 16;  // 13 blocks of 16 pixels (12 padded rows)
constexpr int YP = 20;                 // floats per pixel slot: 16-byte stores and the gather's reads conflict-free
constexpr int YSLOTS = 256;            // slot of input pixel (iy, ix) = (iy + 1) * 16 + ix + 1; the border stays zero

// sum over the wave's 64 lanes, the same value on every lane's row leader path: four rotations within the rows of 16
// lanes (DPP, no LDS round trips), then the four row sums through scalar registers
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define CMVAE_ROR_ADD(n) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + n, 0xf, 0xf, false))
    CMVAE_ROR_ADD(8); CMVAE_ROR_ADD(4); CMVAE_ROR_ADD(2); CMVAE_ROR_ADD(1);
#undef CMVAE_ROR_ADD
    const int vi = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return (r0 + r1) + (r2 + r3);
}

__global__ __launch_bounds__(256, 3) void dec_out_fwd_kernel(const float* __restrict__ y3,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ w4,   // [64][16]
                                                             const float* __restrict__ b4,
                                                             const float* __restrict__ target, int M, int nimg,
                                                             float* __restrict__ p_out, float* __restrict__ loss_part) {
    __shared__ __align__(16) float ys[2][YSLOTS * YP];      // double-buffered: one block barrier per image
    __shared__ float lred[2][4];
    __shared__ __align__(16) float fold[2][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    // MFMA (t, j) contracts channel 16t + 4g + j on lane group g: the pixel operand of a 16-pixel block is one float4
    // per t straight from the pixel's global row (64 contiguous bytes per pixel and t over the four lane groups) -- the
    // activations never pass through LDS; BatchNorm + ReLU are applied on the registers with the clamped
    // one-instruction fold of conv_gemm.hip (relu(bn(y)) * 2^-60 against w4 * 2^60: same bits).
    float bw[16];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) bw[4 * t + j] = w4[(16 * t + 4 * g + j) * 16 + i] * 0x1p60f;
    // the fold constants of the lane's 16 channels would be 32 registers (the kernel runs three workgroups per CU on
    // 168): they are read from LDS per image instead, 8 x 16 bytes
    if (tid < 64) {
        fold[0][tid] = scale[tid] * 0x1p-60f;
        fold[1][tid] = shift[tid] * 0x1p-60f;
    }
    const float bias = b4[0];
    for (int e = tid; e < 2 * YSLOTS * YP; e += 256) (&ys[0][0])[e] = 0.f;
    // where the lane's accumulators go: pixel 16 blk + i, taps 4g .. 4g+3 (float index into a Y buffer)
    int sidx[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int px = 16 * (b < 3 ? w + 4 * b : 12) + i;
        sidx[b] = (px + 2 * (px / HI) + 17) * YP + 4 * g;
    }
    const bool tail_ok = i < NPIX_I - 16 * 12;       // the 13th block holds 4 pixels; its other rows are never stored
    // what output pixel o = tid + 256 k reads: input pixels (iy0 - a, ix0 - b) with taps (ky0 + 2a, kx0 + 2b);
    // gidx = slot(iy0 - 1, ix0 - 1) * YP + 4 ky0 + kx0, the four terms at + {17 YP, 16 YP + 2, YP + 8, 10}
    int gidx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o = tid + 256 * k < NPIX_O ? tid + 256 * k : 0;
        const int oy = o / HO, ox = o - oy * HO;
        gidx[k] = (((oy + 1) >> 1) * 16 + ((ox + 1) >> 1)) * YP + ((oy + 1) & 1) * 4 + ((ox + 1) & 1);
    }
    // row blocks w, w+4, w+8 of every image and the 13th block on wave (image count & 3): a fixed owner would be the
    // slowest wave of every image, and the first waves of the resident workgroups share a SIMD.  One image ahead in
    // registers (two images ahead at two workgroups per CU was measured: 518 vs 485 us -- the kernel is not waiting
    // for its loads)
    float4 pre[4][4];
    auto fetch = [&](int img, int nbf) {
#if defined(CMVAE_DEC_ABL) && CMVAE_DEC_ABL == 3      // timing only: the activations come from 32 images (L2 hits)
        const float* yi = y3 + (long)(img & 31) * NPIX_I * C + 4 * g;
#else
        const float* yi = y3 + (long)img * NPIX_I * C + 4 * g;
#endif
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b < nbf) {
                const int px = 16 * (b < 3 ? w + 4 * b : 12) + i;
                const float* pr = yi + (px < NPIX_I ? px : NPIX_I - 1) * C;
#pragma unroll
                for (int t = 0; t < 4; ++t) pre[b][t] = *reinterpret_cast<const float4*>(pr + 16 * t);
            }
        }
    };
    auto blocks_of = [&](int it) { return w == (it & 3) ? 4 : 3; };       // uniform per wave
    if ((int)blockIdx.x < nimg) fetch(blockIdx.x, blocks_of(0));
    __syncthreads();              // the borders are zero before any wave stores a pixel
    int buf = 0, it = 0;
    for (int img = blockIdx.x; img < nimg; img += gridDim.x, buf ^= 1, ++it) {
        const int nb = blocks_of(it);
        float* yb = ys[buf];
        typedef float floatx2 __attribute__((ext_vector_type(2)));
        floatx4 acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = floatx4{0.f, 0.f, 0.f, 0.f};
        int fo = 4 * g;
        asm volatile("" : "+v"(fo));      // per image: keeps the compiler from holding all 32 constants in registers
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float4 a[4];
            const float4 sct = *reinterpret_cast<const float4*>(&fold[0][16 * t + fo]);
            const float4 sht = *reinterpret_cast<const float4*>(&fold[1][16 * t + fo]);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b < nb) {
                    // both halves and the two wait states a v_mfma needs behind a VALU write in one block (the hazard
                    // recognizer does not look into inline asm)
                    const float4 v = pre[b][t];
                    floatx2 lo, hi;
                    asm("v_pk_fma_f32 %0, %2, %3, %4 clamp\n\tv_pk_fma_f32 %1, %5, %6, %7 clamp\n\ts_nop 1"
                        : "=&v"(lo), "=v"(hi)
                        : "v"(floatx2{v.x, v.y}), "v"(floatx2{sct.x, sct.y}), "v"(floatx2{sht.x, sht.y}),
                          "v"(floatx2{v.z, v.w}), "v"(floatx2{sct.z, sct.w}), "v"(floatx2{sht.z, sht.w}));
                    a[b] = make_float4(lo.x, lo.y, hi.x, hi.y);
                }
            }
            // independent accumulator chains interleaved, so a dependent MFMA never waits for its own chain
#pragma unroll
            for (int b = 0; b < 4; ++b) if (b < nb) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[4 * t + 0], a[b].x, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 4; ++b) if (b < nb) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[4 * t + 1], a[b].y, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 4; ++b) if (b < nb) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[4 * t + 2], a[b].z, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 4; ++b) if (b < nb) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[4 * t + 3], a[b].w, acc[b], 0, 0, 0);
        }
        if (img + (int)gridDim.x < nimg) fetch(img + gridDim.x, blocks_of(it + 1));
#pragma unroll
        for (int b = 0; b < 3; ++b) *reinterpret_cast<floatx4*>(&yb[sidx[b]]) = acc[b];      // taps (ky = g, kx = 0..3)
        if (nb == 4 && tail_ok) *reinterpret_cast<floatx4*>(&yb[sidx[3]]) = acc[3];
        __syncthreads();          // the image's Y is complete; the other buffer is free for the next image
        const float* tg = target + (long)(img / M) * NPIX_O + tid;
        float* po = p_out + (long)img * NPIX_O + tid;
        float lsum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k == 3 && tid >= NPIX_O - 3 * 256) break;
            const float* yq = yb + gidx[k];
            // same order of the four terms as always: (iy0, ix0), (iy0, ix0 - 1), (iy0 - 1, ix0), (iy0 - 1, ix0 - 1)
            float acc1 = bias + yq[17 * YP];
            acc1 += yq[16 * YP + 2];
            acc1 += yq[YP + 8];
            acc1 += yq[10];
            // hardware exp / log (1 ulp-class): the loss is a sum of 784 O(1) terms per image, the tolerance
            // 1e-4 relative; log(1-p) from the rounded p like BCELoss does (saturates to the -100 clamp)
#if defined(CMVAE_DEC_ABL) && CMVAE_DEC_ABL == 1      // timing only: no transcendentals
            const float p = acc1;
            po[256 * k] = p;
            lsum -= tg[256 * k] * p;
#else
            const float p = __frcp_rn(1.f + __expf(-acc1));
            po[256 * k] = p;
            const float x = tg[256 * k];
            lsum -= x * fmaxf(__logf(p), -100.f) + (1.f - x) * fmaxf(__logf(1.f - p), -100.f);
#endif
        }
        lsum = wave_sum_dpp(lsum);
        if (lane == 0) lred[buf][w] = lsum;
        // the wave that finishes last cannot be known without a barrier: the sum of the four is written one image
        // later, after the next image's barrier (and once more after the loop)
        if (img != (int)blockIdx.x && tid == 0) {
            const float* lr = lred[buf ^ 1];
            loss_part[img - gridDim.x] = (lr[0] + lr[1]) + (lr[2] + lr[3]);
        }
    }
    __syncthreads();
    if (tid == 0 && (int)blockIdx.x < nimg) {
        // last image this workgroup processed: index blockIdx.x + k*gridDim.x, in buffer buf ^ 1
        const int last = blockIdx.x + ((nimg - 1 - (int)blockIdx.x) / (int)gridDim.x) * (int)gridDim.x;
        const float* lr = lred[buf ^ 1];
        loss_part[last] = (lr[0] + lr[1]) + (lr[2] + lr[3]);
    }
}

// ------------------------------------------------------------------ last decoder layer, backward
// dlogit = gscale * (p - x) * p(1-p) / max(p(1-p), 1e-12)      (BCELoss backward through the sigmoid)
// dybn3[pix][ci] = mask * sum_tap dlogit[tap pos] * w4[ci][tap]     (mask: relu(bn3(y3)) > 0)
// part[blk] = { dW4[64][16], sum dybn3[64], sum dybn3*xhat[64], db4 }            (L = 1024 + 128 + 1)
//
// Both contractions run on v_mfma_f32_16x16x4_f32, one wave per block of 16 input pixels:
//   G[pix][ci] = D[pix][tap] . w4^T[tap][ci]        D gathered from the zero-padded dlogit image in LDS
//   dW4[ci][tap] += a3^T[ci][pix] . D[pix][tap]     a3 = relu(bn3(y3)), staged per wave in LDS
// G goes through a per-wave LDS tile so that y3 is read and dybn3 written as full 256-B rows.
constexpr int DEC_BWD_L = 64 * 16 + 128 + 1;
constexpr int GP = 68;                 // G staging pitch: accumulator-layout stores conflict-free
constexpr int SP = 80;                 // a3 staging pitch: A-fragment (pixel-major) reads conflict-free

// MODE 0: the sums pass -- part[] (dW4, BatchNorm-backward sums, db4); dybn3 is stored only if `dybn` is given.
// MODE 1: the apply pass -- recomputes dybn3 and writes dy3 = scale*(dybn3 - c1 - xhat*c2) straight away, so the
//         masked gradient never makes a round trip through HBM (saves a 1.9 GB write and a 3.9 GB re-read).
template <int MODE>
__global__ __launch_bounds__(256, MODE == 0 ? 2 : 3) void dec_out_bwd_kernel(const float* __restrict__ y3,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ w4,
                                                          const float* __restrict__ p_in,
                                                          const float* __restrict__ target, int M, int nimg,
                                                          const float* __restrict__ g_dev, float gmul,
                                                          const float* __restrict__ c1,
                                                          const float* __restrict__ c2,
                                                          float* __restrict__ dybn, float* __restrict__ part) {
    const float gscale = gmul * (g_dev ? g_dev[0] : 1.f);
    constexpr int PL = HO + 2;                     // zero-padded border: taps never branch
    __shared__ float dl[2][PL * PL];
    __shared__ __align__(16) float gst[4][16 * GP];
    __shared__ __align__(16) float ast[4][16 * SP];
    __shared__ float dbred[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    float bw[16];                                  // w4[ci = 16cb + i][tap = 4s + g]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) bw[4 * cb + s4] = w4[(16 * cb + i) * 16 + 4 * s4 + g];
    // row pass: lane = (row g + 4j, channels 4i .. 4i+3)
    const float4 sc = reinterpret_cast<const float4*>(scale)[i], sh = reinterpret_cast<const float4*>(shift)[i];
    const float4 mu = reinterpret_cast<const float4*>(mean)[i], is = reinterpret_cast<const float4*>(invstd)[i];
    float4 k1 = make_float4(0.f, 0.f, 0.f, 0.f), k2 = k1;
    if (MODE == 1) {
        k1 = reinterpret_cast<const float4*>(c1)[i];
        k2 = reinterpret_cast<const float4*>(c2)[i];
    }
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    const floatx2 scl = {sc.x, sc.y}, sch = {sc.z, sc.w}, shl = {sh.x, sh.y}, shh = {sh.z, sh.w};
    const floatx2 isl = {is.x, is.y}, ish = {is.z, is.w};
    const floatx2 nml = {-mu.x * is.x, -mu.y * is.y}, nmh = {-mu.z * is.z, -mu.w * is.w};
    const floatx2 k1l = {k1.x, k1.y}, k1h = {k1.z, k1.w}, nk2l = {-k2.x, -k2.y}, nk2h = {-k2.z, -k2.w};
    floatx4 dw[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) dw[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx2 s1l = {0.f, 0.f}, s1h = s1l, s2l = s1l, s2h = s1l;
    float dbs = 0.f;
    for (int e = tid; e < 2 * PL * PL; e += 256) (&dl[0][0])[e] = 0.f;
    __syncthreads();
    int buf = 0;
    float* gw = gst[w];
    float* aw = ast[w];
    // operands are fetched one step ahead: p / target of the next image, y3 rows of the wave's next pixel block
    constexpr int NPT = (NPIX_O + 255) / 256;
    float pp[NPT], tt[NPT];
    float4 vv[4];
    auto fetch_pt = [&](int img) {
        const float* tg = target + (long)(img / M) * NPIX_O;
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int o = tid + 256 * k, oo = o < NPIX_O ? o : 0;
            pp[k] = p_in[(long)img * NPIX_O + oo];
            tt[k] = tg[oo];
        }
    };
    auto fetch_rows = [&](int img, int rbk) {
        const float* yi = y3 + (long)img * NPIX_I * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int px = 16 * rbk + g + 4 * j;
            vv[j] = load_once(yi + (px < NPIX_I ? px : NPIX_I - 1) * C + 4 * i);
        }
    };
    if ((int)blockIdx.x < nimg) {
        fetch_pt(blockIdx.x);
        fetch_rows(blockIdx.x, w);
    }
    for (int img = blockIdx.x; img < nimg; img += gridDim.x, buf ^= 1) {
        float* dlb = dl[buf];
        const bool more = img + (int)gridDim.x < nimg;
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int o = tid + 256 * k;
            const float p = pp[k], x = tt[k];
            const float pq = p * (1.f - p);
            // pq / max(pq, 1e-12) is exactly 1 unless the sigmoid saturated (BCELoss backward's clamp)
            const float d = gscale * (p - x) * (pq >= 1e-12f ? 1.f : pq * 1e12f);
            const int oy = o / HO;
            if (o < NPIX_O) {
                dlb[(oy + 1) * PL + (o - oy * HO) + 1] = d;
                if (MODE == 0) dbs += d;
            }
        }
        if (more) fetch_pt(img + gridDim.x);
        __syncthreads();          // the only block barrier per image (dl is double-buffered)
        float* di = dybn + (long)img * NPIX_I * C;
        for (int rbk = w; rbk < RB; rbk += 4) {
            float4 vr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) vr[j] = vv[j];
            if (MODE == 1) {      // apply pass: next rows requested before this block's work (registers allow it)
                if (rbk + 4 < RB) fetch_rows(img, rbk + 4);
                else if (more) fetch_rows(img + gridDim.x, w);
            }
            // ---- G = D . w4^T
            {
                const int px = 16 * rbk + i;
                const bool ok = px < NPIX_I;
                const int iy = px / HI, ix = px - iy * HI;
                const float* dp = dlb + (ok ? (2 * iy) * PL + 2 * ix + g : 0);      // tap (ky = s, kx = g)
                floatx4 G[4];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) G[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    float a = dp[s4 * PL];
                    a = ok ? a : 0.f;
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        G[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[4 * cb + s4], G[cb], 0, 0, 0);
                }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) gw[(4 * g + reg) * GP + 16 * cb + i] = G[cb][reg];
            }
            // ---- rows: mask, dybn3, BatchNorm-backward sums, a3 for the weight gradient
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = g + 4 * j, px = 16 * rbk + rr;
                const bool ok = px < NPIX_I;
                const float4 gv = *reinterpret_cast<const float4*>(&gw[rr * GP + 4 * i]);
                const float4 v = vr[j];
                // channel pairs on v_pk_fma_f32 / v_pk_add_f32 (VALU instructions take their cycles out of the SIMD's
                // MFMA time); xhat = v*invstd - mean*invstd in one fma
                const floatx2 vl = {v.x, v.y}, vh = {v.z, v.w};
                const floatx2 al = __builtin_elementwise_fma(vl, scl, shl), ah = __builtin_elementwise_fma(vh, sch, shh);
                float4 a, o;
                a.x = fmaxf(al.x, 0.f); a.y = fmaxf(al.y, 0.f); a.z = fmaxf(ah.x, 0.f); a.w = fmaxf(ah.y, 0.f);
                if (!ok) a = make_float4(0.f, 0.f, 0.f, 0.f);
                o.x = a.x > 0.f ? gv.x : 0.f; o.y = a.y > 0.f ? gv.y : 0.f;
                o.z = a.z > 0.f ? gv.z : 0.f; o.w = a.w > 0.f ? gv.w : 0.f;
                const floatx2 xl = __builtin_elementwise_fma(vl, isl, nml), xh = __builtin_elementwise_fma(vh, ish, nmh);
                const floatx2 ol = {o.x, o.y}, oh = {o.z, o.w};
                if (MODE == 1) {
                    const floatx2 rl = (__builtin_elementwise_fma(xl, nk2l, ol) - k1l) * scl;
                    const floatx2 rh = (__builtin_elementwise_fma(xh, nk2h, oh) - k1h) * sch;
                    if (ok) store_once(di + px * C + 4 * i, make_float4(rl.x, rl.y, rh.x, rh.y));
                    continue;
                }
                if (ok && dybn) *reinterpret_cast<float4*>(di + px * C + 4 * i) = o;
                s1l += ol; s1h += oh;
                s2l = __builtin_elementwise_fma(ol, xl, s2l); s2h = __builtin_elementwise_fma(oh, xh, s2h);
                *reinterpret_cast<float4*>(&aw[rr * SP + 4 * i]) = a;
            }
            if (MODE == 0) {      // sums pass: requested after the rows are consumed, hidden behind the dW4 MFMAs and
                                  // the third resident workgroup this buys
                if (rbk + 4 < RB) fetch_rows(img, rbk + 4);
                else if (more) fetch_rows(img + gridDim.x, w);
            }
            // ---- dW4 += a3^T . D   (contraction over the 16 pixels: 4 k-steps of 4)
#pragma unroll
            for (int s4 = 0; MODE == 0 && s4 < 4; ++s4) {
                const int px = 16 * rbk + 4 * s4 + g;
                const bool ok = px < NPIX_I;
                const int iy = px / HI, ix = px - iy * HI;
                float b = dlb[ok ? (2 * iy + (i >> 2)) * PL + 2 * ix + (i & 3) : 0];   // tap i = ky*4 + kx
                b = ok ? b : 0.f;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    dw[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[(4 * s4 + g) * SP + 16 * cb + i], b, dw[cb], 0, 0, 0);
            }
        }
    }
    if (MODE == 1) return;
    // ---- block partials (fixed order): dW4, BatchNorm sums, db4
    __syncthreads();
    float* red = &gst[0][0];                       // 4 waves x 1024 floats = 16 KB <= sizeof(gst)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[w * 1024 + (16 * cb + 4 * g + reg) * 16 + i] = dw[cb][reg];
    float* red2 = &ast[0][0];                      // [2][4 waves][4 row groups][64]
    *reinterpret_cast<float4*>(&red2[((0 * 4 + w) * 4 + g) * 64 + 4 * i]) = make_float4(s1l.x, s1l.y, s1h.x, s1h.y);
    *reinterpret_cast<float4*>(&red2[((1 * 4 + w) * 4 + g) * 64 + 4 * i]) = make_float4(s2l.x, s2l.y, s2h.x, s2h.y);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dbs += __shfl_xor(dbs, o);
    if (lane == 0) dbred[w] = dbs;
    __syncthreads();
    float* pb = part + (long)blockIdx.x * DEC_BWD_L;
    for (int e = tid; e < 1024; e += 256) pb[e] = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
    if (tid < 128) {
        const int k = tid >> 6, c = tid & 63;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red2[(k * 16 + r) * 64 + c];
        pb[1024 + k * 64 + c] = s;
    }
    if (tid == 0) pb[1024 + 128] = (dbred[0] + dbred[1]) + (dbred[2] + dbred[3]);
}

// ------------------------------------------------------------------ the sums pass by linearity
// Everything dec_out_bwd_kernel<0> hands on is linear in dlogit, so it follows from TWO contractions over the pixels,
//   R[ci][tap] = sum_pix m * xhat * D[pix][tap],   P[ci][tap] = sum_pix m * D[pix][tap]       (m = relu mask of bn3(y3)):
//   dW4[ci][tap] = gamma R + beta P   (a3 = m (gamma xhat + beta); gamma = scale/invstd, beta = shift + scale*mean)
//   sum dybn3[ci] = sum_tap w4[ci][tap] P[ci][tap],   sum dybn3*xhat[ci] = sum_tap w4[ci][tap] R[ci][tap].
// The gradient tile G = D . w4^T, its trip through LDS and the per-row mask / sum arithmetic of the general kernel
// (about 100 VALU instructions per 16-pixel block, which on gfx950 take their cycles out of the MFMA time) are gone:
// per block 12 VALU instructions per row, two staged operands (m*xhat, m) and 32 MFMAs.
// part[blk] = { R[64][16], P[64][16], db4 };  dec_out_sums_finalize_kernel turns the column sums into the layout of
// dec_out_bwd_kernel<0>'s partials { dW4, sum dybn3, sum dybn3*xhat, db4 }.
constexpr int DEC_LIN_L = 2 * 64 * 16 + 1;

__global__ __launch_bounds__(256, 3) void dec_out_bwd_lin_kernel(const float* __restrict__ y3,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ p_in,
                                                                 const float* __restrict__ target, int M, int nimg,
                                                                 const float* __restrict__ g_dev, float gmul,
                                                                 float* __restrict__ part) {
    const float gscale = gmul * (g_dev ? g_dev[0] : 1.f);
    constexpr int PL = HO + 2;                     // zero-padded border: taps never branch
    __shared__ float dl[2][PL * PL];
    __shared__ __align__(16) float xst[4][16 * SP];      // m * xhat, per wave
    __shared__ __align__(16) float mst[4][16 * SP];      // m
    __shared__ float dbred[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const float4 sc = reinterpret_cast<const float4*>(scale)[i], sh = reinterpret_cast<const float4*>(shift)[i];
    const float4 mu = reinterpret_cast<const float4*>(mean)[i], is = reinterpret_cast<const float4*>(invstd)[i];
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    const floatx2 scl = {sc.x, sc.y}, sch = {sc.z, sc.w}, shl = {sh.x, sh.y}, shh = {sh.z, sh.w};
    const floatx2 isl = {is.x, is.y}, ish = {is.z, is.w};
    const floatx2 nml = {-mu.x * is.x, -mu.y * is.y}, nmh = {-mu.z * is.z, -mu.w * is.w};
    floatx4 dr[4], dp[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) { dr[cb] = floatx4{0.f, 0.f, 0.f, 0.f}; dp[cb] = dr[cb]; }
    float dbs = 0.f;
    for (int e = tid; e < 2 * PL * PL; e += 256) (&dl[0][0])[e] = 0.f;
    __syncthreads();
    int buf = 0;
    float* xw = xst[w];
    float* mw = mst[w];
    constexpr int NPT = (NPIX_O + 255) / 256;
    // nothing below depends on the image: where the thread's dlogit values go, and where the contraction's D operand
    // (pixel 16 blk + 4 s4 + g, tap i) is read from; pixels past the image read the zero border (dl[0])
    int didx[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int o = tid + 256 * k, oy = o / HO;
        didx[k] = (oy + 1) * PL + (o - oy * HO) + 1;
    }
    int bidx[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int px = 16 * (q < 3 ? w + 4 * q : 12) + 4 * s4 + g;
            const int iy = px / HI, ix = px - iy * HI;
            bidx[q][s4] = px < NPIX_I ? (2 * iy + (i >> 2)) * PL + 2 * ix + (i & 3) : 0;      // tap i = ky*4 + kx
        }
    float pp[NPT], tt[NPT];
    float4 vv[4];
    auto fetch_pt = [&](int img) {
        const float* tg = target + (long)(img / M) * NPIX_O;
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int o = tid + 256 * k, oo = o < NPIX_O ? o : 0;
            pp[k] = p_in[(long)img * NPIX_O + oo];
            tt[k] = tg[oo];
        }
    };
    // rows g + 4j of pixel block q (blocks w, w+4, w+8 and the 13th); the 13th block has four pixels: its rows 4 .. 15
    // are copies of them, multiplied by the zero border in the contraction
    auto fetch_rows = [&](int img, int q) {
#if defined(CMVAE_DEC_ABL) && CMVAE_DEC_ABL == 3      // timing only: the activations come from 32 images (L2 hits)
        const float* yi = y3 + (long)(img & 31) * NPIX_I * C;
#else
        const float* yi = y3 + (long)img * NPIX_I * C;
#endif
        if (q < 3) {
            const float* pr = yi + (16 * (w + 4 * q) + g) * C + 4 * i;
#pragma unroll
            for (int j = 0; j < 4; ++j) vv[j] = *reinterpret_cast<const float4*>(pr + 4 * j * C);
        } else {
            vv[0] = *reinterpret_cast<const float4*>(yi + (16 * 12 + g) * C + 4 * i);
            vv[1] = vv[0]; vv[2] = vv[0]; vv[3] = vv[0];
        }
    };
    if ((int)blockIdx.x < nimg) {
        fetch_pt(blockIdx.x);
        fetch_rows(blockIdx.x, 0);
    }
    int it = 0;
    for (int img = blockIdx.x; img < nimg; img += gridDim.x, buf ^= 1, ++it) {
        float* dlb = dl[buf];
        const bool more = img + (int)gridDim.x < nimg;
        // the 13th block goes round the waves: a fixed owner would be the slowest wave of every image
        const bool has13 = w == (it & 3);
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const float p = pp[k], x = tt[k];
            const float pq = p * (1.f - p);
            const float d = gscale * (p - x) * (pq >= 1e-12f ? 1.f : pq * 1e12f);     // as dec_out_bwd_kernel
            if (tid + 256 * k < NPIX_O) {
                dlb[didx[k]] = d;
                dbs += d;
            }
        }
        if (more) fetch_pt(img + gridDim.x);
        __syncthreads();          // the only block barrier per image (dl is double-buffered)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q == 3 && !has13) break;
            // ---- rows: m * xhat and m, staged for the contraction over the pixels
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = g + 4 * j;
                const float4 v = vv[j];
                const floatx2 vl = {v.x, v.y}, vh = {v.z, v.w};
                const floatx2 al = __builtin_elementwise_fma(vl, scl, shl), ah = __builtin_elementwise_fma(vh, sch, shh);
                const floatx2 xl = __builtin_elementwise_fma(vl, isl, nml), xh = __builtin_elementwise_fma(vh, ish, nmh);
                float4 m;
                m.x = al.x > 0.f ? 1.f : 0.f; m.y = al.y > 0.f ? 1.f : 0.f;
                m.z = ah.x > 0.f ? 1.f : 0.f; m.w = ah.y > 0.f ? 1.f : 0.f;
                const floatx2 rl = floatx2{m.x, m.y} * xl, rh = floatx2{m.z, m.w} * xh;
                *reinterpret_cast<float4*>(&xw[rr * SP + 4 * i]) = make_float4(rl.x, rl.y, rh.x, rh.y);
                *reinterpret_cast<float4*>(&mw[rr * SP + 4 * i]) = m;
            }
            if (q < 2) fetch_rows(img, q + 1);
            else if (q == 2 && has13) fetch_rows(img, 3);
            else if (more) fetch_rows(img + gridDim.x, 0);
            // ---- R += (m xhat)^T . D,  P += m^T . D   (contraction over the 16 pixels: 4 k-steps of 4)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float b = dlb[bidx[q][s4]];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    dr[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xw[(4 * s4 + g) * SP + 16 * cb + i], b, dr[cb], 0, 0, 0);
                    dp[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(mw[(4 * s4 + g) * SP + 16 * cb + i], b, dp[cb], 0, 0, 0);
                }
            }
        }
    }
    // ---- block partials (fixed order): R, P, db4
    __syncthreads();
    float* red = &xst[0][0];                       // 4 waves x 1024 floats = 16 KB <= sizeof(xst)
    float* red2 = &mst[0][0];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            red[w * 1024 + (16 * cb + 4 * g + reg) * 16 + i] = dr[cb][reg];
            red2[w * 1024 + (16 * cb + 4 * g + reg) * 16 + i] = dp[cb][reg];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dbs += __shfl_xor(dbs, o);
    if (lane == 0) dbred[w] = dbs;
    __syncthreads();
    float* pb = part + (long)blockIdx.x * DEC_LIN_L;
    for (int e = tid; e < 1024; e += 256) {
        pb[e] = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
        pb[1024 + e] = (red2[e] + red2[1024 + e]) + (red2[2048 + e] + red2[3072 + e]);
    }
    if (tid == 0) pb[2048] = (dbred[0] + dbred[1]) + (dbred[2] + dbred[3]);
}
static_assert(4 * 16 * SP >= 4 * 1024, "the wave staging tiles double as the 4 x 1024 reduction buffers");

// tot = column sums of the partials {R, P, db4} -> out = { dW4[64][16], sum dybn3[64], sum dybn3*xhat[64], db4 }
__global__ __launch_bounds__(1024) void dec_out_sums_finalize_kernel(const float* __restrict__ tot,
                                                                     const float* __restrict__ w4,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     float* __restrict__ out) {
    __shared__ float s1[1024], s2[1024];
    const int t = threadIdx.x, c = t >> 4, tap = t & 15;
    const float R = tot[t], P = tot[1024 + t];
    const float gamma = scale[c] / invstd[c], beta = fmaf(scale[c], mean[c], shift[c]);
    out[t] = fmaf(gamma, R, beta * P);
    const float wv = w4[t];
    s1[t] = wv * P;
    s2[t] = wv * R;
    __syncthreads();
    if (tap == 0) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { a += s1[t + k]; b += s2[t + k]; }
        out[1024 + c] = a;
        out[1088 + c] = b;
    }
    if (t == 0) out[1152] = tot[2048];
}

static int rowpass_grid(long rows) {
    long g = (rows + 15) / 16;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}
// persistent per-image kernels: one round of resident workgroups (each loops over its images)
static int bwd_sums_grid(int nimg) {
    static const int cap = resident_blocks((const void*)dec_out_bwd_kernel<0>, 256, 1024);
    return nimg < cap ? (nimg > 0 ? nimg : 1) : cap;
}

}  // namespace cmvae

using namespace cmvae;

extern "C" int cmvae_colsum_reduce(const float* part, int nblk, int L, float* out, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(part && out && nblk >= 1 && L >= 1, "cmvae_colsum_reduce: bad arguments");
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((L + 63) / 64), dim3(1024), 0, (hipStream_t)stream, part, nblk, L, out);
    return check_launch("colsum_reduce_kernel");
}

extern "C" int cmvae_bn_finalize(const float* sums, double count, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                                 float* mean, float* invstd, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(sums && gamma && beta && scale && shift && mean && invstd && count > 1, "cmvae_bn_finalize: bad arguments");
    CMVAE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "cmvae_bn_finalize: running stats go together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, (float)count, gamma, beta,
                       eps, momentum, running_mean, running_var, scale, shift, mean, invstd);
    return check_launch("bn_finalize_kernel");
}

extern "C" int cmvae_bn_finalize_part(const float* part, int nblk, double count, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var,
                                      int64_t* num_batches_tracked, float* scale, float* shift, float* mean,
                                      float* invstd, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(part && gamma && beta && scale && shift && mean && invstd && nblk >= 1 && count >= 1,
                  "cmvae_bn_finalize_part: bad arguments");
    CMVAE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "cmvae_bn_finalize_part: running stats go together");
    hipLaunchKernelGGL(bn_finalize_part_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, nblk, (float)count,
                       gamma, beta, eps, momentum, running_mean, running_var, (long long*)num_batches_tracked, scale,
                       shift, mean, invstd);
    return check_launch("bn_finalize_part_kernel");
}

extern "C" int cmvae_bn_bwd_finalize_part(const float* part, int nblk, double count, float* c1, float* c2,
                                          float* dgamma, float* dbeta, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(part && c1 && c2 && dgamma && dbeta && nblk >= 1 && count >= 1, "cmvae_bn_bwd_finalize_part: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_part_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, nblk,
                       (float)count, c1, c2, dgamma, dbeta);
    return check_launch("bn_bwd_finalize_part_kernel");
}

extern "C" int cmvae_bn_eval_fold(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift, float* invstd,
                                  void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && invstd,
                  "cmvae_bn_eval_fold: NULL tensor");
    hipLaunchKernelGGL(bn_eval_fold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, eps, scale, shift, invstd);
    return check_launch("bn_eval_fold_kernel");
}

extern "C" int cmvae_rowpass_grid(int64_t rows) {
    cmvae::clear_stale_error(); return rowpass_grid(rows); }

extern "C" int cmvae_bn_relu_bwd_reduce(const float* d_a, const float* y, const float* scale, const float* shift,
                                        const float* mean, const float* invstd, int64_t rows, int masked, float* part,
                                        void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(d_a && y && scale && shift && mean && invstd && part && rows >= 1, "cmvae_bn_relu_bwd_reduce: bad arguments");
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(rowpass_grid(rows)), dim3(256), 0, (hipStream_t)stream, d_a, y,
                       scale, shift, mean, invstd, (long)rows, masked, part);
    return check_launch("bn_relu_bwd_reduce_kernel");
}

extern "C" int cmvae_bn_bwd_finalize(const float* sums_local, const float* sums_global, double count, float* c1,
                                     float* c2, float* dgamma, float* dbeta, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(sums_local && sums_global && c1 && c2 && dgamma && dbeta && count >= 1, "cmvae_bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums_local, sums_global,
                       (float)count, c1, c2, dgamma, dbeta);
    return check_launch("bn_bwd_finalize_kernel");
}

extern "C" int cmvae_bn_relu_bwd_apply(const float* d_a, const float* y, const float* scale, const float* shift,
                                       const float* mean, const float* invstd, const float* c1, const float* c2,
                                       int64_t rows, int masked, float* dy, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(d_a && y && scale && shift && mean && invstd && c1 && c2 && dy && rows >= 1, "cmvae_bn_relu_bwd_apply: bad arguments");
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(rowpass_grid(rows)), dim3(256), 0, (hipStream_t)stream, d_a, y,
                       scale, shift, mean, invstd, c1, c2, (long)rows, masked, dy);
    return check_launch("bn_relu_bwd_apply_kernel");
}

extern "C" int cmvae_dec_out_grid(int nimg) {
    cmvae::clear_stale_error(); return bwd_sums_grid(nimg); }
extern "C" int cmvae_dec_out_bwd_partial_len(void) {
    cmvae::clear_stale_error(); return DEC_BWD_L; }

extern "C" int cmvae_dec_out_fwd(const float* y3, const float* scale, const float* shift, const float* w4,
                                 const float* b4, const float* target, int M, int nimg, float* p_out,
                                 float* loss_part, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(y3 && scale && shift && w4 && b4 && target && p_out && loss_part, "cmvae_dec_out_fwd: NULL tensor");
    CMVAE_REQUIRE(M >= 1 && nimg >= 1 && nimg % M == 0, "cmvae_dec_out_fwd: images must be a multiple of M");
    static const int cap = resident_blocks((const void*)dec_out_fwd_kernel, 256, 1024);
    hipLaunchKernelGGL(dec_out_fwd_kernel, dim3(nimg < cap ? nimg : cap), dim3(256), 0, (hipStream_t)stream, y3, scale,
                       shift, w4, b4, target, M, nimg, p_out, loss_part);
    return check_launch("dec_out_fwd_kernel");
}

extern "C" int cmvae_dec_out_bwd(const float* y3, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, const float* w4, const float* p, const float* target, int M,
                                 int nimg, const float* g_dev, float gmul, float* dybn, float* part, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(y3 && scale && shift && mean && invstd && w4 && p && target && part, "cmvae_dec_out_bwd: NULL tensor");
    CMVAE_REQUIRE(M >= 1 && nimg >= 1 && nimg % M == 0, "cmvae_dec_out_bwd: images must be a multiple of M");
    const float* none = nullptr;
    hipLaunchKernelGGL(dec_out_bwd_kernel<0>, dim3(bwd_sums_grid(nimg)), dim3(256), 0, (hipStream_t)stream, y3, scale, shift,
                       mean, invstd, w4, p, target, M, nimg, g_dev, gmul, none, none, dybn, part);
    return check_launch("dec_out_bwd_kernel<0>");
}

extern "C" int cmvae_dec_out_bwd_sums_partial_len(void) { cmvae::clear_stale_error(); return DEC_LIN_L; }
extern "C" int cmvae_dec_out_bwd_sums_grid(int nimg) {
    cmvae::clear_stale_error();
    static const int cap = resident_blocks((const void*)dec_out_bwd_lin_kernel, 256, 1024);
    return nimg < cap ? (nimg > 0 ? nimg : 1) : cap;
}

extern "C" int cmvae_dec_out_bwd_sums(const float* y3, const float* scale, const float* shift, const float* mean,
                                      const float* invstd, const float* w4, const float* p, const float* target, int M,
                                      int nimg, const float* g_dev, float gmul, float* part, float* tot, float* out,
                                      void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(y3 && scale && shift && mean && invstd && w4 && p && target && part && tot && out,
                  "cmvae_dec_out_bwd_sums: NULL tensor");
    CMVAE_REQUIRE(M >= 1 && nimg >= 1 && nimg % M == 0, "cmvae_dec_out_bwd_sums: images must be a multiple of M");
    hipStream_t st = (hipStream_t)stream;
    const int grid = cmvae_dec_out_bwd_sums_grid(nimg);
    hipLaunchKernelGGL(dec_out_bwd_lin_kernel, dim3(grid), dim3(256), 0, st, y3, scale, shift, mean, invstd, p, target, M,
                       nimg, g_dev, gmul, part);
    if (int e = check_launch("dec_out_bwd_lin_kernel")) return e;
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((DEC_LIN_L + 63) / 64), dim3(1024), 0, st, (const float*)part, grid,
                       DEC_LIN_L, tot);
    if (int e = check_launch("colsum_reduce_kernel")) return e;
    hipLaunchKernelGGL(dec_out_sums_finalize_kernel, dim3(1), dim3(1024), 0, st, (const float*)tot, w4, scale, shift, mean,
                       invstd, out);
    return check_launch("dec_out_sums_finalize_kernel");
}

extern "C" int cmvae_dec_out_bwd_apply(const float* y3, const float* scale, const float* shift, const float* mean,
                                       const float* invstd, const float* w4, const float* p, const float* target,
                                       int M, int nimg, const float* g_dev, float gmul, const float* c1,
                                       const float* c2, float* dy, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(y3 && scale && shift && mean && invstd && w4 && p && target && c1 && c2 && dy,
                  "cmvae_dec_out_bwd_apply: NULL tensor");
    CMVAE_REQUIRE(M >= 1 && nimg >= 1 && nimg % M == 0, "cmvae_dec_out_bwd_apply: images must be a multiple of M");
    float* nopart = nullptr;
    static const int cap = resident_blocks((const void*)dec_out_bwd_kernel<1>, 256, 1024);
    hipLaunchKernelGGL(dec_out_bwd_kernel<1>, dim3(nimg < cap ? nimg : cap), dim3(256), 0, (hipStream_t)stream, y3, scale, shift,
                       mean, invstd, w4, p, target, M, nimg, g_dev, gmul, c1, c2, dy, nopart);
    return check_launch("dec_out_bwd_kernel<1>");
}
