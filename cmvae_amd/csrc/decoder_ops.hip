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
        float4 g = reinterpret_cast<const float4*>(d_a + r * C)[q];
        const float4 v = reinterpret_cast<const float4*>(y + r * C)[q];
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
        reinterpret_cast<float4*>(dy + r * C)[q] = o;
    }
}

// ------------------------------------------------------------------ last decoder layer, forward
// y3 [n,14,14,64] --relu(bn3)--> convT(64->1,k4,s2,p1) + b --sigmoid--> p [n,28,28];
// loss_part[img] = -sum_pix ( x*max(log p,-100) + (1-x)*max(log1p(-p),-100) ),  x = target[img / M]
constexpr int HI = 14, HO = 28, NPIX_I = HI * HI, NPIX_O = HO * HO;
constexpr int AL = 68;

__global__ __launch_bounds__(256) void dec_out_fwd_kernel(const float* __restrict__ y3,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ w4,   // [64][16]
                                                          const float* __restrict__ b4,
                                                          const float* __restrict__ target, int M, int nimg,
                                                          float* __restrict__ p_out, float* __restrict__ loss_part) {
    __shared__ __align__(16) float as[NPIX_I * AL];
    __shared__ __align__(16) float ws[16 * C];     // [tap][ci]
    __shared__ float lred[4];
    const int tid = threadIdx.x;
    for (int e = tid; e < 16 * C; e += 256) ws[e] = w4[(e & 63) * 16 + (e >> 6)];
    const int q = tid & 15;
    const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
    const float bias = b4[0];
    for (int img = blockIdx.x; img < nimg; img += gridDim.x) {
        __syncthreads();
        const float* yi = y3 + (long)img * NPIX_I * C;
        for (int e = tid; e < NPIX_I * 16; e += 256) {
            const int pix = e >> 4;
            float4 v = reinterpret_cast<const float4*>(yi + pix * C)[q];
            v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
            v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
            v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
            v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
            *reinterpret_cast<float4*>(&as[pix * AL + 4 * q]) = v;
        }
        __syncthreads();
        const float* tg = target + (long)(img / M) * NPIX_O;
        float lsum = 0.f;
        if (tid < NPIX_I) {
            const int m = tid / HI, l = tid - m * HI;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int py = ph >> 1, px = ph & 1;
                float acc = bias;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    // py=0: ky in {1,3}, dy in {0,-1};  py=1: ky in {0,2}, dy in {1,0}
                    const int ky = py ? 2 * a : 2 * a + 1;
                    const int iy = m + (py ? 1 - a : -a);
                    if (iy < 0 || iy >= HI) continue;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int kx = px ? 2 * b : 2 * b + 1;
                        const int ix = l + (px ? 1 - b : -b);
                        if (ix < 0 || ix >= HI) continue;
                        const float* ap = &as[(iy * HI + ix) * AL];
                        const float* wp = &ws[(ky * 4 + kx) * C];
#pragma unroll
                        for (int c4 = 0; c4 < 16; ++c4) {
                            const float4 av = *reinterpret_cast<const float4*>(ap + 4 * c4);
                            const float4 wv = *reinterpret_cast<const float4*>(wp + 4 * c4);
                            acc = fmaf(av.x, wv.x, acc);
                            acc = fmaf(av.y, wv.y, acc);
                            acc = fmaf(av.z, wv.z, acc);
                            acc = fmaf(av.w, wv.w, acc);
                        }
                    }
                }
                const int o = (2 * m + py) * HO + 2 * l + px;
                const float p = 1.f / (1.f + expf(-acc));
                p_out[(long)img * NPIX_O + o] = p;
                const float x = tg[o];
                lsum -= x * fmaxf(logf(p), -100.f) + (1.f - x) * fmaxf(log1pf(-p), -100.f);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
        if ((tid & 63) == 0) lred[tid >> 6] = lsum;
        __syncthreads();
        if (tid == 0) loss_part[img] = (lred[0] + lred[1]) + (lred[2] + lred[3]);
    }
}

// ------------------------------------------------------------------ last decoder layer, backward
// dlogit = gscale * (p - x) * p(1-p) / max(p(1-p), 1e-12)      (BCELoss backward through the sigmoid)
// dybn3[pix][ci] = mask * sum_tap dlogit[tap pos] * w4[ci][tap]     (mask: relu(bn3(y3)) > 0)
// part[blk] = { dW4[64][16], sum dybn3[64], sum dybn3*xhat[64], db4 }            (L = 1024 + 128 + 1)
constexpr int DEC_BWD_L = 64 * 16 + 128 + 1;

__global__ __launch_bounds__(256) void dec_out_bwd_kernel(const float* __restrict__ y3,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ w4,
                                                          const float* __restrict__ p_in,
                                                          const float* __restrict__ target, int M, int nimg,
                                                          const float* __restrict__ g_dev, float gmul,
                                                          float* __restrict__ dybn, float* __restrict__ part) {
    const float gscale = gmul * (g_dev ? g_dev[0] : 1.f);
    __shared__ float dl[(HO + 2) * (HO + 2)];     // zero-padded border: taps never branch
    __shared__ float red[4 * (16 + 2) * 64];
    __shared__ float dbred[4];
    const int tid = threadIdx.x, ci = tid & 63, w = tid >> 6;
    constexpr int PL = HO + 2;
    float wreg[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) wreg[t] = w4[ci * 16 + t];
    const float sc = scale[ci], sh = shift[ci], mu = mean[ci], is = invstd[ci];
    float dw[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) dw[t] = 0.f;
    float s1 = 0.f, s2 = 0.f, dbs = 0.f;
    for (int e = tid; e < PL * PL; e += 256) dl[e] = 0.f;
    for (int img = blockIdx.x; img < nimg; img += gridDim.x) {
        __syncthreads();
        const float* tg = target + (long)(img / M) * NPIX_O;
        for (int o = tid; o < NPIX_O; o += 256) {
            const float p = p_in[(long)img * NPIX_O + o];
            const float x = tg[o];
            const float pq = p * (1.f - p);
            const float d = gscale * (p - x) * pq / fmaxf(pq, 1e-12f);
            dl[(o / HO + 1) * PL + (o % HO) + 1] = d;
            dbs += d;
        }
        __syncthreads();
        const float* yi = y3 + (long)img * NPIX_I * C;
        float* di = dybn + (long)img * NPIX_I * C;
        for (int pix = w; pix < NPIX_I; pix += 4) {
            const int iy = pix / HI, ix = pix - iy * HI;
            const float v = yi[pix * C + ci];
            const float a = fmaxf(fmaf(v, sc, sh), 0.f);
            // output positions (2iy-1+ky, 2ix-1+kx) -> padded index (2iy+ky, 2ix+kx)
            const float* dp = &dl[(2 * iy) * PL + 2 * ix];
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const float d = dp[ky * PL + kx];
                    acc = fmaf(d, wreg[ky * 4 + kx], acc);
                    dw[ky * 4 + kx] = fmaf(a, d, dw[ky * 4 + kx]);
                }
            const float g = a > 0.f ? acc : 0.f;
            di[pix * C + ci] = g;
            s1 += g;
            s2 = fmaf(g, (v - mu) * is, s2);
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) red[(w * 18 + t) * 64 + ci] = dw[t];
    red[(w * 18 + 16) * 64 + ci] = s1;
    red[(w * 18 + 17) * 64 + ci] = s2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dbs += __shfl_xor(dbs, o);
    if (ci == 0) dbred[w] = dbs;
    __syncthreads();
    float* pb = part + (long)blockIdx.x * DEC_BWD_L;
    for (int e = tid; e < 18 * 64; e += 256) {
        const int t = e >> 6, c = e & 63;
        const float s = (red[(0 * 18 + t) * 64 + c] + red[(1 * 18 + t) * 64 + c]) +
                        (red[(2 * 18 + t) * 64 + c] + red[(3 * 18 + t) * 64 + c]);
        if (t < 16) pb[c * 16 + t] = s;            // dW4[ci][tap]
        else pb[1024 + (t - 16) * 64 + c] = s;     // BN-backward sums
    }
    if (tid == 0) pb[1024 + 128] = (dbred[0] + dbred[1]) + (dbred[2] + dbred[3]);
}

static int rowpass_grid(long rows) {
    long g = (rows + 15) / 16;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}
static int img_grid(int nimg) { return nimg < 1024 ? (nimg > 0 ? nimg : 1) : 1024; }

}  // namespace cmvae

using namespace cmvae;

extern "C" int cmvae_colsum_reduce(const float* part, int nblk, int L, float* out, void* stream) {
    CMVAE_REQUIRE(part && out && nblk >= 1 && L >= 1, "cmvae_colsum_reduce: bad arguments");
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((L + 63) / 64), dim3(1024), 0, (hipStream_t)stream, part, nblk, L, out);
    return check_launch("colsum_reduce_kernel");
}

extern "C" int cmvae_bn_finalize(const float* sums, double count, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                                 float* mean, float* invstd, void* stream) {
    CMVAE_REQUIRE(sums && gamma && beta && scale && shift && mean && invstd && count > 1, "cmvae_bn_finalize: bad arguments");
    CMVAE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "cmvae_bn_finalize: running stats go together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, (float)count, gamma, beta,
                       eps, momentum, running_mean, running_var, scale, shift, mean, invstd);
    return check_launch("bn_finalize_kernel");
}

extern "C" int cmvae_rowpass_grid(int64_t rows) { return rowpass_grid(rows); }

extern "C" int cmvae_bn_relu_bwd_reduce(const float* d_a, const float* y, const float* scale, const float* shift,
                                        const float* mean, const float* invstd, int64_t rows, int masked, float* part,
                                        void* stream) {
    CMVAE_REQUIRE(d_a && y && scale && shift && mean && invstd && part && rows >= 1, "cmvae_bn_relu_bwd_reduce: bad arguments");
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(rowpass_grid(rows)), dim3(256), 0, (hipStream_t)stream, d_a, y,
                       scale, shift, mean, invstd, (long)rows, masked, part);
    return check_launch("bn_relu_bwd_reduce_kernel");
}

extern "C" int cmvae_bn_bwd_finalize(const float* sums_local, const float* sums_global, double count, float* c1,
                                     float* c2, float* dgamma, float* dbeta, void* stream) {
    CMVAE_REQUIRE(sums_local && sums_global && c1 && c2 && dgamma && dbeta && count >= 1, "cmvae_bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums_local, sums_global,
                       (float)count, c1, c2, dgamma, dbeta);
    return check_launch("bn_bwd_finalize_kernel");
}

extern "C" int cmvae_bn_relu_bwd_apply(const float* d_a, const float* y, const float* scale, const float* shift,
                                       const float* mean, const float* invstd, const float* c1, const float* c2,
                                       int64_t rows, int masked, float* dy, void* stream) {
    CMVAE_REQUIRE(d_a && y && scale && shift && mean && invstd && c1 && c2 && dy && rows >= 1, "cmvae_bn_relu_bwd_apply: bad arguments");
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(rowpass_grid(rows)), dim3(256), 0, (hipStream_t)stream, d_a, y,
                       scale, shift, mean, invstd, c1, c2, (long)rows, masked, dy);
    return check_launch("bn_relu_bwd_apply_kernel");
}

extern "C" int cmvae_dec_out_grid(int nimg) { return img_grid(nimg); }
extern "C" int cmvae_dec_out_bwd_partial_len(void) { return DEC_BWD_L; }

extern "C" int cmvae_dec_out_fwd(const float* y3, const float* scale, const float* shift, const float* w4,
                                 const float* b4, const float* target, int M, int nimg, float* p_out,
                                 float* loss_part, void* stream) {
    CMVAE_REQUIRE(y3 && scale && shift && w4 && b4 && target && p_out && loss_part, "cmvae_dec_out_fwd: NULL tensor");
    CMVAE_REQUIRE(M >= 1 && nimg >= 1 && nimg % M == 0, "cmvae_dec_out_fwd: images must be a multiple of M");
    hipLaunchKernelGGL(dec_out_fwd_kernel, dim3(nimg < 4096 ? nimg : 4096), dim3(256), 0, (hipStream_t)stream, y3, scale,
                       shift, w4, b4, target, M, nimg, p_out, loss_part);
    return check_launch("dec_out_fwd_kernel");
}

extern "C" int cmvae_dec_out_bwd(const float* y3, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, const float* w4, const float* p, const float* target, int M,
                                 int nimg, const float* g_dev, float gmul, float* dybn, float* part, void* stream) {
    CMVAE_REQUIRE(y3 && scale && shift && mean && invstd && w4 && p && target && dybn && part, "cmvae_dec_out_bwd: NULL tensor");
    CMVAE_REQUIRE(M >= 1 && nimg >= 1 && nimg % M == 0, "cmvae_dec_out_bwd: images must be a multiple of M");
    hipLaunchKernelGGL(dec_out_bwd_kernel, dim3(img_grid(nimg)), dim3(256), 0, (hipStream_t)stream, y3, scale, shift,
                       mean, invstd, w4, p, target, M, nimg, g_dev, gmul, dybn, part);
    return check_launch("dec_out_bwd_kernel");
}
