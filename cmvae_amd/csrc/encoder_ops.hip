// OmniglotEncoder pieces that are not 64->64 GEMMs (omniglot/module.py:12-33), channels-last:
//   * Conv2d(1,64,3,1,1) on 28x28 (K = 9: VALU, HBM-bound on the 64-channel output) + its weight grad
//   * MaxPool2d(2, ceil_mode=True) fused with the preceding BatchNorm + ReLU, forward and backward
//     (the backward also emits the BatchNorm-backward sums).
#include "common.h"

namespace cmvae {

constexpr int EC = 64;

// ------------------------------------------------------------------ conv 1->64, 3x3, pad 1
// x [n,H,W]; w [64][9]; y [n*H*W, 64] (nullptr: statistics only, y is recomputed by its consumers);
// part[blk][128] = {sum y, sum y^2} (optional).
// One image per workgroup pass: the image sits zero-padded in LDS (taps never branch, 16 lanes broadcast-read one
// pixel), a thread owns 4 channels and writes one float4 of a 256-B output row -> HBM-bound on the 64-channel output.
constexpr int EC1_MAX = 34 * 34;       // padded image, H, W <= 32
__global__ __launch_bounds__(256) void enc_conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, int H, int W, int nimg,
                                                            float* __restrict__ y, float* __restrict__ part) {
    __shared__ float red[16 * 128];
    __shared__ float xs[EC1_MAX];
    const int q = threadIdx.x & 15, rb = threadIdx.x >> 4;
    float wr[4][9], br[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        br[c] = b[4 * q + c];
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(4 * q + c) * 9 + t];
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int hw = H * W, PW = W + 2;
    for (int e = threadIdx.x; e < (H + 2) * PW; e += 256) xs[e] = 0.f;
    for (int img = blockIdx.x; img < nimg; img += gridDim.x) {
        __syncthreads();
        const float* xi = x + (long)img * hw;
        for (int e = threadIdx.x; e < hw; e += 256) {
            const int py = e / W;
            xs[(py + 1) * PW + (e - py * W) + 1] = xi[e];
        }
        __syncthreads();
        float* yi = y + (long)img * hw * EC;
        int py = rb / W, px = rb - py * W;
        for (int pix = rb; pix < hw; pix += 16) {
            const float* xp = &xs[py * PW + px];
            float acc[4] = {br[0], br[1], br[2], br[3]};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float xv = xp[ky * PW + kx];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaf(xv, wr[c][ky * 3 + kx], acc[c]);
                }
            if (y) reinterpret_cast<float4*>(yi + pix * EC)[q] = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
            for (int c = 0; c < 4; ++c) { s1[c] += acc[c]; s2[c] = fmaf(acc[c], acc[c], s2[c]); }
            px += 16;
            while (px >= W) { px -= W; ++py; }
        }
    }
    if (part) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { red[rb * 128 + 4 * q + c] = s1[c]; red[rb * 128 + 64 + 4 * q + c] = s2[c]; }
        __syncthreads();
        if (threadIdx.x < 128) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r * 128 + threadIdx.x];
            part[(long)blockIdx.x * 128 + threadIdx.x] = s;
        }
    }
}

// Evaluation mode (BatchNorm from the running statistics, nothing kept for a backward pass): conv 1->64, the
// BatchNorm fold, ReLU and MaxPool2d(2, ceil_mode) in one pass.  The 64-channel pre-pooling activation (200 KB per
// 28x28 image, the largest tensor of GMVAE.prediction) is never written: HBM traffic is the image plus the pooled
// output.  x [n,H,W]; out [n*Ho*Wo, 64] = max over the (clipped) 2x2 window of relu(scale*conv+shift).
__global__ __launch_bounds__(256) void enc_conv1_pool_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int H, int W, int nimg,
                                                             float* __restrict__ out) {
    __shared__ float xs[36 * 36];          // zero-padded image, one extra row/column for the odd-size window
    const int q = threadIdx.x & 15, rb = threadIdx.x >> 4;
    float wr[4][9], br[4], sc[4], sh[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        br[c] = b[4 * q + c]; sc[c] = scale[4 * q + c]; sh[c] = shift[4 * q + c];
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(4 * q + c) * 9 + t];
    }
    const int hw = H * W, PW = W + 4, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    for (int e = threadIdx.x; e < (H + 4) * PW; e += 256) xs[e] = 0.f;
    for (int img = blockIdx.x; img < nimg; img += gridDim.x) {
        __syncthreads();
        const float* xi = x + (long)img * hw;
        for (int e = threadIdx.x; e < hw; e += 256) {
            const int py = e / W;
            xs[(py + 1) * PW + (e - py * W) + 1] = xi[e];
        }
        __syncthreads();
        float* oi = out + (long)img * Ho * Wo * EC;
        for (int op = rb; op < Ho * Wo; op += 16) {
            const int oy = op / Wo, ox = op - oy * Wo;
            const float* xp = &xs[(2 * oy) * PW + 2 * ox];       // 4x4 input patch of the 2x2 window
            float patch[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) patch[i][j] = xp[i * PW + j];
            float best[4] = {0.f, 0.f, 0.f, 0.f};                 // ReLU floor
#pragma unroll
            for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                for (int wx = 0; wx < 2; ++wx) {
                    const bool inside = (2 * oy + wy < H) && (2 * ox + wx < W);
                    float acc[4] = {br[0], br[1], br[2], br[3]};
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                acc[c] = fmaf(patch[wy + ky][wx + kx], wr[c][ky * 3 + kx], acc[c]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = fmaf(acc[c], sc[c], sh[c]);
                        best[c] = inside ? fmaxf(best[c], v) : best[c];
                    }
                }
            reinterpret_cast<float4*>(oi + op * EC)[q] = make_float4(best[0], best[1], best[2], best[3]);
        }
    }
}

// ------------------------------------------------------------------ relu(bn(y)) -> maxpool 2x2 (ceil)
__device__ __forceinline__ float4 fold4(const float4 v, const float4 sc, const float4 sh) {
    return make_float4(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), fmaxf(fmaf(v.y, sc.y, sh.y), 0.f),
                       fmaxf(fmaf(v.z, sc.z, sh.z), 0.f), fmaxf(fmaf(v.w, sc.w, sh.w), 0.f));
}

__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int H, int W, int Ho, int Wo,
                                                       long orows, float* __restrict__ out) {
    const int q = threadIdx.x & 15, rb = threadIdx.x >> 4;
    const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
    const int ohw = Ho * Wo;
    for (long r = (long)blockIdx.x * 16 + rb; r < orows; r += (long)gridDim.x * 16) {
        const long n = r / ohw;
        const int rem = (int)(r - n * ohw);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int sy = 2 * oy + dy;
            if (sy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int sx = 2 * ox + dx;
                if (sx >= W) continue;
                const float4 v = fold4(reinterpret_cast<const float4*>(y + ((n * H + sy) * W + sx) * EC)[q], sc, sh);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        reinterpret_cast<float4*>(out + r * EC)[q] = m;
    }
}

// d_pool [n,Ho,Wo,64] -> dybn [n,H,W,64] (gradient w.r.t. the BatchNorm output, ReLU mask applied; the
// first maximum of a window receives it, as PyTorch's max_pool2d does) and
// part[blk][128] = {sum dybn, sum dybn*xhat}
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ d_pool, const float* __restrict__ y,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, int H, int W, int Ho, int Wo,
                                                       long orows, float* __restrict__ dybn,
                                                       float* __restrict__ part) {
    __shared__ float red[16 * 128];
    const int q = threadIdx.x & 15, rb = threadIdx.x >> 4;
    const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
    const float4 mu4 = reinterpret_cast<const float4*>(mean)[q], is4 = reinterpret_cast<const float4*>(invstd)[q];
    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int ohw = Ho * Wo;
    for (long r = (long)blockIdx.x * 16 + rb; r < orows; r += (long)gridDim.x * 16) {
        const long n = r / ohw;
        const int rem = (int)(r - n * ohw);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const float4 g4 = reinterpret_cast<const float4*>(d_pool + r * EC)[q];
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        float raw[4][4], val[4][4];      // [window slot][channel]
        bool in[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sy = 2 * oy + (s >> 1), sx = 2 * ox + (s & 1);
            in[s] = sy < H && sx < W;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in[s]) v = reinterpret_cast<const float4*>(y + ((n * H + sy) * W + sx) * EC)[q];
            const float4 f = fold4(v, sc, sh);
            raw[s][0] = v.x; raw[s][1] = v.y; raw[s][2] = v.z; raw[s][3] = v.w;
            val[s][0] = f.x; val[s][1] = f.y; val[s][2] = f.z; val[s][3] = f.w;
        }
        float o[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int best = 0;
            float bv = -INFINITY;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (in[s] && val[s][c] > bv) { bv = val[s][c]; best = s; }
            const float gg = bv > 0.f ? g[c] : 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) o[s][c] = (s == best) ? gg : 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s == best) { s1[c] += gg; s2[c] = fmaf(gg, (raw[s][c] - mu[c]) * is[c], s2[c]); }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!in[s]) continue;
            const int sy = 2 * oy + (s >> 1), sx = 2 * ox + (s & 1);
            reinterpret_cast<float4*>(dybn + ((n * H + sy) * W + sx) * EC)[q] =
                make_float4(o[s][0], o[s][1], o[s][2], o[s][3]);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { red[rb * 128 + 4 * q + c] = s1[c]; red[rb * 128 + 64 + 4 * q + c] = s2[c]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r * 128 + threadIdx.x];
        part[(long)blockIdx.x * 128 + threadIdx.x] = s;
    }
}

// ------------------------------------------------------------------ layer 1 backward in ONE pass over y
// conv1's gradient dy = scale * (dybn - c1 - c2 * xhat) feeds only its weight gradient (there is no data gradient below
// the first layer), and that is LINEAR in dybn and xhat:
//   dW[c][t] = scale_c * ( sum dybn*x_t  -  c1_c * sum x_t  -  c2_c * sum xhat*x_t ).
// So the pooling backward never writes dybn: in the pass that finds the window maxima it accumulates, per channel,
//   s1 = sum dybn, s2 = sum dybn*xhat, P1[t] = sum dybn*(x_t - m_t)
// and enc_conv1_lin_finalize_kernel puts c1, c2 in afterwards.  The taps are CENTRED with their means m_t (known before
// the pass, from the images alone): with raw x_t the first two terms would be a covariance taken as a difference of
// products of means.  m_t = (1/N) sum_p x_t(p), zero padding included.  The third sum needs no pass over y at all:
// y_c - mean_c = sum_t' w[c][t'] (x_t' - m_t') + delta_c with a constant delta_c, so
//   sum xhat*(x_t - m_t) = invstd_c * sum_t' w[c][t'] * Gc[t'][t],   Gc = the 9x9 Gram matrix of the centred taps,
//   sum xhat             = invstd_c * N * delta_c,                   delta_c = b_c + sum_t' w[c][t'] m_t' - mean_c
// (delta is zero when the batch mean is this process's own, not with several ranks).
constexpr int TAPM_BLOCKS = 256, TAPM_COLS = 64, TAPM_USED = 9 + 45;   // partials: 9 tap sums, 45 products t' <= t
__host__ __device__ constexpr int gram_idx(int a, int b) {               // a <= b
    return 9 + a * 9 - a * (a - 1) / 2 + (b - a);
}
__global__ __launch_bounds__(1024) void enc_conv1_tap_sums_kernel(const float* __restrict__ x, int H, int W, int npix,
                                                                  float* __restrict__ part_m) {
    __shared__ float red[16 * TAPM_USED];
    float acc[TAPM_USED];
#pragma unroll
    for (int k = 0; k < TAPM_USED; ++k) acc[k] = 0.f;
    const int hw = H * W;
    // four pixels per trip: their 36 loads are in flight together (the kernel is a chain of load latencies otherwise)
    for (int e0 = blockIdx.x * 1024 + threadIdx.x; e0 < npix; e0 += 4 * TAPM_BLOCKS * 1024) {
        float xt[4][9];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * TAPM_BLOCKS * 1024;
            const bool live = e < npix;
            const int ee = live ? e : 0;
            const int n = ee / hw, rem = ee - n * hw;
            const int py = rem / W, px = rem - py * W;
            const float* xi = x + (long)n * hw;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int sy = py + ky - 1, sx = px + kx - 1;
                    xt[u][ky * 3 + kx] = (live && sy >= 0 && sy < H && sx >= 0 && sx < W) ? xi[sy * W + sx] : 0.f;
                }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 9; ++a) {
                acc[a] += xt[u][a];
#pragma unroll
                for (int b = a; b < 9; ++b) acc[gram_idx(a, b)] = fmaf(xt[u][a], xt[u][b], acc[gram_idx(a, b)]);
            }
    }
    // wave reduction of all 54 sums in 63 shuffles instead of 54 * 6: each step halves the values a lane carries
    // (lanes with the offset bit set keep the upper half); lane l ends up with the wave's total of sum number l
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float v[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) v[k] = k < TAPM_USED ? acc[k] : 0.f;
#pragma unroll
    for (int half = 32; half >= 1; half >>= 1) {
        const bool up = (lane & half) != 0;
#pragma unroll
        for (int k = 0; k < half; ++k) {
            const float recv = __shfl_xor(up ? v[k] : v[k + half], half);
            v[k] = (up ? v[k + half] : v[k]) + recv;
        }
    }
    if (lane < TAPM_USED) red[w * TAPM_USED + lane] = v[0];
    __syncthreads();
    if (threadIdx.x < TAPM_USED) {
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < 16; ++ww) v += red[ww * TAPM_USED + threadIdx.x];
        part_m[blockIdx.x * TAPM_COLS + threadIdx.x] = v;
    }
}

// column sums of the first ncol columns of part_m into out[] (LDS), in a fixed order: the pass and the finalize get
// the same bits.  The whole block (>= 256 threads) calls it; buf: 256 * ncol floats of LDS.
template <int ncol>
__device__ __forceinline__ void tap_stats_block(const float* __restrict__ part_m, float* buf, float* out) {
    const int tid = threadIdx.x;
    if (tid < TAPM_BLOCKS) {
        float v[ncol];
#pragma unroll
        for (int k = 0; k < ncol; ++k) v[k] = part_m[tid * TAPM_COLS + k];
#pragma unroll
        for (int k = 0; k < ncol; ++k) buf[k * TAPM_BLOCKS + tid] = v[k];
    }
    __syncthreads();
    if (tid < ncol) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int b = 0; b < TAPM_BLOCKS; b += 4) {
            s0 += buf[tid * TAPM_BLOCKS + b]; s1 += buf[tid * TAPM_BLOCKS + b + 1];
            s2 += buf[tid * TAPM_BLOCKS + b + 2]; s3 += buf[tid * TAPM_BLOCKS + b + 3];
        }
        out[tid] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
}

constexpr int LIN1_COLS = 2 * EC + 9 * EC;           // s1, s2 [64] each, P1 [64][9]
// The pass, with y recomputed from the image (conv1 is 9 multiply-adds per value; the arithmetic of
// enc_conv1_fwd_kernel / enc_conv1_pool_kernel, bit for bit, so the window maxima are the forward's): the 64-channel
// pre-pooling activation of layer 1 is never stored.  One image per workgroup pass, zero-padded in LDS.
__global__ __launch_bounds__(256) void enc_conv1_bwd_lin_rc_kernel(
    const float* __restrict__ d_pool, const float* __restrict__ x, const float* __restrict__ w,
    const float* __restrict__ b, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ part_m, float npix_loc,
    int H, int W, int nimg, float* __restrict__ part) {
    __shared__ float red[4 * LIN1_COLS];
    __shared__ float ms[9];
    __shared__ float xs[36 * 36];
    const int q = threadIdx.x & 15, rb = threadIdx.x >> 4, lane = threadIdx.x & 63;
    tap_stats_block<9>(part_m, red, ms);
    float m[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) m[t] = ms[t] / npix_loc;
    float wr[4][9], br[4], sc[4], sh[4], mu[4], is[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        br[c] = b[4 * q + c]; sc[c] = scale[4 * q + c]; sh[c] = shift[4 * q + c];
        mu[c] = mean[4 * q + c]; is[c] = invstd[4 * q + c];
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(4 * q + c) * 9 + t];
    }
    float s1[4], s2[4], P1[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        s1[c] = s2[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) P1[c][t] = 0.f;
    }
    const int hw = H * W, PW = W + 4, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    for (int e = threadIdx.x; e < (H + 4) * PW; e += 256) xs[e] = 0.f;
    for (int img = blockIdx.x; img < nimg; img += gridDim.x) {
        __syncthreads();
        const float* xi = x + (long)img * hw;
        for (int e = threadIdx.x; e < hw; e += 256) {
            const int py = e / W;
            xs[(py + 1) * PW + (e - py * W) + 1] = xi[e];
        }
        __syncthreads();
        const float* gi = d_pool + (long)img * Ho * Wo * EC;
        for (int op = rb; op < Ho * Wo; op += 16) {
            const int oy = op / Wo, ox = op - oy * Wo;
            const float4 g4 = reinterpret_cast<const float4*>(gi + op * EC)[q];
            const float g[4] = {g4.x, g4.y, g4.z, g4.w};
            const float* xp0 = &xs[(2 * oy) * PW + 2 * ox];      // 4x4 input patch of the 2x2 window
            float xp[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) xp[i][j] = xp0[i * PW + j];
            float raw[4][4], val[4][4];    // [window slot][channel]
            bool in[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                in[s] = (2 * oy + (s >> 1) < H) && (2 * ox + (s & 1) < W);
                float acc[4] = {br[0], br[1], br[2], br[3]};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[c] = fmaf(xp[(s >> 1) + ky][(s & 1) + kx], wr[c][ky * 3 + kx], acc[c]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    raw[s][c] = acc[c];
                    val[s][c] = fmaxf(fmaf(acc[c], sc[c], sh[c]), 0.f);
                }
            }
            float o[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int best = 0;
                float bv = -INFINITY;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (in[s] && val[s][c] > bv) { bv = val[s][c]; best = s; }
                const float gg = bv > 0.f ? g[c] : 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    o[s][c] = (s == best) ? gg : 0.f;
                    s2[c] = fmaf(o[s][c], (raw[s][c] - mu[c]) * is[c], s2[c]);
                }
                s1[c] += gg;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (!in[s]) continue;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float xc = xp[(s >> 1) + ky][(s & 1) + kx] - m[ky * 3 + kx];
#pragma unroll
                        for (int c = 0; c < 4; ++c) P1[c][ky * 3 + kx] = fmaf(o[s][c], xc, P1[c][ky * 3 + kx]);
                    }
            }
        }
    }
    __syncthreads();                                 // red: the scratch of tap_stats_block is free again
    const int wv = threadIdx.x >> 6;
    auto put = [&](int col, float v) {
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) red[wv * LIN1_COLS + col] = v;
    };
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        put(4 * q + c, s1[c]);
        put(EC + 4 * q + c, s2[c]);
#pragma unroll
        for (int t = 0; t < 9; ++t) put(2 * EC + (4 * q + c) * 9 + t, P1[c][t]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < LIN1_COLS; e += 256)
        part[(long)blockIdx.x * LIN1_COLS + e] =
            (red[e] + red[LIN1_COLS + e]) + (red[2 * LIN1_COLS + e] + red[3 * LIN1_COLS + e]);
}

// tot [LIN1_COLS]: this rank's column sums of the partials; sums_glob [128]: {s1, s2} over all ranks (tot itself in a
// single process); count_glob / npix_loc: values per channel over all ranks / on this rank; w [64][9], b [64]: conv1.
__global__ __launch_bounds__(640) void enc_conv1_lin_finalize_kernel(
    const float* __restrict__ tot, const float* __restrict__ sums_glob, float count_glob, float npix_loc,
    const float* __restrict__ part_m, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float buf[TAPM_BLOCKS * TAPM_USED];
    __shared__ float st[TAPM_USED];
    tap_stats_block<TAPM_USED>(part_m, buf, st);
    const int e = threadIdx.x;
    float m[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) m[t] = st[t] / npix_loc;
    const int c = e < 9 * EC ? e / 9 : e - 9 * EC;
    const float c1 = sums_glob[c] / count_glob, c2 = sums_glob[EC + c] / count_glob;
    const float s1 = tot[c], s2 = tot[EC + c];
    float wm = b[c];
#pragma unroll
    for (int t = 0; t < 9; ++t) wm = fmaf(w[c * 9 + t], m[t], wm);
    const float s3 = invstd[c] * npix_loc * (wm - mean[c]);     // sum xhat over this rank's pixels
    const float r1 = s1 - c1 * npix_loc;            // zero in a single process (c1 is then this rank's mean of dybn)
    if (e < 9 * EC) {
        const int t = e - 9 * c;
        float p3 = 0.f;                             // sum xhat * (x_t - m_t)
#pragma unroll
        for (int a = 0; a < 9; ++a) {
            const float gc = st[a <= t ? gram_idx(a, t) : gram_idx(t, a)] - npix_loc * m[a] * m[t];
            p3 = fmaf(w[c * 9 + a], gc, p3);
        }
        p3 *= invstd[c];
        const float p1 = tot[2 * EC + e];
        dW[e] = scale[c] * ((p1 + m[t] * r1) - c2 * (p3 + m[t] * s3));
    } else {
        db[c] = scale[c] * (r1 - c2 * s3);
        dgamma[c] = s2;
        dbeta[c] = s1;
    }
}

static int rp_grid(long rows) {
    long g = (rows + 15) / 16;
    if (g > 1024) g = 1024;
    return g < 1 ? 1 : (int)g;
}

}  // namespace cmvae

using namespace cmvae;

extern "C" int cmvae_enc_conv1_fwd(const float* x, const float* w, const float* b, int n, int H, int W, float* y,
                                   float* stats_part, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(x && w && b && (y || stats_part) && n >= 1 && H >= 1 && W >= 1 && H <= 32 && W <= 32,
                  "cmvae_enc_conv1_fwd: bad arguments (images up to 32x32)");
    const long rows = (long)n * H * W;
    hipLaunchKernelGGL(enc_conv1_fwd_kernel, dim3(rp_grid(rows)), dim3(256), 0, (hipStream_t)stream, x, w, b, H, W, n,
                       y, stats_part);
    return check_launch("enc_conv1_fwd_kernel");
}

extern "C" int cmvae_enc_conv1_pool_fwd(const float* x, const float* w, const float* b, const float* scale,
                                        const float* shift, int n, int H, int W, float* out, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(x && w && b && scale && shift && out && n >= 1, "cmvae_enc_conv1_pool_fwd: bad arguments");
    CMVAE_REQUIRE(H >= 1 && W >= 1 && H <= 32 && W <= 32, "cmvae_enc_conv1_pool_fwd: image %dx%d exceeds 32x32", H, W);
    const int grid = n < 2048 ? n : 2048;
    hipLaunchKernelGGL(enc_conv1_pool_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, b, scale, shift, H, W,
                       n, out);
    return check_launch("enc_conv1_pool_kernel");
}

extern "C" int cmvae_enc_conv1_bwd_lin_cols(void) { cmvae::clear_stale_error(); return LIN1_COLS; }

extern "C" int cmvae_enc_conv1_tap_sums(const float* x, int n, int H, int W, float* part_m, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(x && part_m && n >= 1 && H >= 1 && W >= 1 && (long)n * H * W < (1L << 30),
                  "cmvae_enc_conv1_tap_sums: bad arguments");
    hipLaunchKernelGGL(enc_conv1_tap_sums_kernel, dim3(TAPM_BLOCKS), dim3(1024), 0, (hipStream_t)stream, x, H, W,
                       n * H * W, part_m);
    return check_launch("enc_conv1_tap_sums_kernel");
}

extern "C" int cmvae_enc_conv1_bwd_lin_rc_grid(int n) { cmvae::clear_stale_error(); return n < 2048 ? (n < 1 ? 1 : n) : 2048; }

extern "C" int cmvae_enc_conv1_bwd_lin_rc(const float* d_pool, const float* x, const float* w, const float* b,
                                         const float* scale, const float* shift, const float* mean, const float* invstd,
                                         const float* part_m, int n, int H, int W, float* part, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(d_pool && x && w && b && scale && shift && mean && invstd && part_m && part && n >= 1,
                  "cmvae_enc_conv1_bwd_lin_rc: bad arguments");
    CMVAE_REQUIRE(H >= 1 && W >= 1 && H <= 32 && W <= 32, "cmvae_enc_conv1_bwd_lin_rc: image %dx%d exceeds 32x32", H, W);
    const int grid = n < 2048 ? n : 2048;
    hipLaunchKernelGGL(enc_conv1_bwd_lin_rc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_pool, x, w, b, scale,
                       shift, mean, invstd, part_m, (float)((long)n * H * W), H, W, n, part);
    return check_launch("enc_conv1_bwd_lin_rc_kernel");
}

extern "C" int cmvae_enc_conv1_lin_finalize(const float* tot, const float* sums_glob, float count_glob, int n, int H,
                                           int W, const float* part_m, const float* w, const float* b,
                                           const float* scale, const float* mean, const float* invstd, float* dW,
                                           float* db, float* dgamma, float* dbeta, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(tot && sums_glob && part_m && w && b && scale && mean && invstd && dW && db && dgamma && dbeta &&
                      n >= 1 && count_glob >= 1.f,
                  "cmvae_enc_conv1_lin_finalize: bad arguments");
    hipLaunchKernelGGL(enc_conv1_lin_finalize_kernel, dim3(1), dim3(640), 0, (hipStream_t)stream, tot, sums_glob,
                       count_glob, (float)((long)n * H * W), part_m, w, b, scale, mean, invstd, dW, db, dgamma, dbeta);
    return check_launch("enc_conv1_lin_finalize_kernel");
}

extern "C" int cmvae_pool_fwd(const float* y, const float* scale, const float* shift, int n, int H, int W, float* out,
                              void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(y && scale && shift && out && n >= 1 && H >= 1 && W >= 1, "cmvae_pool_fwd: bad arguments");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long orows = (long)n * Ho * Wo;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(rp_grid(orows)), dim3(256), 0, (hipStream_t)stream, y, scale, shift, H, W,
                       Ho, Wo, orows, out);
    return check_launch("pool_fwd_kernel");
}

extern "C" int cmvae_pool_bwd(const float* d_pool, const float* y, const float* scale, const float* shift,
                              const float* mean, const float* invstd, int n, int H, int W, float* dybn, float* part,
                              void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(d_pool && y && scale && shift && mean && invstd && dybn && part && n >= 1, "cmvae_pool_bwd: bad arguments");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long orows = (long)n * Ho * Wo;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(rp_grid(orows)), dim3(256), 0, (hipStream_t)stream, d_pool, y, scale, shift,
                       mean, invstd, H, W, Ho, Wo, orows, dybn, part);
    return check_launch("pool_bwd_kernel");
}
