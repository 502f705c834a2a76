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

// The same for few rows with many samples each (the feature models: 80 rows x 256 samples — one wave per row walked
// its 256 samples in a dependent chain for 112 us): NW waves share a row, wave w takes samples w, w + NW, ..., and the
// partial sums are added in wave order (fixed order: reproducible).
template <int NW>
__global__ __launch_bounds__(64 * NW) void reparam_logq_bwd_wide_kernel(const float* __restrict__ logvar,
                                                                        const float* __restrict__ eps,
                                                                        const float* __restrict__ gz,
                                                                        const float* __restrict__ glogq, int M,
                                                                        float* __restrict__ dmean,
                                                                        float* __restrict__ dlogvar) {
    __shared__ float red[3][NW][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t r = blockIdx.x;
    const float hs = 0.5f * expf(0.5f * logvar[r * D + lane]);
    float am = 0.f, av = 0.f, ag = 0.f;
    for (int m = w; m < M; m += NW) {
        const size_t o = (r * M + m) * D + lane;
        const float g = gz ? gz[o] : 0.f;
        am += g;
        av = fmaf(g * eps[o], hs, av);
        if (glogq) ag += glogq[r * M + m];
    }
    red[0][w][lane] = am; red[1][w][lane] = av; red[2][w][lane] = ag;
    __syncthreads();
    if (w == 0) {
        am = av = ag = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) { am += red[0][k][lane]; av += red[1][k][lane]; ag += red[2][k][lane]; }
        dmean[r * D + lane] = am;
        dlogvar[r * D + lane] = av - 0.5f * ag;
    }
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

// ------------------------------------------------------------------ centring of an episode's latents
// out[b][n][:] = x[b][n][:] - mean_n x[b][n][:]   (omniglot/model.py:449: the bulk causal layer sees q_z minus its
// per-episode mean).  Two launches: per-chunk column sums, then the subtraction (every workgroup re-reduces the
// <= 64 chunk sums of its episode from L2 in a fixed order).  The map is a symmetric projection, so the backward
// pass is the same call on the gradient.
constexpr int CENTER_ROWS = 256;          // rows per chunk
__global__ __launch_bounds__(256) void center_partial_kernel(const float* __restrict__ x, int N,
                                                             float* __restrict__ part) {
    __shared__ __align__(16) float red[16 * 64];
    const int q = threadIdx.x & 15, rb = threadIdx.x >> 4;
    const int chunk = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
    const float* xb = x + (size_t)b * N * 64;
    const int r0 = chunk * CENTER_ROWS, r1 = min(N, r0 + CENTER_ROWS);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0 + rb; r < r1; r += 16) {
        const float4 v = reinterpret_cast<const float4*>(xb + (size_t)r * 64)[q];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(&red[rb * 64 + 4 * q]) = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r * 64 + threadIdx.x];
        part[((size_t)b * nchunk + chunk) * 64 + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(256) void center_apply_kernel(const float* __restrict__ x, int N,
                                                           const float* __restrict__ part, float* __restrict__ out) {
    const int q = threadIdx.x & 15, rb = threadIdx.x >> 4;
    const int chunk = blockIdx.x, b = blockIdx.y, nchunk = gridDim.x;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < nchunk; ++c) {
        const float4 v = reinterpret_cast<const float4*>(part + ((size_t)b * nchunk + c) * 64)[q];
        m.x += v.x; m.y += v.y; m.z += v.z; m.w += v.w;
    }
    const float inv = 1.f / (float)N;
    m.x *= inv; m.y *= inv; m.z *= inv; m.w *= inv;
    const float* xb = x + (size_t)b * N * 64;
    float* ob = out + (size_t)b * N * 64;
    const int r0 = chunk * CENTER_ROWS, r1 = min(N, r0 + CENTER_ROWS);
    for (int r = r0 + rb; r < r1; r += 16) {
        float4 v = reinterpret_cast<const float4*>(xb + (size_t)r * 64)[q];
        v.x -= m.x; v.y -= m.y; v.z -= m.z; v.w -= m.w;
        reinterpret_cast<float4*>(ob + (size_t)r * 64)[q] = v;
    }
}

}  // namespace cmvae

using namespace cmvae;

extern "C" int cmvae_center_rows_chunks(int N) {
    cmvae::clear_stale_error();
    return (N + CENTER_ROWS - 1) / CENTER_ROWS;
}

extern "C" int cmvae_center_rows(const float* x, int B, int N, float* part, float* out, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(x && part && out, "cmvae_center_rows: NULL tensor");
    CMVAE_REQUIRE(B >= 1 && B <= 65535 && N >= 1, "cmvae_center_rows: bad shape B=%d N=%d", B, N);
    const dim3 grid((N + CENTER_ROWS - 1) / CENTER_ROWS, B);
    hipLaunchKernelGGL(center_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, N, part);
    if (int e = check_launch("center_partial_kernel")) return e;
    hipLaunchKernelGGL(center_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, N, (const float*)part, out);
    return check_launch("center_apply_kernel");
}

extern "C" int cmvae_reparam_logq_fwd(const float* mean, const float* logvar, const float* eps, int rows, int M,
                                      float* z, float* logq, void* stream) {
    cmvae::clear_stale_error();
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
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(logvar && eps && dmean && dlogvar, "cmvae_reparam_logq_bwd: NULL tensor");
    CMVAE_REQUIRE(rows >= 0 && M > 0, "cmvae_reparam_logq_bwd: bad shape");
    if (rows == 0) return 0;
    if (M >= 64 && rows <= 4 * compute_units()) {        // few rows, long sample chains: sixteen waves per row
        hipLaunchKernelGGL(reparam_logq_bwd_wide_kernel<16>, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logvar, eps,
                           gz, glogq, M, dmean, dlogvar);
        return check_launch("reparam_logq_bwd_wide_kernel");
    }
    hipLaunchKernelGGL(reparam_logq_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logvar, eps,
                       gz, glogq, rows, M, dmean, dlogvar);
    return check_launch("reparam_logq_bwd_kernel");
}

extern "C" int cmvae_expm_trace(const float* A, float* h, float* grad, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(A && h, "cmvae_expm_trace: NULL tensor");
    hipLaunchKernelGGL(expm_trace_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, A, h, grad);
    return check_launch("expm_trace_kernel");
}

// ============================================================================ Linear-layer helpers
namespace cmvae {

// dZ = dY * act'(pre-activation), from the saved post-activation Y (ReLU: y > 0; ELU: y > 0 ? 1 : y + 1)
template <int ACT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float4* __restrict__ dY, const float4* __restrict__ Y,
                                                      long n4, float4* __restrict__ dZ) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        const float4 g = load_once(reinterpret_cast<const float*>(dY + e)), y = load_once(reinterpret_cast<const float*>(Y + e));
        float4 o;
        if (ACT == 1) {
            o.x = y.x > 0.f ? g.x : 0.f; o.y = y.y > 0.f ? g.y : 0.f;
            o.z = y.z > 0.f ? g.z : 0.f; o.w = y.w > 0.f ? g.w : 0.f;
        } else {
            o.x = y.x > 0.f ? g.x : g.x * (y.x + 1.f); o.y = y.y > 0.f ? g.y : g.y * (y.y + 1.f);
            o.z = y.z > 0.f ? g.z : g.z * (y.z + 1.f); o.w = y.w > 0.f ? g.w : g.w * (y.w + 1.f);
        }
        store_once(reinterpret_cast<float*>(dZ + e), o);
    }
}

// dZ = act'(.) * gscale * 2 (Y - target[row / M]) from the saved post-activation Y [rows, F]: the backward of the
// squared-error sum of the feature models (mimgnet/model.py:391) through the last proj activation
template <int ACT>
__global__ __launch_bounds__(256) void mse_act_bwd_kernel(const float4* __restrict__ Y, const float4* __restrict__ T,
                                                          long n4, int f4, int M, const float* __restrict__ g_dev,
                                                          float gmul, float4* __restrict__ dZ) {
    const float gs = 2.f * gmul * (g_dev ? g_dev[0] : 1.f);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        const long row = e / f4;
        const int c = (int)(e - row * f4);
        const float4 y = Y[e], t = T[(row / M) * f4 + c];
        float4 g = make_float4(gs * (y.x - t.x), gs * (y.y - t.y), gs * (y.z - t.z), gs * (y.w - t.w)), o;
        if (ACT == 0) {
            o = g;
        } else if (ACT == 1) {
            o.x = y.x > 0.f ? g.x : 0.f; o.y = y.y > 0.f ? g.y : 0.f;
            o.z = y.z > 0.f ? g.z : 0.f; o.w = y.w > 0.f ? g.w : 0.f;
        } else {
            o.x = y.x > 0.f ? g.x : g.x * (y.x + 1.f); o.y = y.y > 0.f ? g.y : g.y * (y.y + 1.f);
            o.z = y.z > 0.f ? g.z : g.z * (y.z + 1.f); o.w = y.w > 0.f ? g.w : g.w * (y.w + 1.f);
        }
        dZ[e] = o;
    }
}

// ---------------------------------------------------------------- NOTEARS-MLP tail on bulk rows
// eps[r][j] = x[r][j] - (sum_m hid[r][32j+m] * w2[j][m] + b2[j]),  hid = elu(fc1(x)) from the Linear kernel.
// 8 lanes x float4 cover the 32 hidden units of one (row, j).
__global__ __launch_bounds__(256) void causal_lc_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ hid,
                                                            const float* __restrict__ w2,
                                                            const float* __restrict__ b2, long items,
                                                            float* __restrict__ eps) {
    const int l8 = threadIdx.x & 7;
    for (long it = (long)blockIdx.x * 32 + (threadIdx.x >> 3); it < items; it += (long)gridDim.x * 32) {
        const int j = (int)(it & 63);
        const float4 h = load_once(hid + it * M1 + 4 * l8);
        const float4 w = *reinterpret_cast<const float4*>(w2 + j * M1 + 4 * l8);
        float s = fmaf(h.x, w.x, fmaf(h.y, w.y, fmaf(h.z, w.z, h.w * w.w)));
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (l8 == 0) eps[it] = x[it] - (s + b2[j]);
    }
}

// backward of the tail: g = -d_eps[r][j];  d_pre[r][32j+m] = g * w2[j][m] * elu'(.);  per-block partials of
// dw2[j][m] = sum_r g * hid[r][32j+m], db2[j] = sum_r g and db1 = column sums of d_pre (fc1's bias gradient).
// 512 threads = one float4 of every hidden column.
constexpr int LC_BWD_L = HID + D + HID;
__global__ __launch_bounds__(512) void causal_lc_bwd_kernel(const float* __restrict__ d_eps,
                                                            const float* __restrict__ hid,
                                                            const float* __restrict__ w2, int rows,
                                                            float* __restrict__ d_pre, float* __restrict__ part) {
    const int t = threadIdx.x, j = t >> 3;
    const float4 w = *reinterpret_cast<const float4*>(w2 + 4 * t);
    float4 dw = make_float4(0.f, 0.f, 0.f, 0.f), d1 = make_float4(0.f, 0.f, 0.f, 0.f);
    float db = 0.f;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const float g = -d_eps[(long)r * D + j];
        const float4 h = load_once(hid + (long)r * HID + 4 * t);
        float4 o;
        o.x = g * w.x * (h.x > 0.f ? 1.f : h.x + 1.f);
        o.y = g * w.y * (h.y > 0.f ? 1.f : h.y + 1.f);
        o.z = g * w.z * (h.z > 0.f ? 1.f : h.z + 1.f);
        o.w = g * w.w * (h.w > 0.f ? 1.f : h.w + 1.f);
        store_once(d_pre + (long)r * HID + 4 * t, o);
        dw.x = fmaf(g, h.x, dw.x); dw.y = fmaf(g, h.y, dw.y);
        dw.z = fmaf(g, h.z, dw.z); dw.w = fmaf(g, h.w, dw.w);
        d1.x += o.x; d1.y += o.y; d1.z += o.z; d1.w += o.w;
        db += g;
    }
    float* pb = part + (long)blockIdx.x * LC_BWD_L;
    *reinterpret_cast<float4*>(pb + 4 * t) = dw;
    if ((t & 7) == 0) pb[HID + j] = db;
    *reinterpret_cast<float4*>(pb + HID + D + 4 * t) = d1;
}

// ---------------------------------------------------------------- Adam with L2 weight decay (torch.optim.Adam)
// g' = g + wd*p; m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                   float4* __restrict__ m, float4* __restrict__ v, long n4, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1,
                                                   float bc2_sqrt) {
    const float step = lr / bc1;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        float4 pp = p[e], gg = g[e], mm = m[e], vv = v[e];
#define CMVAE_ADAM1(c)                                                   \
        {                                                                \
            const float gr = fmaf(wd, pp.c, gg.c);                       \
            mm.c = fmaf(b1, mm.c, (1.f - b1) * gr);                      \
            vv.c = fmaf(b2, vv.c, (1.f - b2) * gr * gr);                 \
            pp.c -= step * (mm.c / (sqrtf(vv.c) / bc2_sqrt + eps));      \
        }
        CMVAE_ADAM1(x) CMVAE_ADAM1(y) CMVAE_ADAM1(z) CMVAE_ADAM1(w)
#undef CMVAE_ADAM1
        p[e] = pp; m[e] = mm; v[e] = vv;
    }
}

// the same update with the step count read from device memory (the launch can be replayed from a HIP graph: the caller
// increments *step_dev before every replay); bias corrections in double like the host does for adam_kernel
__global__ __launch_bounds__(256) void adam_devstep_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                           float4* __restrict__ m, float4* __restrict__ v, long n4,
                                                           const long long* __restrict__ step_dev, float lr, float b1,
                                                           float b2, float eps, float wd) {
    const double t = (double)step_dev[0];
    const float bc1 = (float)(1.0 - pow((double)b1, t));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    const float step = lr / bc1;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        float4 pp = p[e], gg = g[e], mm = m[e], vv = v[e];
#define CMVAE_ADAM1(c)                                                   \
        {                                                                \
            const float gr = fmaf(wd, pp.c, gg.c);                       \
            mm.c = fmaf(b1, mm.c, (1.f - b1) * gr);                      \
            vv.c = fmaf(b2, vv.c, (1.f - b2) * gr * gr);                 \
            pp.c -= step * (mm.c / (sqrtf(vv.c) / bc2_sqrt + eps));      \
        }
        CMVAE_ADAM1(x) CMVAE_ADAM1(y) CMVAE_ADAM1(z) CMVAE_ADAM1(w)
#undef CMVAE_ADAM1
        p[e] = pp; m[e] = mm; v[e] = vv;
    }
}

static int grid_for(long n, int per_block) {
    long b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace cmvae

extern "C" int cmvae_act_bwd(const float* dY, const float* Y, int64_t n, int act, float* dZ, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(dY && Y && dZ && n >= 0 && n % 4 == 0 && (act == 1 || act == 2), "cmvae_act_bwd: bad arguments");
    if (n == 0) return 0;
    const long n4 = n / 4;
    if (act == 1)
        hipLaunchKernelGGL(act_bwd_kernel<1>, dim3(grid_for(n4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)dY, (const float4*)Y, n4, (float4*)dZ);
    else
        hipLaunchKernelGGL(act_bwd_kernel<2>, dim3(grid_for(n4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)dY, (const float4*)Y, n4, (float4*)dZ);
    return check_launch("act_bwd_kernel");
}

extern "C" int cmvae_mse_act_bwd(const float* Y, const float* target, int rows, int F, int M, int act,
                                 const float* g_dev, float gmul, float* dZ, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(Y && target && dZ && rows >= 1 && F >= 4 && F % 4 == 0 && M >= 1 && rows % M == 0 && act >= 0 && act <= 2,
                  "cmvae_mse_act_bwd: bad arguments");
    const long n4 = (long)rows * F / 4;
    const dim3 grid(grid_for(n4, 256));
#define CMVAE_MSE_BWD(ACT) hipLaunchKernelGGL(mse_act_bwd_kernel<ACT>, grid, dim3(256), 0, (hipStream_t)stream, \
        (const float4*)Y, (const float4*)target, n4, F / 4, M, g_dev, gmul, (float4*)dZ)
    if (act == 0) CMVAE_MSE_BWD(0); else if (act == 1) CMVAE_MSE_BWD(1); else CMVAE_MSE_BWD(2);
#undef CMVAE_MSE_BWD
    return check_launch("mse_act_bwd_kernel");
}

extern "C" int cmvae_causal_lc_fwd(const float* x, const float* hid, const float* w2, const float* b2, int rows,
                                   float* eps, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(x && hid && w2 && b2 && eps && rows >= 0, "cmvae_causal_lc_fwd: bad arguments");
    if (rows == 0) return 0;
    const long items = (long)rows * D;
    hipLaunchKernelGGL(causal_lc_fwd_kernel, dim3(grid_for(items, 32)), dim3(256), 0, (hipStream_t)stream, x, hid, w2,
                       b2, items, eps);
    return check_launch("causal_lc_fwd_kernel");
}

extern "C" int cmvae_causal_lc_bwd_grid(int rows) {
    cmvae::clear_stale_error(); return rows < 1024 ? (rows > 0 ? rows : 1) : 1024; }
extern "C" int cmvae_causal_lc_bwd_partial_len(void) {
    cmvae::clear_stale_error(); return LC_BWD_L; }

extern "C" int cmvae_causal_lc_bwd(const float* d_eps, const float* hid, const float* w2, int rows, float* d_pre,
                                   float* part, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(d_eps && hid && w2 && d_pre && part && rows >= 1, "cmvae_causal_lc_bwd: bad arguments");
    hipLaunchKernelGGL(causal_lc_bwd_kernel, dim3(cmvae_causal_lc_bwd_grid(rows)), dim3(512), 0, (hipStream_t)stream,
                       d_eps, hid, w2, rows, d_pre, part);
    return check_launch("causal_lc_bwd_kernel");
}

extern "C" int cmvae_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                               int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                               void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && n % 4 == 0 && step >= 1,
                  "cmvae_adam_step: bad arguments (n must be a multiple of 4, step >= 1)");
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (float4*)param,
                       (const float4*)grad, (float4*)exp_avg, (float4*)exp_avg_sq, n / 4, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2));
    return check_launch("adam_kernel");
}

extern "C" int cmvae_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const int64_t* step_dev, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_dev && n >= 0 && n % 4 == 0,
                  "cmvae_adam_step_dev: bad arguments (n must be a multiple of 4)");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cmvae::adam_devstep_kernel, dim3(cmvae::grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       (float4*)param, (const float4*)grad, (float4*)exp_avg, (float4*)exp_avg_sq, n / 4,
                       (const long long*)step_dev, lr, beta1, beta2, eps, weight_decay);
    return cmvae::check_launch("adam_devstep_kernel");
}

// ============================================================================ ELBO tail
// The scalar tail of GMVAE.forward (omniglot/model.py:468-489) in two launches instead of ~25 ATen ones, and its
// backward in one instead of ~30:
//   kl    = mean(log_q_z) - mean(log_p_z) - mean_rows(glp(q_eps, 0))        glp(x,0) = sum_d(-0.5 x_d^2 + c)
//   h_aug = h_lambda * h + 0.5 * h_c * h^2 + a_lambda * sum|A|
// and the prior log-weights of the mixture term (:455-461),
//   logw[b,k] = log(pi[b,k]) + glp(x[b,k,:], mean = log(trace[b,k]), logvar = 0)
// (the reference passes log(trace_k) in gaussian_log_prob's MEAN slot, :483-490, and so does this)
namespace cmvae {

constexpr int TAIL_BLOCKS = 256;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// part[blk] = {sum log_q, sum log_p, sum q_eps^2, sum |A|} over the block's grid-stride share (fixed order)
__global__ __launch_bounds__(256) void elbo_tail_partial_kernel(const float* __restrict__ logq, long n1,
                                                                const float* __restrict__ logp, long n2,
                                                                const float4* __restrict__ qeps4, long n3,
                                                                const float* __restrict__ A, long nA,
                                                                float* __restrict__ part) {
    __shared__ float red[4];
    const long t = (long)blockIdx.x * 256 + threadIdx.x, T = (long)gridDim.x * 256;
    float s0 = 0.f, s1 = 0.f, s3 = 0.f;
    float q0 = 0.f, q1 = 0.f;
    for (long e = t; e < n1; e += T) s0 += logq[e];
    for (long e = t; e < n2; e += T) s1 += logp[e];
    long e = t;
    for (; e + T < n3; e += 2 * T) {
        const float4 a = qeps4[e], b = qeps4[e + T];
        q0 = fmaf(a.x, a.x, q0); q0 = fmaf(a.y, a.y, q0); q0 = fmaf(a.z, a.z, q0); q0 = fmaf(a.w, a.w, q0);
        q1 = fmaf(b.x, b.x, q1); q1 = fmaf(b.y, b.y, q1); q1 = fmaf(b.z, b.z, q1); q1 = fmaf(b.w, b.w, q1);
    }
    if (e < n3) {
        const float4 a = qeps4[e];
        q0 = fmaf(a.x, a.x, q0); q0 = fmaf(a.y, a.y, q0); q0 = fmaf(a.z, a.z, q0); q0 = fmaf(a.w, a.w, q0);
    }
    for (long k = t; k < nA; k += T) s3 += fabsf(A[k]);
    const float r0 = block_sum_256(s0, red), r1 = block_sum_256(s1, red), r2 = block_sum_256(q0 + q1, red),
                r3 = block_sum_256(s3, red);
    if (threadIdx.x == 0) {
        float* p = part + 4 * blockIdx.x;
        p[0] = r0; p[1] = r1; p[2] = r2; p[3] = r3;
    }
}

// out = {kl, h_aug}
__global__ __launch_bounds__(256) void elbo_tail_final_kernel(const float* __restrict__ part, int nblk,
                                                              const float* __restrict__ h, float n1, float n2,
                                                              float rows, float dim, float h_lambda, float h_c,
                                                              float a_lambda, float* __restrict__ out) {
    __shared__ float red[4];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblk; b += 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += part[4 * b + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = block_sum_256(s[k], red);
    if (threadIdx.x == 0) {
        const float hv = h[0];
        const float log_p_eps_mean = -0.5f * s[2] / rows + dim * LOG_NORM;
        out[0] = s[0] / n1 - s[1] / n2 - log_p_eps_mean;
        out[1] = h_lambda * hv + 0.5f * h_c * hv * hv + a_lambda * s[3];
    }
}

// g = {d kl, d h_aug} (device scalars):  d log_q = g0/n1, d log_p = -g0/n2, d q_eps = g0 * q_eps / rows,
// dA = g1 * a_lambda * sign(A), dh = g1 * (h_lambda + h_c * h)
__global__ __launch_bounds__(256) void elbo_tail_bwd_kernel(const float* __restrict__ g, const float* __restrict__ h,
                                                            const float4* __restrict__ qeps4, long n3,
                                                            const float* __restrict__ A, long nA, long n1, long n2,
                                                            float rows, float h_lambda, float h_c, float a_lambda,
                                                            float* __restrict__ d_logq, float* __restrict__ d_logp,
                                                            float4* __restrict__ d_qeps4, float* __restrict__ dA,
                                                            float* __restrict__ dh) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x, T = (long)gridDim.x * 256;
    const float g0 = g[0], g1 = g[1];
    const float a = g0 / (float)n1, b = -g0 / (float)n2, c = g0 / rows;
    for (long e = t; e < n1; e += T) d_logq[e] = a;
    for (long e = t; e < n2; e += T) d_logp[e] = b;
    for (long e = t; e < n3; e += T) {
        const float4 v = qeps4[e];
        d_qeps4[e] = make_float4(c * v.x, c * v.y, c * v.z, c * v.w);
    }
    for (long k = t; k < nA; k += T) {
        const float v = A[k];
        dA[k] = g1 * a_lambda * (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f));
    }
    if (t == 0) dh[0] = g1 * (h_lambda + h_c * h[0]);
}

// logw[r] = log(pi[r]) + sum_d(-0.5 * (x[r][d] - lt)^2 + c),  lt = log(trace[r]);  one wave per row, D = 64
__global__ __launch_bounds__(64) void prior_logw_fwd_kernel(const float* __restrict__ pi, const float* __restrict__ trace,
                                                            const float* __restrict__ x, float* __restrict__ logw) {
    const int r = blockIdx.x, d = threadIdx.x;
    const float lt = logf(trace[r]);
    const float v = x[(long)r * 64 + d] - lt;
    float p = -0.5f * (v * v) + LOG_NORM;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
    if (d == 0) logw[r] = logf(pi[r]) + p;
}

// dx[r][d] = -g[r] * (x[r][d] - log(trace[r]))
__global__ __launch_bounds__(64) void prior_logw_bwd_kernel(const float* __restrict__ trace, const float* __restrict__ x,
                                                            const float* __restrict__ g, float* __restrict__ dx) {
    const int r = blockIdx.x, d = threadIdx.x;
    dx[(long)r * 64 + d] = -g[r] * (x[(long)r * 64 + d] - logf(trace[r]));
}

}  // namespace cmvae

extern "C" int cmvae_elbo_tail_blocks(void) { cmvae::clear_stale_error(); return cmvae::TAIL_BLOCKS; }

extern "C" int cmvae_elbo_tail_fwd(const float* log_q, int64_t n1, const float* log_p, int64_t n2, const float* q_eps,
                                   int64_t rows, int dim, const float* h, const float* A, int64_t nA, float h_lambda,
                                   float h_c, float a_lambda, float* part, float* out, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(log_q && log_p && q_eps && h && A && part && out, "cmvae_elbo_tail_fwd: NULL argument");
    CMVAE_REQUIRE(n1 >= 1 && n2 >= 1 && rows >= 1 && dim >= 4 && dim % 4 == 0 && nA >= 0,
                  "cmvae_elbo_tail_fwd: empty tensor or latent size not a multiple of 4");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cmvae::elbo_tail_partial_kernel, dim3(cmvae::TAIL_BLOCKS), dim3(256), 0, st, log_q, (long)n1, log_p,
                       (long)n2, (const float4*)q_eps, (long)(rows * dim / 4), A, (long)nA, part);
    if (int e = cmvae::check_launch("elbo_tail_partial_kernel")) return e;
    hipLaunchKernelGGL(cmvae::elbo_tail_final_kernel, dim3(1), dim3(256), 0, st, (const float*)part, cmvae::TAIL_BLOCKS, h,
                       (float)n1, (float)n2, (float)rows, (float)dim, h_lambda, h_c, a_lambda, out);
    return cmvae::check_launch("elbo_tail_final_kernel");
}

extern "C" int cmvae_elbo_tail_bwd(const float* g, const float* h, const float* q_eps, int64_t rows, int dim,
                                   const float* A, int64_t nA, int64_t n1, int64_t n2, float h_lambda, float h_c,
                                   float a_lambda, float* d_log_q, float* d_log_p, float* d_q_eps, float* dA, float* dh,
                                   void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(g && h && q_eps && A && d_log_q && d_log_p && d_q_eps && dA && dh, "cmvae_elbo_tail_bwd: NULL argument");
    CMVAE_REQUIRE(n1 >= 1 && n2 >= 1 && rows >= 1 && dim >= 4 && dim % 4 == 0 && nA >= 0,
                  "cmvae_elbo_tail_bwd: empty tensor or latent size not a multiple of 4");
    const long n3 = rows * dim / 4;
    hipLaunchKernelGGL(cmvae::elbo_tail_bwd_kernel, dim3(cmvae::grid_for(n3, 256)), dim3(256), 0, (hipStream_t)stream, g, h,
                       (const float4*)q_eps, n3, A, (long)nA, (long)n1, (long)n2, (float)rows, h_lambda, h_c, a_lambda,
                       d_log_q, d_log_p, (float4*)d_q_eps, dA, dh);
    return cmvae::check_launch("elbo_tail_bwd_kernel");
}

extern "C" int cmvae_prior_logw_fwd(const float* pi, const float* trace, const float* x, int rows, float* logw,
                                    void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(pi && trace && x && logw && rows >= 0, "cmvae_prior_logw_fwd: bad arguments");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(cmvae::prior_logw_fwd_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, pi, trace, x, logw);
    return cmvae::check_launch("prior_logw_fwd_kernel");
}

extern "C" int cmvae_prior_logw_bwd(const float* trace, const float* x, const float* g, int rows, float* dx,
                                    void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(trace && x && g && dx && rows >= 0, "cmvae_prior_logw_bwd: bad arguments");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(cmvae::prior_logw_bwd_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, trace, x, g, dx);
    return cmvae::check_launch("prior_logw_bwd_kernel");
}
