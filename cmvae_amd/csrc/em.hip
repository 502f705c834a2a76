// Per-episode latent GMM EM (unsupervised + semi-supervised), the causal prior term, the
// batched (I + C C^T)^-1, the mixture log-likelihood (fwd/bwd) and the few-shot posterior.
// gfx950 / wave64.  All episodes of the meta-batch run in one launch (grid.y = episode).
//
// Layout in HBM: z [B,N,64] row-major fp32 (256-B rows); mixture parameters
// mu/logvar [B,K,64], pi/tau/logw [B,K].  Per-block partial sums
// part_N [B,nblk,32], part_S1/part_S2 [B,nblk,K,64] are reduced in a fixed order by the
// finalize kernels (no float atomics: results are run-to-run reproducible).
#include "common.h"
#ifndef CMVAE_ABLATE_EM
#define CMVAE_ABLATE_EM 0      // profiling builds only: 1 = no distance loops, 2 = no M-step MFMA
#endif
#include <mutex>
#include <type_traits>
#include <utility>

#ifndef CMVAE_EM_SERIAL_FIN
#define CMVAE_EM_SERIAL_FIN 0
#endif
namespace cmvae {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ small math helpers
__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expm1_fast(x); }

// ------------------------------------------------------------------ causal residual of ONE row
// xs[64] (LDS) -> es[64] (LDS) = xs - mask(xs).  The whole block of NW waves participates (each wave owns
// 64/NW output dims = 2048/NW hidden units, so NW*8 weight loads are in flight per CU);
// caller must __syncthreads() before reading es.
template <int NW>
__device__ __forceinline__ void causal_row(const float* xs, float* es, const cmvae_causal& c) {
    const int tid = threadIdx.x;
    if (c.linear) {
        if (tid < D) {
            float acc = 0.f;
#pragma unroll 8
            for (int i = 0; i < D; ++i) acc = fmaf(xs[i], c.W[i * D + tid], acc);
            es[tid] = xs[tid] - acc;
        }
        return;
    }
    const int lane = tid & 63, w = tid >> 6;
    const int q = lane & 15;     // which float4 of the 64-wide fc1 row
    const int sub = lane >> 4;   // which of the 4 hidden units of a 1-KiB packet
    const float4 xv = reinterpret_cast<const float4*>(xs)[q];
    // every load of a pair of outputs is requested before the first is used (the kernels that call this are chains of
    // L2 round trips: 16 weight packets and their bias / output-weight scalars in flight per lane)
#pragma unroll 2
    for (int jj = 0; jj < 64 / NW; ++jj) {
        const int j = (64 / NW) * w + jj;
        float acc = 0.f;
        float4 wvs[8];
        float b1s[8], w2s[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int u = 32 * j + 4 * it + sub;
            wvs[it] = reinterpret_cast<const float4*>(c.W + (size_t)u * D)[q];
            b1s[it] = c.b1[u];
            w2s[it] = c.w2[u];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const float4 wv = wvs[it];
            float p = wv.x * xv.x;
            p = fmaf(wv.y, xv.y, p);
            p = fmaf(wv.z, xv.z, p);
            p = fmaf(wv.w, xv.w, p);
            p += __shfl_xor(p, 1);
            p += __shfl_xor(p, 2);
            p += __shfl_xor(p, 4);
            p += __shfl_xor(p, 8);
            acc = fmaf(elu_f(p + b1s[it]), w2s[it], acc);
        }
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        if (lane == 0) es[j] = xs[j] - (acc + c.b2[j]);
    }
}

// sum over 64 values held one per thread of the first wave... generic block reduce of 64 floats in LDS
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// glp(eps, log tau) for a row held in LDS: sum_j -0.5*(eps_j - lt)^2 + c   (wave 0 only)
__device__ __forceinline__ float prior_term_wave0(const float* es, float log_tau) {
    const int lane = threadIdx.x & 63;
    float d = es[lane] - log_tau;
    float v = -0.5f * (d * d) + LOG_NORM;
    return wave_sum(v);
}

// ------------------------------------------------------------------ kernel: causal on rows
__global__ __launch_bounds__(1024) void causal_rows_kernel(const float* __restrict__ x, int rows, cmvae_causal c,
                                                          const float* __restrict__ pi,
                                                          const float* __restrict__ tau,
                                                          float* __restrict__ eps, float* __restrict__ logw) {
    __shared__ __align__(16) float xs[D];
    __shared__ __align__(16) float es[D];
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        if (threadIdx.x < D) xs[threadIdx.x] = x[(size_t)r * D + threadIdx.x];
        __syncthreads();
        causal_row<16>(xs, es, c);
        __syncthreads();
        if (threadIdx.x < D) {
            if (eps) eps[(size_t)r * D + threadIdx.x] = es[threadIdx.x];
            if (logw) {
                float p = prior_term_wave0(es, logf(tau[r]));
                if (threadIdx.x == 0) logw[r] = logf(pi[r]) + p;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ kernels: (I + C C^T)^-1
// C row i = causal(s_i e_i).  Two launches per batch of matrices:
//   causal_diag_kernel  grid (16, nmat): wave = one output dim j, lane = input dim i; writes C^T into out[m]
//                       (the one-hot rows need no GEMM: 2048 fused multiply-adds + ELUs per wave).  Spreading a
//                       matrix over 16 workgroups matters when nmat < #CUs (few-shot evaluation: 80 matrices).
//   gram_inverse_kernel grid nmat, 4 waves: Gram matrix on the MFMA (one 32x32 tile per wave), then ONE wave inverts
//                       it in registers (lane = row), overwriting out[m].
__global__ __launch_bounds__(256) void causal_diag_kernel(const float* __restrict__ logvar, float s0, cmvae_causal c,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63, j = 4 * blockIdx.x + (threadIdx.x >> 6);
    const int m = blockIdx.y;
    float s = s0;
    if (logvar) s = expf(0.5f * logvar[(size_t)m * D + lane]) * s0;
    float v;
    if (c.linear) {
        // C = S (I - A): C[i][j] = s_i (delta_ij - A[i][j])
        v = s * ((lane == j ? 1.f : 0.f) - c.W[lane * D + j]);
    } else {
        float acc0 = 0.f, acc1 = 0.f;
#if defined(CMVAE_ABLATE_GRAM) && CMVAE_ABLATE_GRAM == 2
        for (int mm = 0; mm < 2; mm += 2) {
#else
#pragma unroll 16
        for (int mm = 0; mm < M1; mm += 2) {
#endif
            const int u = 32 * j + mm;
            const float pre0 = fmaf(s, c.W[(size_t)u * D + lane], c.b1[u]);
            const float pre1 = fmaf(s, c.W[(size_t)(u + 1) * D + lane], c.b1[u + 1]);
            acc0 = fmaf(elu_f(pre0), c.w2[u], acc0);
            acc1 = fmaf(elu_f(pre1), c.w2[u + 1], acc1);
        }
        v = (lane == j ? s : 0.f) - ((acc0 + acc1) + c.b2[j]);
    }
    out[(size_t)m * D * D + j * D + lane] = v;      // C^T[j][i]
}

constexpr int MS_LD = 65;     // Gram matrix pitch in LDS: lane = row reads are conflict-free

// In-place Gauss-Jordan without pivoting (SPD, eigenvalues >= 1) by one wave, lane = row r, the row in 64 registers.
// After the pivots P = {0..p-1} the working matrix is [[A_PP^-1, X], [-X^T, S]] with S symmetric, so the pivot
// row p is the pivot COLUMN up to a sign: every lane publishes its own column-p element (one ds_write_b32) and
// reads the row back as 16 broadcast b128 reads -- no cross-lane register traffic, no barrier (one wave: LDS
// operations complete in order).  The scaling of a finished pivot row is kept as a pending factor `rs`
// (true row = rs * own): later updates of that row are then the same fused multiply-add as for every other row.
// The column for step p+1 is updated and published first, so its LDS round trip overlaps the other 62 updates.
template <int P>
__device__ __forceinline__ void gauss_jordan_step(float (&own)[D], float* prow2, int lane, float& rs) {
    const float* buf = prow2 + (P & 1) * D;
    const float pivot = buf[P];
    float dinv = __builtin_amdgcn_rcpf(pivot);
    dinv = fmaf(fmaf(-pivot, dinv, 1.f), dinv, dinv);
    const bool is_p = lane == P;
    const float nfd = is_p ? 0.f : -own[P] * dinv;       // minus the (unscaled) multiplier of this row
    rs = is_p ? dinv : rs;
    if constexpr (P + 1 < D) {
        own[P + 1] = fmaf(nfd, buf[P + 1], own[P + 1]);
        float* nbuf = prow2 + ((P + 1) & 1) * D;
        nbuf[lane] = lane < P + 1 ? -rs * own[P + 1] : own[P + 1];
    }
#pragma unroll
    for (int c4 = 0; c4 < D; c4 += 4) {
        const float4 pr = *reinterpret_cast<const float4*>(buf + c4);
        if (c4 + 0 != P && c4 + 0 != P + 1) own[c4 + 0] = fmaf(nfd, pr.x, own[c4 + 0]);
        if (c4 + 1 != P && c4 + 1 != P + 1) own[c4 + 1] = fmaf(nfd, pr.y, own[c4 + 1]);
        if (c4 + 2 != P && c4 + 2 != P + 1) own[c4 + 2] = fmaf(nfd, pr.z, own[c4 + 2]);
        if (c4 + 3 != P && c4 + 3 != P + 1) own[c4 + 3] = fmaf(nfd, pr.w, own[c4 + 3]);
    }
    own[P] = is_p ? 1.f : nfd;
}

template <int... P>
__device__ __forceinline__ void gauss_jordan_steps(float (&own)[D], float* prow2, int lane, float& rs,
                                                   std::integer_sequence<int, P...>) {
    (gauss_jordan_step<P>(own, prow2, lane, rs), ...);
}

__device__ __forceinline__ void gauss_jordan_wave(float (&own)[D], float* prow2, int lane, float& rs) {
    rs = 1.f;
    prow2[lane] = own[0];
#if defined(CMVAE_ABLATE_GRAM) && CMVAE_ABLATE_GRAM == 1
    gauss_jordan_steps(own, prow2, lane, rs, std::make_integer_sequence<int, 1>{});
#else
    gauss_jordan_steps(own, prow2, lane, rs, std::make_integer_sequence<int, D>{});
#endif
}

// body of the Gram + inverse step for one matrix; the whole 256-thread workgroup calls it with >= 8384 floats of LDS
__device__ __forceinline__ void gram_inverse_body(float* lds, float* __restrict__ o) {
    float* CT = lds;                    // [64][64]  CT[j][i] = C[i][j]
    float* Ms = lds + D * D;            // [64][MS_LD]
    float* prow2 = Ms + D * MS_LD;      // [2][64]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        reinterpret_cast<float4*>(CT)[tid + 256 * i] = reinterpret_cast<const float4*>(o)[tid + 256 * i];
    __syncthreads();
    // M = I + C C^T: wave w owns the 32x32 tile (ti, tk); M[i][k] = sum_j CT[j][i] CT[j][k]
    {
        const int ti = w >> 1, tk = w & 1, l31 = lane & 31, h = lane >> 5;
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
        for (int s = 0; s < D / 2; ++s) {
            const float av = CT[(2 * s + h) * D + 32 * ti + l31];
            const float bv = CT[(2 * s + h) * D + 32 * tk + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * h, col = 32 * tk + l31;
            Ms[row * MS_LD + col] = acc[r] + (row == col ? 1.f : 0.f);
        }
    }
    __syncthreads();
    if (w != 0) return;
    float own[D], rs;
#pragma unroll
    for (int cc = 0; cc < D; ++cc) own[cc] = Ms[lane * MS_LD + cc];
    gauss_jordan_wave(own, prow2, lane, rs);
    // the inverse is symmetric: lane r writes its row as column r, 256 contiguous bytes per store instruction
#pragma unroll
    for (int cc = 0; cc < D; ++cc) o[cc * D + lane] = own[cc] * rs;
}
constexpr int GRAM_LDS_FLOATS = D * D + D * MS_LD + 2 * D;

__global__ __launch_bounds__(256) void gram_inverse_kernel(float* __restrict__ out) {
    __shared__ __align__(16) float lds[GRAM_LDS_FLOATS];
    gram_inverse_body(lds, out + (size_t)blockIdx.x * D * D);
}

static int launch_causal_diag(const float* logvar, float s0, int nmat, const cmvae_causal& c, float* out,
                              hipStream_t st) {
    hipLaunchKernelGGL(causal_diag_kernel, dim3(16, nmat), dim3(256), 0, st, logvar, s0, c, out);
    return check_launch("causal_diag_kernel");
}

static int launch_gram_inverse(const float* logvar, float s0, int nmat, const cmvae_causal& c, float* out,
                               hipStream_t st) {
    if (int e = launch_causal_diag(logvar, s0, nmat, c, out, st)) return e;
    hipLaunchKernelGGL(gram_inverse_kernel, dim3(nmat), dim3(256), 0, st, out);
    return check_launch("gram_inverse_kernel");
}

// ------------------------------------------------------------------ the E/M pass
constexpr int TR = 64;        // rows per tile
constexpr int ZS_LD = 68;     // z tile leading dim: 16-B aligned rows, conflict-free b128 row reads

struct PassArgs {
    const float* z;        // [B,N,64]
    const float* mu;       // [B,K,64]
    const float* logw;     // [B,K]
    const int64_t* labels; // [B,N] or null: responsibilities = one_hot(label)
    const float* g;        // [B,N] or null: weight rows by g (backward)
    float* part_N;         // [B,nblk,32] or null
    float* part_S1;        // [B,nblk,K,64] or null
    float* part_S2;        // [B,nblk,K,64] or null
    float* lse;            // [B,N] or null
    float* dz;             // [B,N,64] or null
    int N, K, tiles_per_block, nblk;
    // side job (semi-supervised loop): the grid's last side_rows rows invert side_nmat matrices whose C^T sits in
    // side_out (gram_inverse_body) while the other rows run the pass -- the two are independent
    float* side_out;
    int side_nmat, side_rows;
};

// responsibilities of a 64-row tile: on entry lls[row][k] holds the joint log-likelihoods; on exit the
// posteriors.  4 threads per row.  Returns the row's logsumexp to the caller.  (posterior_argmax_kernel)
template <int KQ>
__device__ __forceinline__ float softmax_rows(float* lls, int K) {
    constexpr int KP = 4 * KQ;
    constexpr int LD = KP + 1;
    const int row = threadIdx.x >> 2, sub = threadIdx.x & 3;
    float v[KQ];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        const int k = sub + 4 * j;
        v[j] = (k < K) ? lls[row * LD + k] : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < KQ; ++j) sum += expf(v[j] - mx);
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    const float lse = mx + logf(sum);
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
        const int k = sub + 4 * j;
        if (k < KP) lls[row * LD + k] = (k < K) ? expf(v[j] - lse) : 0.f;
    }
    return lse;
}

// ------------------------------------------------------------------ fused E-step / M-step pass
// One workgroup streams 128-row tiles of one episode's latents (32 KB each) through LDS:
//   E-step   2 threads per row (row in registers), each ceil(K/2) components against the LDS-resident means;
//            the pair's softmax is two shuffles; responsibilities go to LDS as R[row][k] (optionally scaled by g);
//   M-step   S1[k][d] += R^T Z (and S2 with Z*Z) on v_mfma_f32_32x32x2_f32: the contraction runs over the rows
//            of the tile, wave = (row half, d half); K <= 32 fills the M side of the tile exactly;
//   N_k      accumulated in the E-step threads' registers, reduced once at the end.
// The next tile's rows are fetched into registers while the current one is processed.  Per-block partials, reduced
// by em_finalize / the reduce kernels in a fixed order: no float atomics, results reproducible.
constexpr int TRP = 128;      // rows per tile

// K <= 16: M-step on v_mfma_f32_16x16x4_f32 (16 components x 16 latent dims per accumulator, wave = 16-column
// block of d), R pitch 16; K <= 32: v_mfma_f32_32x32x2_f32 (wave = (row half, d half)), R pitch 33.
template <int KH, bool MOM2, bool DZ>     // KH: components per E-step thread (K <= 2*KH); second moments; dz output
__global__ __launch_bounds__(256, (KH <= 8 && !DZ) ? 3 : 2) void em_pass_kernel(PassArgs a) {
    constexpr int KP = 2 * KH;
    constexpr bool SMALLK = KH <= 8;
    constexpr int RL = SMALLK ? 20 : 33;
    __shared__ __align__(16) float zs[TRP * ZS_LD];
    constexpr int MP = D;                            // pitch of the means (padding it to 68 measured slower)
    __shared__ __align__(16) float mus[KP * MP];
    __shared__ float Rs[TRP * RL];
    __shared__ float lws[KP];
    if constexpr (MOM2 && !DZ) {
        static_assert(TRP * ZS_LD >= GRAM_LDS_FLOATS, "the side job borrows the tile buffer");
        const int pass_rows = (int)gridDim.y - a.side_rows;
        if ((int)blockIdx.y >= pass_rows) {              // uniform per workgroup
            const int m = ((int)blockIdx.y - pass_rows) * (int)gridDim.x + (int)blockIdx.x;
            if (m < a.side_nmat) gram_inverse_body(zs, a.side_out + (size_t)m * D * D);
            return;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y, blk = blockIdx.x;
    const int N = a.N, K = a.K;
    const float* zb = a.z + (size_t)b * N * D;
    const int erow = tid >> 1, half = tid & 1;      // E-step: 2 threads per row
    const int i = lane & 31, h = lane >> 5;         // MFMA lane coordinates
    const int dh = w & 1, rh = w >> 1;              // wave = (d half, row half)
    const int q = tid & 15, lr = tid >> 4;          // tile load: 16 lanes per row, 16 rows per pass

    if (a.mu) {
        for (int e = tid; e < KP * D; e += 256) mus[(e >> 6) * MP + (e & 63)] = (e < K * D) ? a.mu[(size_t)b * K * D + e] : 0.f;
        if (tid < KP) lws[tid] = (tid < K) ? a.logw[b * K + tid] : -INFINITY;
    }
    for (int e = tid; e < TRP * RL; e += 256) Rs[e] = 0.f;       // columns >= K stay zero for the MFMA
    const bool mstep = a.part_S1 != nullptr;
    constexpr bool mom2 = MOM2;
    floatx16 acc1 = {}, acc2 = {};
    floatx4 q1 = {0.f, 0.f, 0.f, 0.f}, q2 = {0.f, 0.f, 0.f, 0.f};
    float nk[KH];
#pragma unroll
    for (int j = 0; j < KH; ++j) nk[j] = 0.f;
    float mud[DZ ? KP : 1];                          // dz path: this lane's column of the means
    if (DZ) {
#pragma unroll
        for (int k = 0; k < KP; ++k) mud[k] = (k < K) ? a.mu[((size_t)b * K + k) * D + lane] : 0.f;
    }

    const int tile0 = blk * a.tiles_per_block;
    // The next tile is fetched one iteration ahead and used raw: rows beyond N are zeroed when the tile is written
    // to LDS, not when it is loaded (a select right after the load would wait for it and expose the whole HBM
    // latency).
    float4 pre[8];
    auto fetch = [&](int row0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = row0 + lr + 16 * e;
            pre[e] = reinterpret_cast<const float4*>(zb + (size_t)(r < N ? r : 0) * D)[q];
        }
    };
    if (tile0 * TRP < N) fetch(tile0 * TRP);
    for (int t = 0; t < a.tiles_per_block; ++t) {
        const int row0 = (tile0 + t) * TRP;
        if (row0 >= N) break;          // uniform across the block
        __syncthreads();               // previous tile fully consumed (also covers mus / lws / Rs stores)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float4 v = pre[e];
            if (row0 + lr + 16 * e >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&zs[(lr + 16 * e) * ZS_LD + 4 * q]) = v;
        }
        __syncthreads();
        const bool valid = row0 + erow < N;
        // per-row operands from global memory are requested before the prefetch (see above)
        int64_t lab = -1;
        float gv = 1.f;
        if (a.labels && valid) lab = a.labels[(size_t)b * N + row0 + erow];
        if (DZ && a.g && valid) gv = a.g[(size_t)b * N + row0 + erow];
        if (t + 1 < a.tiles_per_block && row0 + TRP < N) fetch(row0 + TRP);
        float r[KH];
        if (a.labels) {
#pragma unroll
            for (int j = 0; j < KH; ++j) r[j] = (lab == half + 2 * j) ? 1.f : 0.f;
        } else {
            // squared distances, the row taken in two halves of 32 dims to keep it in 32 registers (packed
            // v_pk_fma_f32 math was measured: no faster at K = 5, 25 % slower at K = 16)
#pragma unroll
            for (int j = 0; j < KH; ++j) r[j] = 0.f;
#if !(CMVAE_ABLATE_EM & 1)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float zr[32];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float4 v = *reinterpret_cast<const float4*>(&zs[erow * ZS_LD + 32 * hf + 4 * c]);
                    zr[4 * c] = v.x; zr[4 * c + 1] = v.y; zr[4 * c + 2] = v.z; zr[4 * c + 3] = v.w;
                }
#pragma unroll
                for (int j = 0; j < KH; ++j) {
                    const int k = half + 2 * j;
                    // two independent chains of 16 terms per half row, combined pairwise: a single 64-term chain
                    // loses ~4e-5 on the squared distance of an outlier (|ll| ~ 200), which a component that owns
                    // less than one latent turns into a 2e-4 error of its mean; this order stays at the level of the
                    // oracle's vectorised sum (and gives the VALU two chains to interleave)
                    float acc_a = 0.f, acc_b = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float4 m = *reinterpret_cast<const float4*>(&mus[k * MP + 32 * hf + 4 * c]);
                        const float d0 = zr[4 * c] - m.x, d1 = zr[4 * c + 1] - m.y;
                        const float d2 = zr[4 * c + 2] - m.z, d3 = zr[4 * c + 3] - m.w;
                        acc_a = fmaf(d0, d0, acc_a);
                        acc_b = fmaf(d1, d1, acc_b);
                        acc_a = fmaf(d2, d2, acc_a);
                        acc_b = fmaf(d3, d3, acc_b);
                    }
                    r[j] += acc_a + acc_b;
                }
            }
#endif
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < KH; ++j) {
                r[j] = fmaf(-0.5f, r[j], D * LOG_NORM) + lws[half + 2 * j];        // -inf for k >= K
                mx = fmaxf(mx, r[j]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < KH; ++j) sum += __expf(r[j] - mx);     // hardware exp/log: 1e-4 tolerance on EM parameters
            sum += __shfl_xor(sum, 1);
            const float lse = mx + __logf(sum);
            if (a.lse && half == 0 && valid) a.lse[(size_t)b * N + row0 + erow] = lse;
            const float gs = valid ? gv : 0.f;
#pragma unroll
            for (int j = 0; j < KH; ++j) r[j] = __expf(r[j] - lse) * gs;
        }
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            Rs[erow * RL + half + 2 * j] = r[j];
            nk[j] += r[j];
        }
        __syncthreads();
        if (mstep && !(CMVAE_ABLATE_EM & 2)) {
            if (SMALLK) {
                // A[k = lane&15][row], B[row][d = 16w + (lane&15)] with row = 16*(s>>2) + 4*(lane>>4) + (s&3): the four
                // lane groups sit 4 rows apart, which with pitches 20 and 68 puts them in disjoint banks;
                // operands are read one step ahead
                const int i16 = lane & 15, g4 = lane >> 4;
                const float* rp = &Rs[4 * g4 * RL + i16];
                const float* zp = &zs[4 * g4 * ZS_LD + 16 * w + i16];
                float av = rp[0], zv = zp[0];
#pragma unroll 8
                for (int s4 = 0; s4 < TRP / 4; ++s4) {
                    float an = av, zn = zv;
                    if (s4 + 1 < TRP / 4) {
                        const int rn = 16 * ((s4 + 1) >> 2) + ((s4 + 1) & 3);
                        an = rp[rn * RL];
                        zn = zp[rn * ZS_LD];
                    }
                    q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, zv, q1, 0, 0, 0);
                    if (mom2) q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, zv * zv, q2, 0, 0, 0);
                    av = an; zv = zn;
                }
            } else {
                const float* rp = &Rs[(64 * rh + h) * RL + i];
                const float* zp = &zs[(64 * rh + h) * ZS_LD + 32 * dh + i];
                float av = rp[0], zv = zp[0];
#pragma unroll 8
                for (int s2 = 0; s2 < 32; ++s2) {
                    float an = av, zn = zv;
                    if (s2 + 1 < 32) { an = rp[(2 * s2 + 2) * RL]; zn = zp[(2 * s2 + 2) * ZS_LD]; }
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, zv, acc1, 0, 0, 0);
                    if (mom2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, zv * zv, acc2, 0, 0, 0);
                    av = an; zv = zn;
                }
            }
        }
        if (DZ) {
            // dz[row][d] = sum_k w_k mu_k[d] - z[d] * sum_k w_k   (w = g * r); wave w owns rows 32w .. 32w+31
            for (int rr = 0; rr < 32; ++rr) {
                const int rw = 32 * w + rr;
                if (row0 + rw >= N) break;
                float acc = 0.f, ws = 0.f;
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    const float rv = Rs[rw * RL + k];
                    acc = fmaf(rv, mud[k], acc);
                    ws += rv;
                }
                a.dz[((size_t)b * N + row0 + rw) * D + lane] = acc - zs[rw * ZS_LD + lane] * ws;
            }
        }
    }
    // ---- block partials
    __syncthreads();
    // both scratch areas of the epilogue live in buffers that are dead by now, each checked against its size:
    // the S1 (and S2) slabs in the tile buffer, the N_k scratch [128][KP] in the responsibility buffer
    constexpr int RED_FLOATS = (MOM2 ? 4 : 2) * 32 * 64;
    static_assert(RED_FLOATS <= TRP * ZS_LD, "S1/S2 slabs must fit the tile buffer");
    static_assert(TRP * KP <= TRP * RL, "N_k scratch must fit the responsibility buffer");
    float* red = zs;
    float* rn = Rs;                                   // [128][KP]
    if (a.part_N) {
#pragma unroll
        for (int j = 0; j < KH; ++j) rn[erow * KP + half + 2 * j] = nk[j];
    }
    if (mstep) {
        if (SMALLK) {
            // accumulator: row k = 4*(lane>>4) + reg, column d = 16w + (lane&15); the second row-half slab stays zero
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int k = 4 * (lane >> 4) + reg, d = 16 * w + (lane & 15);
                red[k * 64 + d] = q1[reg];
                red[(32 + k) * 64 + d] = 0.f;
                if (mom2) { red[2 * 32 * 64 + k * 64 + d] = q2[reg]; red[2 * 32 * 64 + (32 + k) * 64 + d] = 0.f; }
            }
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int k = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                red[(rh * 32 + k) * 64 + 32 * dh + i] = acc1[reg];
                if (mom2) red[2 * 32 * 64 + (rh * 32 + k) * 64 + 32 * dh + i] = acc2[reg];
            }
        }
    }
    __syncthreads();
    if (a.part_N && tid < 32) {
        float s = 0.f;
        if (tid < KP) {
            for (int r2 = 0; r2 < TRP; ++r2) s += rn[r2 * KP + tid];
        }
        a.part_N[((size_t)b * a.nblk + blk) * 32 + tid] = s;
    }
    if (mstep) {
        for (int e = tid; e < K * D; e += 256) {
            const size_t o = ((size_t)b * a.nblk + blk) * K * D + e;
            a.part_S1[o] = red[e] + red[32 * 64 + e];
            if (mom2) a.part_S2[o] = red[2 * 32 * 64 + e] + red[3 * 32 * 64 + e];
        }
    }
}

// ------------------------------------------------------------------ finalize (M-step tail)
struct FinArgs {
    // partial sums of the unlabeled pass
    const float* part_N; const float* part_S1; const float* part_S2;
    int nblk;
    // supervised sums (semi) — already reduced: [B,K], [B,K,64], [B,K,64]
    const float* sN; const float* sS1; const float* sS2;
    // init: gather rows of z
    const float* z; const int64_t* init_idx; int N;
    const float* minv;      // shared [64,64] (minv_stride == 0) or per (b,k) [B,K,64,64]
    size_t minv_stride;
    int K; int fix_pi; int semi; int init;
    float tau_new;          // gamma1 (or 1 for init)
    cmvae_causal c;
    float* pi; float* mu; float* logvar; float* tau; float* logw;
};

constexpr int FW = 16;      // waves per finalize block

__global__ __launch_bounds__(1024) void em_finalize_kernel(FinArgs a) {
    __shared__ float red[FW * D];
    __shared__ float red2[FW * D];
    __shared__ float nred[32 * 32];
    __shared__ __align__(16) float ms[D];
    __shared__ __align__(16) float xs[D];
    __shared__ __align__(16) float es[D];
    __shared__ float nall[32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y, K = a.K;
#if CMVAE_EM_SERIAL_FIN      // timing only: ONE workgroup per episode finalizes its K components one after the
                             // other — what the last-arriving workgroup of a single-launch EM iteration would have to do
    for (int k = blockIdx.x; k < K; k += gridDim.x) {
    __syncthreads();
#else
    const int k = blockIdx.x;
    {
#endif
    float S1 = 0.f, S2 = 0.f, Nk = 0.f, pi_k = 1.f / K;
    float uN = 0.f;
    // operands that do not depend on the reduction are requested first: their latency hides behind it
    const float* Mi = a.minv + (a.minv_stride ? ((size_t)b * K + k) * a.minv_stride : 0);
    float mi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mi[i] = Mi[(4 * w + i) * D + lane];
    float sup1 = 0.f, sup2 = 0.f, supN = 0.f;
    if (a.semi) {
        supN = a.sN[b * K + k];
        if (tid < D) {
            sup1 = a.sS1[((size_t)b * K + k) * D + tid];
            if (!a.init) sup2 = a.sS2[((size_t)b * K + k) * D + tid];
        }
    }

    if (a.init && !a.semi) {
        // mean0 = z[idx] (then shrunk below)
        // a seed outside [0, N) must never become an out-of-bounds read: it is clamped here; the host side validates
        // indices that originate on the CPU and, with CMVAE_CHECK_INDICES=1, device tensors too (cmvae_amd/ops.py)
        int64_t ix = a.init_idx[b * K + k];
        ix = ix < 0 ? 0 : (ix >= a.N ? (int64_t)a.N - 1 : ix);
        if (tid < D) ms[tid] = a.z[((size_t)b * a.N + (size_t)ix) * D + tid];
    } else if (a.init && a.semi) {
        if (tid < D) ms[tid] = sup1 / supN;
    } else {
        // ---- N_k' for every component (needed for pi), fixed-order reduction
        {
            // (loads four partials ahead: a loop of dependent load -> add round trips to L2 is what this kernel's time
            // was made of; the summation order is unchanged)
            const int kk = tid & 31, ch = tid >> 5;   // 32 chunks of blocks
            float acc = 0.f;
            const float* pn = a.part_N + (size_t)b * a.nblk * 32 + kk;
            int i = ch;
            for (; i + 96 < a.nblk; i += 128) {
                const float v0 = pn[(size_t)i * 32], v1 = pn[(size_t)(i + 32) * 32], v2 = pn[(size_t)(i + 64) * 32],
                            v3 = pn[(size_t)(i + 96) * 32];
                acc += v0; acc += v1; acc += v2; acc += v3;
            }
            for (; i < a.nblk; i += 32) acc += pn[(size_t)i * 32];
            nred[ch * 32 + kk] = acc;
        }
        // ---- S1 (and S2) for this component
        {
            float acc = 0.f, acc2 = 0.f;
            const size_t st = (size_t)K * D;                       // partials of consecutive blocks
            const size_t o0 = ((size_t)b * a.nblk * K + k) * D + lane;
            int i = w;
            if (a.part_S2) {
                for (; i + 3 * FW < a.nblk; i += 4 * FW) {
                    const size_t o = o0 + (size_t)i * st;
                    const float v0 = a.part_S1[o], v1 = a.part_S1[o + FW * st], v2 = a.part_S1[o + 2 * FW * st],
                                v3 = a.part_S1[o + 3 * FW * st];
                    const float u0 = a.part_S2[o], u1 = a.part_S2[o + FW * st], u2 = a.part_S2[o + 2 * FW * st],
                                u3 = a.part_S2[o + 3 * FW * st];
                    acc += v0; acc += v1; acc += v2; acc += v3;
                    acc2 += u0; acc2 += u1; acc2 += u2; acc2 += u3;
                }
                for (; i < a.nblk; i += FW) { acc += a.part_S1[o0 + (size_t)i * st]; acc2 += a.part_S2[o0 + (size_t)i * st]; }
            } else {
                for (; i + 3 * FW < a.nblk; i += 4 * FW) {
                    const size_t o = o0 + (size_t)i * st;
                    const float v0 = a.part_S1[o], v1 = a.part_S1[o + FW * st], v2 = a.part_S1[o + 2 * FW * st],
                                v3 = a.part_S1[o + 3 * FW * st];
                    acc += v0; acc += v1; acc += v2; acc += v3;
                }
                for (; i < a.nblk; i += FW) acc += a.part_S1[o0 + (size_t)i * st];
            }
            red[w * D + lane] = acc;
            red2[w * D + lane] = acc2;
        }
        __syncthreads();
        if (tid < 32) {
            float acc = 0.f;
#pragma unroll
            for (int ch = 0; ch < 32; ++ch) acc += nred[ch * 32 + tid];
            nall[tid] = acc;
        }
        __syncthreads();
        uN = nall[k];
#pragma unroll
        for (int ww = 0; ww < FW; ++ww) { S1 += red[ww * D + lane]; S2 += red2[ww * D + lane]; }
        if (!a.semi) {
            // N guard, then pi from the guarded N (omniglot/model.py:192-207)
            float ntot = 0.f;
            for (int kk = 0; kk < K; ++kk) {
                float v = nall[kk];
                ntot += (v == 0.f) ? 1e-22f : v;
            }
            Nk = (uN == 0.f) ? 1e-22f : uN;
            if (!a.fix_pi) pi_k = Nk / ntot;
            if (tid < D) ms[tid] = S1 / Nk;
        } else {
            Nk = supN + uN;                // joint denominator
            if (tid < D) ms[tid] = (sup1 + S1) / Nk;
        }
    }
    __syncthreads();
    // ---- shrink: mean = m (row vector) @ Minv
    {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = fmaf(ms[4 * w + i], mi[i], acc);
        __syncthreads();             // everyone is done reading red/red2 as partial sums
        red[w * D + lane] = acc;
    }
    __syncthreads();
    if (tid < D) {
        float mean = 0.f;
#pragma unroll
        for (int ww = 0; ww < FW; ++ww) mean += red[ww * D + tid];
        xs[tid] = mean;
        a.mu[((size_t)b * K + k) * D + tid] = mean;
        if (a.semi) {
            float lv = 0.f;
            if (!a.init) {
                const float m2 = mean * mean;
                const float svar = sup2 - 2.f * (mean * sup1) + supN * m2;
                const float uvar = S2 - 2.f * (mean * S1) + uN * m2;
                lv = logf((svar + uvar) / Nk);
            }
            a.logvar[((size_t)b * K + k) * D + tid] = lv;
        }
    }
    __syncthreads();
    // ---- prior term of the next E-step: log pi + glp(causal(mean), log tau)
    causal_row<FW>(xs, es, a.c);
    __syncthreads();
    if (tid < D) {
        const float p = prior_term_wave0(es, logf(a.tau_new));
        if (tid == 0) {
            a.pi[b * K + k] = pi_k;
            a.tau[b * K + k] = a.tau_new;
            a.logw[b * K + k] = logf(pi_k) + p;
        }
    }
    }
}

// reduce the label-pass partials once: sN [B,K], sS1/sS2 [B,K,64]
__global__ __launch_bounds__(64) void sup_reduce_kernel(const float* __restrict__ part_N,
                                                        const float* __restrict__ part_S1,
                                                        const float* __restrict__ part_S2, int nblk, int K,
                                                        float* __restrict__ sN, float* __restrict__ sS1,
                                                        float* __restrict__ sS2) {
    const int k = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    float a1 = 0.f, a2 = 0.f, n = 0.f;
    for (int i = 0; i < nblk; ++i) {
        const size_t o = (((size_t)b * nblk + i) * K + k) * D + lane;
        a1 += part_S1[o];
        a2 += part_S2[o];
        n += part_N[((size_t)b * nblk + i) * 32 + k];
    }
    sS1[((size_t)b * K + k) * D + lane] = a1;
    sS2[((size_t)b * K + k) * D + lane] = a2;
    if (lane == 0) sN[b * K + k] = n;
}

// dlogw[b,k] = sum over blocks of part_N
__global__ __launch_bounds__(64) void partn_reduce_kernel(const float* __restrict__ part_N, int nblk, int K,
                                                          float* __restrict__ out) {
    const int b = blockIdx.x, k = threadIdx.x;
    if (k >= K) return;
    float acc = 0.f;
    for (int i = 0; i < nblk; ++i) acc += part_N[((size_t)b * nblk + i) * 32 + k];
    out[b * K + k] = acc;
}

// ------------------------------------------------------------------ few-shot posterior + argmax
// One block per group of `ipb` query items (rows = ipb*M <= 64) or per item (M > 64).
template <int KQ>
__global__ __launch_bounds__(256) void posterior_argmax_kernel(const float* __restrict__ z,
                                                               const float* __restrict__ mu,
                                                               const float* __restrict__ logvar,
                                                               const float* __restrict__ logw, int te, int M,
                                                               int K, int ipb, int64_t* __restrict__ y_pred,
                                                               float* __restrict__ post_mean) {
    constexpr int KP = 4 * KQ;
    constexpr int LD = KP + 1;
    __shared__ __align__(16) float zs[TR * ZS_LD];
    __shared__ __align__(16) float mus[KP * D];
    __shared__ __align__(16) float lvs[KP * D];
    __shared__ __align__(16) float evs[KP * D];
    __shared__ float lls[TR * LD];
    __shared__ float lws[KP];
    __shared__ float pm[64 * KP];   // [items(<=64)][KP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y;
    const int item0 = blockIdx.x * ipb;
    const int items = min(ipb, te - item0);
    const int N = te * M;
    const float* zb = z + (size_t)b * N * D;
    for (int e = tid; e < KP * D; e += 256) {
        const bool ok = e < K * D;
        const float lv = ok ? logvar[(size_t)b * K * D + e] : 0.f;
        mus[e] = ok ? mu[(size_t)b * K * D + e] : 0.f;
        lvs[e] = lv;
        evs[e] = expf(lv);
    }
    if (tid < KP) lws[tid] = (tid < K) ? logw[b * K + tid] : -INFINITY;
    for (int e = tid; e < 64 * KP; e += 256) pm[e] = 0.f;
    const int rows_total = items * M;
    const int row_base = item0 * M;
    for (int t0 = 0; t0 < rows_total; t0 += TR) {
        __syncthreads();
        // tile rows [row_base+t0, +64) limited to this block's rows
        {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int e = tid + 256 * it;
                const int r = e >> 4, q = e & 15;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t0 + r < rows_total) v = reinterpret_cast<const float4*>(zb + (size_t)(row_base + t0 + r) * D)[q];
                *reinterpret_cast<float4*>(&zs[r * ZS_LD + 4 * q]) = v;
            }
        }
        __syncthreads();
        const int row = lane;
        float zr[D];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(&zs[row * ZS_LD + 4 * q]);
            zr[4 * q] = v.x; zr[4 * q + 1] = v.y; zr[4 * q + 2] = v.z; zr[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const int k = w + 4 * j;
            float acc = 0.f;
#pragma unroll 4
            for (int d4 = 0; d4 < D / 4; ++d4) {   // b128 broadcast reads, the reference's division and summation order
                const float4 m4 = *reinterpret_cast<const float4*>(&mus[k * D + 4 * d4]);
                const float4 l4 = *reinterpret_cast<const float4*>(&lvs[k * D + 4 * d4]);
                const float4 e4 = *reinterpret_cast<const float4*>(&evs[k * D + 4 * d4]);
                const float d0 = zr[4 * d4] - m4.x, d1 = zr[4 * d4 + 1] - m4.y;
                const float d2 = zr[4 * d4 + 2] - m4.z, d3 = zr[4 * d4 + 3] - m4.w;
                acc += l4.x + (d0 * d0) / e4.x;
                acc += l4.y + (d1 * d1) / e4.y;
                acc += l4.z + (d2 * d2) / e4.z;
                acc += l4.w + (d3 * d3) / e4.w;
            }
            lls[row * LD + k] = fmaf(-0.5f, acc, D * LOG_NORM) + lws[k];
        }
        __syncthreads();
        softmax_rows<KQ>(lls, K);
        __syncthreads();
        // accumulate per item: thread (item-local row range) -> pm[item][k]
        // one thread per (item, k) pair sums its rows of this tile in ascending order
        for (int e = tid; e < 64 * KP; e += 256) {
            const int il = e / KP, k = e % KP;      // item slot within the block
            if (il < items && k < K) {
                const int r_lo = max(il * M - t0, 0), r_hi = min((il + 1) * M - t0, TR);
                float acc = pm[e];
                for (int r = r_lo; r < r_hi; ++r) acc += lls[r * LD + k];
                pm[e] = acc;
            }
        }
    }
    __syncthreads();
    if (tid < items) {
        int best = 0;
        float bv = -INFINITY;
        for (int k = 0; k < K; ++k) {
            const float v = pm[tid * KP + k] / (float)M;
            if (post_mean) post_mean[((size_t)b * te + item0 + tid) * K + k] = v;
            if (v > bv) { bv = v; best = k; }
        }
        y_pred[(size_t)b * te + item0 + tid] = best;
    }
}

// ------------------------------------------------------------------ host side
// Split every episode's tiles over as many workgroups as stay resident at once (3 per CU on 256 CUs; the em_pass
// instantiations for K > 16 and the dz variant hold 2 per CU, their second round is a short one): a single
// workgroup per CU would serialise its own load -> E-step -> M-step phases with nothing to overlap them.
static int pick_tiles_per_block(int B, int N, int K, bool dz = false) {
    // degenerate shapes (no episode, no latent) reach this through the *_workspace_bytes queries, which cannot
    // report an error: they are sized as one tile of one episode, the entry points reject them before any launch
    if (B < 1) B = 1;
    int tiles_ep = (N + TRP - 1) / TRP;
    if (tiles_ep < 1) tiles_ep = 1;
    const int resident = compute_units() * ((K <= 16 && !dz) ? 3 : 2);
    int nblk = resident / B;                             // workgroups per episode: at most one resident round
    if (nblk > tiles_ep) nblk = tiles_ep;
    if (nblk < 1) nblk = 1;
    return (tiles_ep + nblk - 1) / nblk;                 // >= 1
}
static int nblk_for(int N, int tpb) {
    int tiles_ep = (N + TRP - 1) / TRP;
    if (tiles_ep < 1) tiles_ep = 1;
    if (tpb < 1) tpb = 1;
    return (tiles_ep + tpb - 1) / tpb;
}

template <typename F>
static int dispatch_kq(int K, F&& f) {
    const int kq = (K + 3) / 4;
    switch (kq) {
        case 1: return f(std::integral_constant<int, 1>{});
        case 2: return f(std::integral_constant<int, 2>{});
        case 3: return f(std::integral_constant<int, 3>{});
        case 4: return f(std::integral_constant<int, 4>{});
        case 5: return f(std::integral_constant<int, 5>{});
        case 6: return f(std::integral_constant<int, 6>{});
        case 7: return f(std::integral_constant<int, 7>{});
        case 8: return f(std::integral_constant<int, 8>{});
    }
    return set_error("component count K=%d outside [1,%d]", K, CMVAE_KMAX);
}

template <int KH>
static void launch_pass_kh(const PassArgs& a, int B, hipStream_t st) {
    const dim3 grid(a.nblk, B + (a.part_S2 && !a.dz ? a.side_rows : 0));
    if (a.dz) hipLaunchKernelGGL((em_pass_kernel<KH, false, true>), grid, dim3(256), 0, st, a);
    else if (a.part_S2) hipLaunchKernelGGL((em_pass_kernel<KH, true, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((em_pass_kernel<KH, false, false>), grid, dim3(256), 0, st, a);
}

static int launch_pass(const PassArgs& a, int B, hipStream_t st) {
    if (a.dz && a.part_S2) return set_error("em_pass: dz and second moments are not produced together");
#define CMVAE_PASS(KH) launch_pass_kh<KH>(a, B, st); break
    switch ((a.K + 1) / 2) {
        case 1: CMVAE_PASS(1); case 2: CMVAE_PASS(2); case 3: CMVAE_PASS(3); case 4: CMVAE_PASS(4);
        case 5: CMVAE_PASS(5); case 6: CMVAE_PASS(6); case 7: CMVAE_PASS(7); case 8: CMVAE_PASS(8);
        case 9: case 10: CMVAE_PASS(10);
        case 11: case 12: CMVAE_PASS(12);
        case 13: case 14: CMVAE_PASS(14);
        case 15: case 16: CMVAE_PASS(16);
        default: return set_error("component count K=%d outside [1,%d]", a.K, CMVAE_KMAX);
    }
#undef CMVAE_PASS
    return check_launch("em_pass_kernel");
}

static int check_causal(const cmvae_causal* c) {
    CMVAE_REQUIRE(c != nullptr, "causal descriptor is NULL");
    CMVAE_REQUIRE(c->W != nullptr, "causal.W is NULL");
    if (!c->linear) CMVAE_REQUIRE(c->b1 && c->w2 && c->b2, "nonlinear causal layer needs b1, w2, b2");
    return 0;
}

struct EmWorkspace {
    float *minv0, *minv, *part_N, *part_S1, *part_S2, *sN, *sS1, *sS2, *spart_N, *spart_S1, *spart_S2;
    size_t bytes;
};

static EmWorkspace carve(void* base, int B, int Nu, int Ns, int K, bool semi) {
    EmWorkspace w{};
    size_t off = 0;
    auto take = [&](size_t nfloat) {
        float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
        off += align_up(nfloat * sizeof(float));
        return p;
    };
    const int tpb = pick_tiles_per_block(B, Nu, K), nblk = nblk_for(Nu, tpb);
    w.minv0 = take(D * D);
    w.minv = take(semi ? (size_t)B * K * D * D : D * D);
    w.part_N = take((size_t)B * nblk * 32);
    w.part_S1 = take((size_t)B * nblk * K * D);
    if (semi) {
        w.part_S2 = take((size_t)B * nblk * K * D);
        const int stpb = pick_tiles_per_block(B, Ns, K), snblk = nblk_for(Ns, stpb);
        w.spart_N = take((size_t)B * snblk * 32);
        w.spart_S1 = take((size_t)B * snblk * K * D);
        w.spart_S2 = take((size_t)B * snblk * K * D);
        w.sN = take((size_t)B * K);
        w.sS1 = take((size_t)B * K * D);
        w.sS2 = take((size_t)B * K * D);
    }
    w.bytes = off;
    return w;
}

}  // namespace cmvae

using namespace cmvae;

extern "C" int cmvae_causal_rows(const float* x, int rows, const cmvae_causal* c, const float* pi, const float* tau,
                                 float* eps, float* logw, void* stream) {
    cmvae::clear_stale_error();
    if (int e = check_causal(c)) return e;
    CMVAE_REQUIRE(x && rows >= 0, "cmvae_causal_rows: bad arguments");
    CMVAE_REQUIRE(!logw || (pi && tau), "cmvae_causal_rows: logw needs pi and tau");
    if (rows == 0) return 0;
    const int grid = rows < 4096 ? rows : 4096;
    hipLaunchKernelGGL(causal_rows_kernel, dim3(grid), dim3(1024), 0, (hipStream_t)stream, x, rows, *c, pi, tau, eps,
                       logw);
    return check_launch("causal_rows_kernel");
}

extern "C" int cmvae_gram_inverse(const float* logvar, float s0, int nmat, const cmvae_causal* c, float* out,
                                  void* stream) {
    cmvae::clear_stale_error();
    if (int e = check_causal(c)) return e;
    CMVAE_REQUIRE(out && nmat >= 0, "cmvae_gram_inverse: bad arguments");
    if (nmat == 0) return 0;
    return launch_gram_inverse(logvar, s0, nmat, *c, out, (hipStream_t)stream);
}

extern "C" size_t cmvae_unsup_prior_workspace_bytes(int B, int N, int K) {
    return carve(nullptr, B, N, 0, K, false).bytes;
}

// the loop of cmvae_unsup_prior over a carved workspace (shared with the entry point that allocates nothing)
static int unsup_prior_run(const float* z, int B, int N, int K, const int64_t* init_idx, const cmvae_causal* c,
                           int iters, int fix_pi, float gamma1, const EmWorkspace& w, float* pi, float* mu,
                           float* tau, float* logw, hipStream_t st) {
    const int tpb = pick_tiles_per_block(B, N, K), nblk = nblk_for(N, tpb);
    // initial inverse (tau0 = 1) and the loop inverse (tau = gamma1); both data independent
    if (int e = launch_gram_inverse(nullptr, 1.0f, 1, *c, w.minv0, st)) return e;
    const float* minv_loop = w.minv0;
    if (gamma1 != 1.0f) {
        if (int e = launch_gram_inverse(nullptr, gamma1, 1, *c, w.minv, st)) return e;
        minv_loop = w.minv;
    }
    FinArgs f{};
    f.z = z; f.init_idx = init_idx; f.N = N; f.K = K; f.fix_pi = fix_pi; f.semi = 0; f.init = 1;
    f.minv = w.minv0; f.minv_stride = 0; f.tau_new = 1.0f; f.c = *c;
    f.pi = pi; f.mu = mu; f.logvar = nullptr; f.tau = tau; f.logw = logw;
    hipLaunchKernelGGL(em_finalize_kernel, dim3(CMVAE_EM_SERIAL_FIN ? 1 : K, B), dim3(1024), 0, st, f);
    if (int e = check_launch("em_finalize_kernel(init)")) return e;
    PassArgs p{};
    p.z = z; p.mu = mu; p.logw = logw; p.part_N = w.part_N; p.part_S1 = w.part_S1;
    p.N = N; p.K = K; p.tiles_per_block = tpb; p.nblk = nblk;
    f.init = 0; f.part_N = w.part_N; f.part_S1 = w.part_S1; f.part_S2 = nullptr; f.nblk = nblk;
    f.minv = minv_loop; f.tau_new = gamma1;
    for (int it = 0; it < iters; ++it) {
        if (int e = launch_pass(p, B, st)) return e;
        hipLaunchKernelGGL(em_finalize_kernel, dim3(CMVAE_EM_SERIAL_FIN ? 1 : K, B), dim3(1024), 0, st, f);
        if (int e = check_launch("em_finalize_kernel")) return e;
    }
    return 0;
}

static int unsup_prior_check(const float* z, int B, int N, int K, const int64_t* init_idx, const cmvae_causal* c,
                             int iters, float* pi, float* mu, float* tau, float* logw) {
    if (int e = check_causal(c)) return e;
    CMVAE_REQUIRE(z && init_idx && pi && mu && tau && logw, "cmvae_unsup_prior: NULL tensor");
    CMVAE_REQUIRE(B > 0 && N > 0 && K >= 1 && K <= CMVAE_KMAX && iters >= 0, "cmvae_unsup_prior: bad shape B=%d N=%d K=%d",
                  B, N, K);
    CMVAE_REQUIRE(K <= N, "cmvae_unsup_prior: K=%d > N=%d", K, N);
    return 0;
}

extern "C" int cmvae_unsup_prior(const float* z, int B, int N, int K, const int64_t* init_idx, const cmvae_causal* c,
                                 int iters, int fix_pi, float gamma1, void* workspace, size_t workspace_bytes,
                                 float* pi, float* mu, float* tau, float* logw, void* stream) {
    cmvae::clear_stale_error();
    if (int e = unsup_prior_check(z, B, N, K, init_idx, c, iters, pi, mu, tau, logw)) return e;
    EmWorkspace w = carve(workspace, B, N, 0, K, false);
    CMVAE_REQUIRE(workspace && workspace_bytes >= w.bytes, "cmvae_unsup_prior: workspace %zu < %zu", workspace_bytes,
                  w.bytes);
    return unsup_prior_run(z, B, N, K, init_idx, c, iters, fix_pi, gamma1, w, pi, mu, tau, logw, (hipStream_t)stream);
}

// ---- the same loop for callers that cannot allocate (HIP graph capture, serving loops): the workspace is a
// file-scope device array.  Its name is a host-side stand-in, never a device address: every use below goes through
// the per-device pointer hipGetSymbolAddress returned (each GPU holds its own copy of the array).
static __device__ __align__(256) unsigned char scratch[CMVAE_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices];     // per device, filled on the first call on that device
std::mutex g_scratch_mutex;

int scratch_pointer(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return cmvae::set_error("cmvae_unsup_prior_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return cmvae::set_error("cmvae_unsup_prior_noalloc: device %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return cmvae::set_error("cmvae_unsup_prior_noalloc: hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace

extern "C" size_t cmvae_unsup_prior_noalloc_max_bytes(void) { return CMVAE_NOALLOC_SCRATCH_BYTES; }

extern "C" int cmvae_unsup_prior_noalloc(const float* z, int B, int N, int K, const int64_t* init_idx,
                                         const cmvae_causal* c, int iters, int fix_pi, float gamma1, float* pi,
                                         float* mu, float* tau, float* logw, void* stream) {
    cmvae::clear_stale_error();
    if (int e = unsup_prior_check(z, B, N, K, init_idx, c, iters, pi, mu, tau, logw)) return e;
    const size_t need = carve(nullptr, B, N, 0, K, false).bytes;
    CMVAE_REQUIRE(need <= (size_t)CMVAE_NOALLOC_SCRATCH_BYTES,
                  "cmvae_unsup_prior_noalloc: B=%d N=%d K=%d needs %zu bytes of scratch, the static buffer holds %zu "
                  "(use cmvae_unsup_prior with a caller workspace)", B, N, K, need, (size_t)CMVAE_NOALLOC_SCRATCH_BYTES);
    void* base = nullptr;
    if (int e = scratch_pointer(&base)) return e;
    // nothing in the workspace has to start at zero: every partial-sum slot that a finalize launch reads was written
    // by the pass launch just before it, and both inverses are written before their first use
    EmWorkspace w = carve(base, B, N, 0, K, false);
    return unsup_prior_run(z, B, N, K, init_idx, c, iters, fix_pi, gamma1, w, pi, mu, tau, logw, (hipStream_t)stream);
}

extern "C" int cmvae_em_pass(const float* z, const float* mu, const float* logw, int B, int N, int K, void* workspace,
                             size_t workspace_bytes, int* nblk_out, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(z && mu && logw, "cmvae_em_pass: NULL tensor");
    CMVAE_REQUIRE(B > 0 && N > 0 && K >= 1 && K <= CMVAE_KMAX, "cmvae_em_pass: bad shape");
    EmWorkspace w = carve(workspace, B, N, 0, K, false);
    CMVAE_REQUIRE(workspace && workspace_bytes >= w.bytes, "cmvae_em_pass: workspace %zu < %zu", workspace_bytes,
                  w.bytes);
    PassArgs p{};
    p.z = z; p.mu = mu; p.logw = logw; p.part_N = w.part_N; p.part_S1 = w.part_S1;
    p.N = N; p.K = K; p.tiles_per_block = pick_tiles_per_block(B, N, K); p.nblk = nblk_for(N, p.tiles_per_block);
    if (nblk_out) *nblk_out = p.nblk;
    return launch_pass(p, B, (hipStream_t)stream);
}

extern "C" size_t cmvae_semi_prior_workspace_bytes(int B, int Nu, int Ns, int K) {
    return carve(nullptr, B, Nu, Ns, K, true).bytes;
}

extern "C" int cmvae_semi_prior(const float* uz, const float* sz, const int64_t* y, int B, int Nu, int Ns, int K,
                                const cmvae_causal* c, int iters, float gamma1, void* workspace,
                                size_t workspace_bytes, float* pi, float* mu, float* logvar, float* tau, float* logw,
                                void* stream) {
    cmvae::clear_stale_error();
    if (int e = check_causal(c)) return e;
    CMVAE_REQUIRE(uz && sz && y && pi && mu && logvar && tau && logw, "cmvae_semi_prior: NULL tensor");
    CMVAE_REQUIRE(B > 0 && Nu > 0 && Ns > 0 && K >= 1 && K <= CMVAE_KMAX && iters >= 0,
                  "cmvae_semi_prior: bad shape B=%d Nu=%d Ns=%d K=%d", B, Nu, Ns, K);
    EmWorkspace w = carve(workspace, B, Nu, Ns, K, true);
    CMVAE_REQUIRE(workspace && workspace_bytes >= w.bytes, "cmvae_semi_prior: workspace %zu < %zu", workspace_bytes,
                  w.bytes);
    hipStream_t st = (hipStream_t)stream;
    const int tpb = pick_tiles_per_block(B, Nu, K), nblk = nblk_for(Nu, tpb);
    const int stpb = pick_tiles_per_block(B, Ns, K), snblk = nblk_for(Ns, stpb);
    // supervised sums (constant over the loop): y^T sz, y^T sz^2, sum y
    PassArgs s{};
    s.z = sz; s.labels = y; s.part_N = w.spart_N; s.part_S1 = w.spart_S1; s.part_S2 = w.spart_S2;
    s.N = Ns; s.K = K; s.tiles_per_block = stpb; s.nblk = snblk;
    if (int e = launch_pass(s, B, st)) return e;
    hipLaunchKernelGGL(sup_reduce_kernel, dim3(K, B), dim3(64), 0, st, w.spart_N, w.spart_S1, w.spart_S2, snblk, K,
                       w.sN, w.sS1, w.sS2);
    if (int e = check_launch("sup_reduce_kernel")) return e;
    if (int e = launch_gram_inverse(nullptr, 1.0f, 1, *c, w.minv0, st)) return e;
    FinArgs f{};
    f.K = K; f.fix_pi = 1; f.semi = 1; f.init = 1; f.sN = w.sN; f.sS1 = w.sS1; f.sS2 = w.sS2;
    f.minv = w.minv0; f.minv_stride = 0; f.tau_new = 1.0f; f.c = *c;
    f.pi = pi; f.mu = mu; f.logvar = logvar; f.tau = tau; f.logw = logw;
    hipLaunchKernelGGL(em_finalize_kernel, dim3(CMVAE_EM_SERIAL_FIN ? 1 : K, B), dim3(1024), 0, st, f);
    if (int e = check_launch("em_finalize_kernel(semi init)")) return e;
    PassArgs p{};
    p.z = uz; p.mu = mu; p.logw = logw; p.part_N = w.part_N; p.part_S1 = w.part_S1; p.part_S2 = w.part_S2;
    p.N = Nu; p.K = K; p.tiles_per_block = tpb; p.nblk = nblk;
    f.init = 0; f.part_N = w.part_N; f.part_S1 = w.part_S1; f.part_S2 = w.part_S2; f.nblk = nblk;
    f.minv = w.minv; f.minv_stride = (size_t)D * D; f.tau_new = gamma1;
    // the Gram inverses ride in the pass launch as extra grid rows, unless that would exceed the grid's y limit
    const int side_rows = (B * K + nblk - 1) / nblk;
    const bool fuse_inverse = B + side_rows <= 65535;
    if (fuse_inverse) { p.side_out = w.minv; p.side_nmat = B * K; p.side_rows = side_rows; }
    for (int it = 0; it < iters; ++it) {
        // inverse from the incoming logvar and tau = gamma1 (omniglot/model.py:259-265): C^T of every (episode,
        // component), then ONE launch for the Gram inverses and the E/M pass (they are independent of each other)
        if (int e = launch_causal_diag(logvar, gamma1, B * K, *c, w.minv, st)) return e;
        if (!fuse_inverse) {
            hipLaunchKernelGGL(gram_inverse_kernel, dim3(B * K), dim3(256), 0, st, w.minv);
            if (int e = check_launch("gram_inverse_kernel")) return e;
        }
        if (int e = launch_pass(p, B, st)) return e;
        hipLaunchKernelGGL(em_finalize_kernel, dim3(CMVAE_EM_SERIAL_FIN ? 1 : K, B), dim3(1024), 0, st, f);
        if (int e = check_launch("em_finalize_kernel")) return e;
    }
    return 0;
}

extern "C" int cmvae_posterior_argmax(const float* z, const float* mu, const float* logvar, const float* logw, int B,
                                      int te, int M, int K, int64_t* y_pred, float* post_mean, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(z && mu && logvar && logw && y_pred, "cmvae_posterior_argmax: NULL tensor");
    CMVAE_REQUIRE(B > 0 && te > 0 && M > 0 && K >= 1 && K <= CMVAE_KMAX, "cmvae_posterior_argmax: bad shape");
    const int ipb = M >= TR ? 1 : TR / M;
    const int grid = (te + ipb - 1) / ipb;
    return dispatch_kq(K, [&](auto kq) {
        constexpr int KQ = decltype(kq)::value;
        hipLaunchKernelGGL(posterior_argmax_kernel<KQ>, dim3(grid, B), dim3(256), 0, (hipStream_t)stream, z, mu, logvar,
                           logw, te, M, K, ipb, y_pred, post_mean);
        return check_launch("posterior_argmax_kernel");
    });
}

extern "C" size_t cmvae_mixture_workspace_bytes(int B, int N, int K) {
    const int tpb = pick_tiles_per_block(B, N, K), nblk = nblk_for(N, tpb);
    return align_up((size_t)B * nblk * 32 * sizeof(float));
}

extern "C" int cmvae_mixture_loglik_fwd(const float* z, const float* mu, const float* logw, int B, int N, int K,
                                        float* lse, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(z && mu && logw && lse, "cmvae_mixture_loglik_fwd: NULL tensor");
    CMVAE_REQUIRE(B > 0 && N > 0 && K >= 1 && K <= CMVAE_KMAX, "cmvae_mixture_loglik_fwd: bad shape");
    PassArgs p{};
    p.z = z; p.mu = mu; p.logw = logw; p.lse = lse; p.N = N; p.K = K;
    p.tiles_per_block = pick_tiles_per_block(B, N, K); p.nblk = nblk_for(N, p.tiles_per_block);
    return launch_pass(p, B, (hipStream_t)stream);
}

extern "C" int cmvae_mixture_loglik_bwd(const float* z, const float* mu, const float* logw, const float* g, int B,
                                        int N, int K, void* workspace, size_t workspace_bytes, float* dz, float* dlogw,
                                        void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(z && mu && logw && g && dz && dlogw, "cmvae_mixture_loglik_bwd: NULL tensor");
    CMVAE_REQUIRE(B > 0 && N > 0 && K >= 1 && K <= CMVAE_KMAX, "cmvae_mixture_loglik_bwd: bad shape");
    CMVAE_REQUIRE(workspace && workspace_bytes >= cmvae_mixture_workspace_bytes(B, N, K),
                  "cmvae_mixture_loglik_bwd: workspace too small");
    PassArgs p{};
    p.z = z; p.mu = mu; p.logw = logw; p.g = g; p.dz = dz; p.part_N = (float*)workspace; p.N = N; p.K = K;
    p.tiles_per_block = pick_tiles_per_block(B, N, K); p.nblk = nblk_for(N, p.tiles_per_block);
    if (int e = launch_pass(p, B, (hipStream_t)stream)) return e;
    hipLaunchKernelGGL(partn_reduce_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, p.nblk,
                       K, dlogw);
    return check_launch("partn_reduce_kernel");
}
