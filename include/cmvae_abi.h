/*
 * cmvae_abi.h — C-ABI of libcmvae_hip.so, the MI355X (gfx950) hot path of CMVAE's episodic
 * train / eval step.
 *
 * The reference (GuodongQi/CMVAE) has no FFI/plugin layer: its boundary is the Python
 * surface of GMVAE.  These entry points are what the Python host in
 * cmvae_amd/ binds through ctypes (cmvae_amd/_lib.py); each one cites the reference
 * function (file:line under /root/reference) whose arithmetic it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HIP, gfx950) unless a name ends in _host;
 *     tensors are dense row-major fp32, labels / indices int64.
 *   - the caller owns every buffer, including `workspace` (size from the matching
 *     *_workspace_bytes query); no entry point allocates, frees or synchronises.
 *   - kernels are enqueued on `stream` (a hipStream_t; NULL = default stream) and the
 *     call returns immediately.
 *   - return 0 on success, negative on error; cmvae_last_error() gives a thread-local
 *     message.  Shape limits: latent size D == 64, locally-connected width 32,
 *     1 <= K <= 32.
 *   - entry points keep no global state; they are re-entrant across host threads.
 */
#ifndef CMVAE_ABI_H
#define CMVAE_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMVAE_D 64        /* latent size the kernels are built for            */
#define CMVAE_M1 32       /* hidden units per latent dim in the NOTEARS-MLP   */
#define CMVAE_KMAX 32     /* max mixture components                           */

/* The causal layer (omniglot/module.py:136-196, mimgnet/module.py:47-110).
 * linear != 0 : mask(x) = x @ A,            W = A            [64,64]
 * linear == 0 : mask(x) = LC(ELU(fc1 x)),   W = fc1.weight   [2048,64], b1 = fc1.bias [2048],
 *               w2 = fc2.0.weight [64,32(,1)], b2 = fc2.0.bias [64(,1)]               */
typedef struct cmvae_causal {
    int32_t linear;
    const float* W;
    const float* b1;
    const float* w2;
    const float* b2;
} cmvae_causal;

const char* cmvae_last_error(void);
int cmvae_abi_version(void);

/* ---- causal layer on a small number of rows -------------------------------------------
 * eps[r,:] = x[r,:] - mask(x[r,:])                  CausalEncoder.forward, omniglot/module.py:165-188
 * Optionally also logw[r] = log(pi[r]) + glp(eps[r,:], log(tau[r]))  — the per-component prior
 * term of every E-step (omniglot/model.py:177-184).  eps / logw / pi / tau may be NULL.       */
int cmvae_causal_rows(const float* x, int rows, const cmvae_causal* c,
                      const float* pi, const float* tau,
                      float* eps, float* logw, void* stream);

/* ---- (I + C C^T)^-1, C row i = causal(s_i e_i) -----------------------------------------
 * replaces `std_causal = causal(diag(...)); inverse(I + std_causal @ std_causal^T)`
 * (omniglot/model.py:199-203, 259-265, 342-348).
 * scale: [nmat,64] per-row scales s (semi-EM: exp(0.5*logvar)*tau), or NULL with scalar `s0`
 * (unsupervised: tau).  logvar != NULL means s = exp(0.5*logvar[m,i]) * s0.
 * out: [nmat,64,64].                                                                         */
int cmvae_gram_inverse(const float* logvar, float s0, int nmat, const cmvae_causal* c,
                       float* out, void* stream);

/* ---- unsupervised EM prior: GMVAE.get_unsupervised_prior (omniglot/model.py:332-368)
 * with get_unsupervised_params (168-222) as the loop body.
 * z [B,N,64]; init_idx [B,K] int64 rows of z that seed the means (the reference draws them
 * with np.random.choice on the host, model.py:351-356 — the caller still does).
 * Outputs pi [B,K], mu [B,K,64], tau [B,K], logw [B,K] (= log pi + glp(causal(mu), log tau),
 * the term the next consumer adds to every log-likelihood).                                  */
size_t cmvae_unsup_prior_workspace_bytes(int B, int N, int K);
int cmvae_unsup_prior(const float* z, int B, int N, int K, const int64_t* init_idx,
                      const cmvae_causal* c, int iters, int fix_pi, float gamma1,
                      void* workspace, size_t workspace_bytes,
                      float* pi, float* mu, float* tau, float* logw, void* stream);

/* ---- cmvae_unsup_prior for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that may not call hipMalloc — code under HIP graph capture, tight serving loops, harnesses that forbid
 * allocation.  Same arguments, kernels and results as cmvae_unsup_prior, minus the workspace: the EM partial sums
 * and the two Gram inverses live in a file-scope device array of CMVAE_NOALLOC_SCRATCH_BYTES that every GPU the
 * library runs on holds until the process ends (its address is looked up once per device, on the first call on
 * that device — make that call outside graph capture).  Shapes whose workspace
 * (cmvae_unsup_prior_workspace_bytes) exceeds the array are rejected with an error; nothing is launched then.
 * All calls on one GPU share the array: calls that overlap on one GPU (two streams, two host threads) race.   */
#define CMVAE_NOALLOC_SCRATCH_BYTES (8u << 20)
size_t cmvae_unsup_prior_noalloc_max_bytes(void);
int cmvae_unsup_prior_noalloc(const float* z, int B, int N, int K, const int64_t* init_idx,
                              const cmvae_causal* c, int iters, int fix_pi, float gamma1,
                              float* pi, float* mu, float* tau, float* logw, void* stream);

/* ---- one fused E-step + M-step pass over z (the loop body's hot kernel, exposed for
 * measurement and tests): responsibilities from (mu, logw) and per-block partial sums
 * part_N [B,nblk,32], part_S1 [B,nblk,K,64] in `workspace` (layout of cmvae_unsup_prior).
 * Returns nblk through *nblk_out.  omniglot/model.py:173-193, 213-218.                     */
int cmvae_em_pass(const float* z, const float* mu, const float* logw, int B, int N, int K,
                  void* workspace, size_t workspace_bytes, int* nblk_out, void* stream);

/* ---- semi-supervised EM prior: GMVAE.get_semisupervised_prior (omniglot/model.py:370-412)
 * with get_semisupervised_params (224-330) as the loop body.
 * uz [B,Nu,64] unlabeled latents, sz [B,Ns,64] labeled latents, y [B,Ns] int64 labels in
 * [0,K) (the reference passes one_hot(y).float(); model.py:550-554).
 * Outputs pi [B,K] (uniform: never updated, model.py:330), mu, logvar [B,K,64], tau, logw.  */
size_t cmvae_semi_prior_workspace_bytes(int B, int Nu, int Ns, int K);
int cmvae_semi_prior(const float* uz, const float* sz, const int64_t* y,
                     int B, int Nu, int Ns, int K,
                     const cmvae_causal* c, int iters, float gamma1,
                     void* workspace, size_t workspace_bytes,
                     float* pi, float* mu, float* logvar, float* tau, float* logw, void* stream);

/* ---- final few-shot posterior: GMVAE.prediction tail (omniglot/model.py:564-583)
 * ll = glp(z, mu_k, logvar_k) + logw_k ; post = exp(ll - logsumexp) ; mean over the M
 * samples of each query item ; argmax over K.  z [B, te*M, 64]; y_pred [B,te] int64;
 * post_mean [B,te,K] optional (NULL to skip).                                               */
int cmvae_posterior_argmax(const float* z, const float* mu, const float* logvar, const float* logw,
                           int B, int te, int M, int K,
                           int64_t* y_pred, float* post_mean, void* stream);

/* ---- mixture log-likelihood with gradient (omniglot/model.py:475-494)
 * fwd: lse[b,n] = logsumexp_k( glp(z[b,n], mu[b,k]) + logw[b,k] )
 * bwd: dz[b,n,:] = g[b,n] * sum_k r_k (mu_k - z) ; dlogw[b,k] = sum_n g[b,n] r[b,n,k]
 * workspace: cmvae_mixture_workspace_bytes(B,N,K) (per-block partials of dlogw).           */
size_t cmvae_mixture_workspace_bytes(int B, int N, int K);
int cmvae_mixture_loglik_fwd(const float* z, const float* mu, const float* logw,
                             int B, int N, int K, float* lse, void* stream);
int cmvae_mixture_loglik_bwd(const float* z, const float* mu, const float* logw, const float* g,
                             int B, int N, int K, void* workspace, size_t workspace_bytes,
                             float* dz, float* dlogw, void* stream);

/* q_z - q_z.mean(1, keepdim=True) (omniglot/model.py:449, the input of the bulk causal layer):
 * x [B,N,64] -> out [B,N,64]; part [B][cmvae_center_rows_chunks(N)][64] is scratch.  The map is a symmetric
 * projection: its backward is the same call on the gradient.                                     */
int cmvae_center_rows_chunks(int N);
int cmvae_center_rows(const float* x, int B, int N, float* part, float* out, void* stream);

/* ---- reparametrisation + log q(z|x) (omniglot/model.py:111-122, 496-501)
 * z[r,m,:] = eps[r,m,:]*exp(0.5*logvar[r,:]) + mean[r,:] ; logq[r,m] = glp(z, mean, logvar)
 * rows = B*S, mean/logvar [rows,64], eps/z [rows,M,64], logq [rows,M] (may be NULL).
 * bwd: dmean[r,:] = sum_m gz ; dlogvar[r,:] = sum_m gz*0.5*eps*std - 0.5*sum_m glogq
 *      (the analytic gradient of the two reference expressions).                           */
int cmvae_reparam_logq_fwd(const float* mean, const float* logvar, const float* eps,
                           int rows, int M, float* z, float* logq, void* stream);
int cmvae_reparam_logq_bwd(const float* logvar, const float* eps, const float* gz, const float* glogq,
                           int rows, int M, float* dmean, float* dlogvar, void* stream);

/* ---- acyclicity: h = tr(expm(A∘A)) - 64 and G = dh/dA = expm(A∘A)^T ∘ 2A
 * (GMVAE.h_a, omniglot/model.py:159-166).  A [64,64]; h [1]; grad [64,64] (may be NULL). */
int cmvae_expm_trace(const float* A, float* h, float* grad, void* stream);

/* ==== scalar tail of GMVAE.forward (omniglot/model.py:496-516) in two launches (backward: one).
 * out = {kl, h_aug}:  kl = mean(log_q) - mean(log_p) - mean_rows(sum_d(-0.5 q_eps^2 + c)),  c = -0.5 log(2 pi);
 *                     h_aug = h_lambda * h + 0.5 * h_c * h^2 + a_lambda * sum|A|.
 * log_q [n1], log_p [n2], q_eps [rows, dim] (dim % 4 == 0), h: device scalar (cmvae_expm_trace), A [nA];
 * part: [cmvae_elbo_tail_blocks()][4] scratch.
 * Backward from g = {d kl, d h_aug} (device): d_log_q = g0/n1, d_log_p = -g0/n2, d_q_eps = g0 * q_eps / rows,
 * dA = g1 * a_lambda * sign(A) (the |A| term only), dh = g1 * (h_lambda + h_c * h).                              */
int cmvae_elbo_tail_blocks(void);
int cmvae_elbo_tail_fwd(const float* log_q, int64_t n1, const float* log_p, int64_t n2, const float* q_eps,
                        int64_t rows, int dim, const float* h, const float* A, int64_t nA, float h_lambda, float h_c,
                        float a_lambda, float* part, float* out, void* stream);
int cmvae_elbo_tail_bwd(const float* g, const float* h, const float* q_eps, int64_t rows, int dim, const float* A,
                        int64_t nA, int64_t n1, int64_t n2, float h_lambda, float h_c, float a_lambda,
                        float* d_log_q, float* d_log_p, float* d_q_eps, float* dA, float* dh, void* stream);
/* Prior log-weights of the mixture term (omniglot/model.py:481-491; log(trace_k) sits in gaussian_log_prob's MEAN
 * slot there, and here): logw[r] = log(pi[r]) + sum_d(-0.5 * (x[r][d] - lt)^2 + c), lt = log(trace[r]);
 * x [rows, 64] = the causal layer's image of mu_k.  Backward: dx[r][d] = -g[r] * (x[r][d] - lt)  (pi and trace are
 * EM outputs, constants).                                                                                       */
int cmvae_prior_logw_fwd(const float* pi, const float* trace, const float* x, int rows, float* logw, void* stream);
int cmvae_prior_logw_bwd(const float* trace, const float* x, const float* g, int rows, float* dx, void* stream);

/* ==== dense contractions: fp32-MFMA gather-GEMM ========================================
 * Activations are row-major [pixels, C] (channels-last).  One geometry describes conv 3x3
 * (omniglot/module.py:14-29), each sub-pixel phase of the stride-2 transposed convs
 * (omniglot/module.py:50-62), their data gradients, and nn.Linear (omniglot/model.py:65-76,
 * module.py:80-86,148):
 *   out[outpix(r), nb*64 + 0:64] = act( sum_t T(A[src(r,t), a_col[t] : +64]) @ W_t + bias )
 * r enumerates (image, y, x) over the per-image grid [Hg,Wg]; src(r,t) = (image, y*sy+dy[t],
 * x*sx+dx[t]) in the [Hi,Wi] input image (zero outside); outpix(r) = (image, y*oys+oy0,
 * x*oxs+ox0) in the [Ho,Wo] output image.  W_t is stored [out col][k] (k contiguous):
 * W + nb*w_nblk_stride + w_off[t] + col*w_pitch + k  (per-tap offsets: a sub-pixel phase picks its taps out of
 * the full [k*k][64][64] kernel without a gather).  T = relu(a*a_scale[c]+a_shift[c])
 * when a_scale != NULL (the previous BatchNorm+ReLU folded into the operand load).          */
typedef struct cmvae_conv_geom {
    int32_t rows_total, Hg, Wg, Hi, Wi, sy, sx, Ho, Wo, oys, oxs, oy0, ox0, ntaps;
    int32_t a_pitch, o_pitch, w_pitch;
    int64_t w_nblk_stride;
    int32_t dy[32], dx[32], a_col[32], w_off[32];
    /* != 0: rows enumerate (y, x, image) — r = (y*Wg + x)*n_images + image — instead of (image, y, x).  Results are the
     * same sums; a tile of rows then sits at ONE grid position, so taps outside the image are skipped for the whole
     * tile instead of multiplying zero padding (the stride-2 transposed convs of the decoder and their gradients).
     * act must be 0 for such launches.                                                                              */
    int32_t pixel_major;
} cmvae_conv_geom;

/* act: 0 none, 1 ReLU, 2 ELU.  stats_part (optional, n_col_blocks == 1): per-block partial
 * column sums of the pre-activation output, [cmvae_conv_gemm_grid(rows,1)][128] = {sum[64],
 * sum of squares[64]} — the BatchNorm statistics of the layer, reduced in a fixed order.      */
int cmvae_conv_gemm_grid(int rows_total, int n_col_blocks);
int cmvae_conv_gemm(const cmvae_conv_geom* geom, const float* A, const float* Wt, const float* bias,
                    const float* a_scale, const float* a_shift, int act, int n_col_blocks,
                    float* out, float* stats_part, void* stream);
/* Data gradient of a decoder layer with the BatchNorm-backward sums of the layer below from the epilogue
 * (backward of ConvTranspose2d -> BatchNorm2d -> ReLU, omniglot/module.py:50-58): out = conv_gemm(geom, A, Wt) is the
 * gradient at a = relu(bn(x)); x [pixels,64] is read at out's own pixels and
 * sums_part [cmvae_conv_gemm_grid(rows,1)][128] receives per-workgroup partials of
 * {sum g[64], sum g*xhat[64]}, g = out * (x*scale + shift > 0), xhat = (x - mean) * invstd
 * (= cmvae_bn_relu_bwd_reduce(out, x, ...) without its pass over out and x).  No bias, no activation.          */
int cmvae_conv_gemm_bnbwd(const cmvae_conv_geom* geom, const float* A, const float* Wt, float* out, const float* x,
                          const float* scale, const float* shift, const float* mean, const float* invstd,
                          float* sums_part, void* stream);
/* Up to four geometries over the same operand, weight and output tensors in one launch: the sub-pixel phases of a
 * ConvTranspose2d(64,64,k,2,1) (omniglot/module.py:50-58), each with at most 4 taps, identity activation, one
 * 64-column block, all in the same row order.  stats_part (or NULL): [cmvae_conv_gemm_multi_grid()][128] partial
 * column sums {sum, sum of squares} over ALL geometries' rows.                                                  */
int cmvae_conv_gemm_multi_grid(void);
/* Rows from which cmvae_conv_gemm (7 .. 9 taps, act 0, one column block), the pixel-major launches of
 * cmvae_conv_gemm_multi and cmvae_conv_gemm_bnbwd (<= 18 taps; more than 6 as accumulating passes, cut at the taps'
 * stride-2 parity classes) and cmvae_conv_wgrad_multi (four-tap phases, image count a multiple of 32) run their fp32
 * products on the bf16 matrix pipe (cmvae_linear_fwd / cmvae_linear_bwd: launches of >= 16384 rows, see there): every
 * operand split exactly into three bf16 pieces, six v_mfma_*_bf16 products per fp32 product, fp32 accumulation — the
 * results are fp32 results (tests/test_gpu_conv.py: test_bf16_pipe_products_are_fp32_products).  No counterpart in the
 * reference (omniglot/module.py:50-58 calls torch's conv_transpose2d); a measurement aid for bench.py and tests.  */
int cmvae_conv_bf16_pipe_min_rows(void);
int cmvae_conv_gemm_multi(const cmvae_conv_geom* geoms, int ngeom, const float* A, const float* Wt, const float* bias,
                          const float* a_scale, const float* a_shift, float* out, float* stats_part, void* stream);

/* Weight gradient of the same contraction: dW[tap_off_host[t] + ci*s_ci + co*s_co] =
 * sum_r T(A[src(r,t)])[ci] * G[outpix(r), g_col + co];  db[co] (+)= sum_r G[outpix(r), g_col+co].
 * tap_off_host is a HOST array of geom->ntaps offsets.                                        */
size_t cmvae_conv_wgrad_workspace_bytes(int rows_total);
int cmvae_conv_wgrad(const cmvae_conv_geom* geom, const float* A, const float* a_scale, const float* a_shift,
                     const float* G, int g_pitch, int g_col, const int64_t* tap_off_host,
                     int64_t s_ci, int64_t s_co, float* dW, float* db, int accumulate_db,
                     void* workspace, size_t workspace_bytes, void* stream);
/* Weight and bias gradient of up to four geometries over the same tensors (the sub-pixel phases of a
 * ConvTranspose2d, omniglot/module.py:50-58; at most 4 taps each) in one launch + one reduction.  tap_off_host: the
 * geometries' taps one after the other; db (or NULL) receives the bias gradient over ALL geometries' rows (not
 * accumulated).  Image counts the streaming kernel does not take (not a multiple of 16) run one launch per geometry.
 * Workspace as for cmvae_conv_wgrad.                                                                            */
int cmvae_conv_wgrad_multi(const cmvae_conv_geom* geoms, int ngeom, const float* A, const float* a_scale,
                           const float* a_shift, const float* G, int g_pitch, const int64_t* tap_off_host,
                           int64_t s_ci, int64_t s_co, float* dW, float* db, void* workspace, size_t workspace_bytes,
                           void* stream);

/* ==== nn.Linear (+ReLU/ELU) on the gather-GEMM kernels: proj (omniglot/model.py:70-77, mimgnet/model.py:59-66),
 * the Set-Transformer projections fc_q/k/v/o and the (mu, logvar) head (omniglot/module.py:80-86, model.py:65-68),
 * CausalEncoder.fc1 (omniglot/module.py:148,181).  W [out_f,in_f] as torch stores it; features are multiples of
 * 64, at most 2048.  Y = act(X W^T + bias); act: 0 none, 1 ReLU, 2 ELU.
 * Launches of >= 16384 rows whose contraction has at most 384 columns run their fp32 products on the bf16 matrix
 * pipe (see cmvae_conv_bf16_pipe_min_rows); a data gradient with a longer contraction over few column blocks (fc1:
 * 2048 -> 64) is split over the contraction into tap groups whose partial products (in the workspace) are added in
 * fixed order.  Results are fp32 results and reproducible run to run.                                          */
int cmvae_linear_fwd(const float* X, int x_pitch, int rows, int in_f, int out_f, const float* W,
                     const float* bias, int act, float* Y, int y_pitch, void* stream);
/* The last proj layer of the feature models fused with their Gaussian likelihood (mimgnet/model.py:59-66, 388-391):
 * Y = act(X W^T + bias) as cmvae_linear_fwd, and the squared error of Y against target[row / M, :] (target
 * [rows / M, out_f]: the features every MC sample reconstructs) summed into loss_part
 * [cmvae_linear_mse_partials(rows, out_f)][128] — F.mse_loss(reduction='sum') is the sum of all its entries.
 * Backward of that sum through the activation: dZ = act'(.) * gmul * g_dev[0] * 2 (Y - target) (g_dev: device scalar,
 * the upstream gradient of the loss sum, or NULL for 1); dZ then feeds cmvae_linear_bwd.                        */
int cmvae_linear_mse_partials(int rows, int out_f);
int cmvae_linear_mse_fwd(const float* X, int x_pitch, int rows, int in_f, int out_f, const float* W,
                         const float* bias, int act, const float* target, int M, float* Y, int y_pitch,
                         float* loss_part, void* stream);
int cmvae_mse_act_bwd(const float* Y, const float* target, int rows, int F, int M, int act,
                      const float* g_dev, float gmul, float* dZ, void* stream);
/* dZ = dY * act'(.) from the saved output Y (n elements, n % 4 == 0; act 1 or 2).                         */
int cmvae_act_bwd(const float* dY, const float* Y, int64_t n, int act, float* dZ, void* stream);
/* dX = dZ W (skipped when dX == NULL), dW[out_f,in_f] = dZ^T X, db[out_f] = column sums of dZ (may be NULL). */
size_t cmvae_linear_bwd_workspace_bytes(int rows, int in_f, int out_f);
int cmvae_linear_bwd(const float* X, int x_pitch, const float* W, const float* dZ, int dz_pitch, int rows,
                     int in_f, int out_f, float* dX, int dx_pitch, float* dW, float* db,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ==== softmax attention core of the Set-Transformer MAB (omniglot/module.py:92-99), heads of 64 channels.
 * Q, K, V, O, dO, dQ, dK, dV: [B, S, heads*64]; O = Q + softmax(scale * Q K^T) V per (episode, head);
 * lse, delta: [B, heads, S] (log-sum-exp of the scaled scores, saved for the backward; scratch).  S <= 1024. */
int cmvae_attention_fwd(const float* Q, const float* K, const float* V, int B, int S, int heads, float scale,
                        float* O, float* lse, void* stream);
int cmvae_attention_bwd(const float* Q, const float* K, const float* V, const float* lse, const float* dO,
                        int B, int S, int heads, float scale, float* dQ, float* dK, float* dV, float* delta,
                        void* stream);
/* The same over a packed projection QKV [B, S, 3*heads*64], row = [q | k | v] — the self-attention blocks (SAB,
 * omniglot/module.py:102-108: MAB(X, X)), whose fc_q, fc_k, fc_v read the same rows and so run as ONE GEMM with the
 * three weights stacked.  dQKV has the layout of QKV and feeds that GEMM's backward.                          */
int cmvae_attention_qkv_fwd(const float* QKV, int B, int S, int heads, float scale, float* O, float* lse,
                            void* stream);
int cmvae_attention_qkv_bwd(const float* QKV, const float* lse, const float* dO, int B, int S, int heads,
                            float scale, float* dQKV, float* delta, void* stream);

/* ==== NOTEARS-MLP causal layer on bulk rows (CausalEncoder.forward on [B*N,64], omniglot/module.py:165-188,
 * LocallyConnected :232-288).  hid [rows,2048] = elu(fc1(x)) comes from cmvae_linear_fwd(act=2);
 * eps[r][j] = x[r][j] - (sum_m hid[r][32j+m] w2[j][m] + b2[j]).
 * Backward: d_pre [rows,2048] = gradient at fc1's pre-activation (feeds cmvae_linear_bwd) and
 * part [cmvae_causal_lc_bwd_grid(rows)][cmvae_causal_lc_bwd_partial_len()] = {dw2[64][32], db2[64], db1[2048]} per block
 * (reduce with cmvae_colsum_reduce).                                                                        */
int cmvae_causal_lc_fwd(const float* x, const float* hid, const float* w2, const float* b2, int rows,
                        float* eps, void* stream);
int cmvae_causal_lc_bwd_grid(int rows);
int cmvae_causal_lc_bwd_partial_len(void);
int cmvae_causal_lc_bwd(const float* d_eps, const float* hid, const float* w2, int rows, float* d_pre,
                        float* part, void* stream);

/* ==== Adam with L2 weight decay over one flat parameter buffer (torch.optim.Adam as configured in
 * omniglot/trainer.py:48-52); step counts from 1; n % 4 == 0.                                               */
int cmvae_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                    float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);
/* The same update with the step count (>= 1) read from device memory at run time: a launch that is replayed from a
 * HIP graph (the caller increments *step_dev before each replay).                                              */
int cmvae_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                        const int64_t* step_dev, float lr, float beta1, float beta2, float eps, float weight_decay,
                        void* stream);

/* ==== BatchNorm (training statistics over the whole meta-batch) on channels-last [rows,64]
 * nn.BatchNorm2d after every conv (omniglot/module.py:15,20,25,30,51,55,59).  The conv GEMM
 * emits per-block partial sums; these finish the job.                                        */
int cmvae_colsum_reduce(const float* part, int nblk, int L, float* out, void* stream);   /* out[L] = sum_b part[b][L] */
/* sums = {sum x[64], sum x^2[64]} over `count` values per channel (all-reduce them first when the
 * episodes are sharded).  Writes scale = gamma*invstd, shift = beta - mean*scale, mean, invstd and
 * updates running_mean / running_var (momentum, unbiased variance) when they are not NULL.     */
int cmvae_bn_finalize(const float* sums, double count, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_var,
                      float* scale, float* shift, float* mean, float* invstd, void* stream);
/* The same straight from the partials [nblk][128] of the conv GEMM's epilogue, in one launch (a single process has
 * no all-reduce between the column sums and the finalize); num_batches_tracked (int64 on the device, or NULL) is
 * incremented like nn.BatchNorm2d does.                                                                        */
int cmvae_bn_finalize_part(const float* part, int nblk, double count, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           float* scale, float* shift, float* mean, float* invstd, void* stream);
/* model.eval(): the same fold from the running statistics (nn.BatchNorm2d in eval mode, used by
 * GMVAE.prediction, omniglot/model.py:520-585): scale = gamma*rsqrt(running_var+eps),
 * shift = beta - running_mean*scale, invstd.                                                   */
int cmvae_bn_eval_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, float* scale, float* shift, float* invstd, void* stream);
/* backward of relu(bn(y)): dybn = d_a*(bn(y)>0) (masked != 0: d_a is already masked);
 * part [cmvae_rowpass_grid(rows)][128] = {sum dybn, sum dybn*xhat}; then
 * dy = scale*(dybn - c1 - xhat*c2) with c1, c2 from cmvae_bn_bwd_finalize.                     */
int cmvae_rowpass_grid(int64_t rows);
int cmvae_bn_relu_bwd_reduce(const float* d_a, const float* y, const float* scale, const float* shift,
                             const float* mean, const float* invstd, int64_t rows, int masked,
                             float* part, void* stream);
int cmvae_bn_bwd_finalize(const float* sums_local, const float* sums_global, double count,
                          float* c1, float* c2, float* dgamma, float* dbeta, void* stream);
/* ... from the partials [nblk][128] of one process (no all-reduce), in one launch.                              */
int cmvae_bn_bwd_finalize_part(const float* part, int nblk, double count, float* c1, float* c2, float* dgamma,
                               float* dbeta, void* stream);
int cmvae_bn_relu_bwd_apply(const float* d_a, const float* y, const float* scale, const float* shift,
                            const float* mean, const float* invstd, const float* c1, const float* c2,
                            int64_t rows, int masked, float* dy, void* stream);

/* ==== last decoder layer fused with the Bernoulli likelihood
 * relu(bn3(y3)) -> ConvTranspose2d(64,1,4,2,1) -> Sigmoid -> BCELoss(sum)
 * (omniglot/module.py:62-63, omniglot/model.py:83,463-466).  y3 [nimg,14,14,64]; w4 [64,16];
 * target [nimg/M, 784] (the image each MC sample reconstructs); p_out [nimg,784];
 * loss_part [nimg] (per-image BCE sums, log clamped at -100 as nn.BCELoss does).
 * bwd: dybn [nimg,14,14,64] = masked gradient w.r.t. bn3 output (NULL: not stored, sums only); part
 * [cmvae_dec_out_grid(nimg)][cmvae_dec_out_bwd_partial_len()] = {dW4[64][16], sum dybn[64],
 * sum dybn*xhat[64], db4}.  Upstream gradient of the BCE sum = gmul * g_dev[0] (g_dev: device
 * scalar or NULL for 1).                                                                       */
int cmvae_dec_out_grid(int nimg);
int cmvae_dec_out_bwd_partial_len(void);
int cmvae_dec_out_fwd(const float* y3, const float* scale, const float* shift, const float* w4, const float* b4,
                      const float* target, int M, int nimg, float* p_out, float* loss_part, void* stream);
int cmvae_dec_out_bwd(const float* y3, const float* scale, const float* shift, const float* mean,
                      const float* invstd, const float* w4, const float* p, const float* target, int M, int nimg,
                      const float* g_dev, float gmul, float* dybn, float* part, void* stream);
/* The sums pass of the last layer's backward by linearity (no gradient tile): two contractions over the pixels,
 * R = sum m*xhat*D and P = sum m*D, give dW4 = gamma R + beta P, sum dybn3 = sum_tap w4 P, sum dybn3*xhat = sum_tap w4 R.
 * out [1153] has the layout of cmvae_dec_out_bwd's column-summed partials { dW4[64][16], sum dybn3[64],
 * sum dybn3*xhat[64], db4 }; part: [cmvae_dec_out_bwd_sums_grid(nimg)][cmvae_dec_out_bwd_sums_partial_len()] and
 * tot: [partial_len] are scratch.  Same reference lines as cmvae_dec_out_bwd.                                  */
int cmvae_dec_out_bwd_sums_partial_len(void);
int cmvae_dec_out_bwd_sums_grid(int nimg);
int cmvae_dec_out_bwd_sums(const float* y3, const float* scale, const float* shift, const float* mean,
                           const float* invstd, const float* w4, const float* p, const float* target, int M, int nimg,
                           const float* g_dev, float gmul, float* part, float* tot, float* out, void* stream);
/* Second pass of the same backward: with c1, c2 from cmvae_bn_bwd_finalize (computed from the sums of the first
 * pass, called with dybn == NULL) it recomputes the masked gradient and writes the gradient at layer 3's
 * pre-BatchNorm output, dy3 = scale*(dybn3 - c1 - xhat*c2), so dybn3 never travels through HBM.               */
int cmvae_dec_out_bwd_apply(const float* y3, const float* scale, const float* shift, const float* mean,
                            const float* invstd, const float* w4, const float* p, const float* target,
                            int M, int nimg, const float* g_dev, float gmul, const float* c1, const float* c2,
                            float* dy, void* stream);

/* ==== OmniglotEncoder pieces that are not 64->64 GEMMs (omniglot/module.py:12-33)
 * Conv2d(1,64,3,1,1): x [n,H,W] -> y [n*H*W,64]; w [64,9]; stats_part (optional)
 * [cmvae_rowpass_grid(n*H*W)][128] (y may be NULL: statistics only).
 * pool: relu(bn(y)) then MaxPool2d(2, ceil_mode=True): y [n,H,W,64] -> out [n,ceil(H/2),ceil(W/2),64];
 * bwd: d_pool -> dybn [n,H,W,64] (first maximum takes the gradient, ReLU mask applied) and
 * part [cmvae_rowpass_grid(n*Ho*Wo)][128] = {sum dybn, sum dybn*xhat}.                            */
int cmvae_enc_conv1_fwd(const float* x, const float* w, const float* b, int n, int H, int W,
                        float* y, float* stats_part, void* stream);
/* Layer 1's whole backward (autograd of omniglot/module.py:14-17 for the first conv block) in one pass: the
 * weight gradient is linear in dybn and xhat, so the pooling backward accumulates per channel s1 = sum dybn,
 * s2 = sum dybn*xhat, P1[t] = sum dybn*(x_t - m_t) and never writes dybn; m_t = mean of tap t of the images (zero
 * padding included).  cmvae_enc_conv1_tap_sums: part_m [256][64] = partial {sum x_t [9], sum x_t'*x_t [45, t' <= t]}.
 * part [cmvae_enc_conv1_bwd_lin_rc_grid(n)][cmvae_enc_conv1_bwd_lin_cols()]: {s1, s2}[64], P1 [64][9].
 * cmvae_enc_conv1_lin_finalize: tot = column sums of part (this rank), sums_glob [128] = {s1, s2} summed over the
 * ranks (tot itself in one process), count_glob = values per channel over all ranks, w [64][9], b [64] = conv1;
 * sum xhat*(x_t - m_t) = invstd_c * sum_t' w[c][t'] Gc[t'][t] (Gc: Gram matrix of the centred taps) and
 * s3 = sum xhat = invstd_c * N * (b_c + sum_t' w[c][t'] m_t' - mean_c) follow from y = conv1(x):
 *   dW[c][t] = scale_c * (P1 + m_t*(s1 - c1*N) - c2*(P3 + m_t*s3)),  db = scale_c * (s1 - c1*N - c2*s3),
 *   dgamma = s2, dbeta = s1,  c1 = sums_glob[c]/count_glob, c2 = sums_glob[64+c]/count_glob, N = n*H*W.            */
int cmvae_enc_conv1_bwd_lin_cols(void);
int cmvae_enc_conv1_tap_sums(const float* x, int n, int H, int W, float* part_m, void* stream);
/* The pass itself, with y recomputed from the images (the arithmetic of cmvae_enc_conv1_fwd, bit for bit): the
 * training step then never stores layer 1's pre-pooling activation (cmvae_enc_conv1_fwd with y = NULL for the batch
 * statistics, cmvae_enc_conv1_pool_fwd with the batch fold for the pooled output).
 * part [cmvae_enc_conv1_bwd_lin_rc_grid(n)][cmvae_enc_conv1_bwd_lin_cols()].                                       */
int cmvae_enc_conv1_bwd_lin_rc_grid(int n);
int cmvae_enc_conv1_bwd_lin_rc(const float* d_pool, const float* x, const float* w, const float* b, const float* scale,
                               const float* shift, const float* mean, const float* invstd, const float* part_m, int n,
                               int H, int W, float* part, void* stream);
int cmvae_enc_conv1_lin_finalize(const float* tot, const float* sums_glob, float count_glob, int n, int H, int W,
                                 const float* part_m, const float* w, const float* b, const float* scale,
                                 const float* mean, const float* invstd, float* dW, float* db, float* dgamma,
                                 float* dbeta, void* stream);
/* model.eval() (GMVAE.prediction, omniglot/model.py:520-531 -> module.py:14-17): Conv2d(1,64,3,1,1), BatchNorm from
 * the running statistics (scale/shift of cmvae_bn_eval_fold), ReLU and MaxPool2d(2, ceil_mode=True) in one pass:
 * x [n,H,W] -> out [n,ceil(H/2),ceil(W/2),64]; the pre-pooling activation is never written.  H, W <= 32.      */
int cmvae_enc_conv1_pool_fwd(const float* x, const float* w, const float* b, const float* scale, const float* shift,
                             int n, int H, int W, float* out, void* stream);
int cmvae_pool_fwd(const float* y, const float* scale, const float* shift, int n, int H, int W,
                   float* out, void* stream);
int cmvae_pool_bwd(const float* d_pool, const float* y, const float* scale, const float* shift,
                   const float* mean, const float* invstd, int n, int H, int W,
                   float* dybn, float* part, void* stream);

/* ==== EM seed stream (HOST function: every pointer is host memory, nothing is launched) ==================
 * The reference seeds the unsupervised EM with `np.random.choice(N, K, replace=False)` once per episode, in episode
 * order, from numpy's global generator (omniglot/model.py:351-356; mimgnet/model.py likewise).  This draws the
 * same indices for `n_episodes` episodes from the same generator state — numpy's legacy algorithm restated
 * (permutation(N)[:K]: Fisher-Yates from the top with masked-rejection 32-bit MT19937 draws) — and advances the
 * state exactly as numpy would, ~5x faster than the interpreter path and without the interpreter lock, so the
 * O(world size) draw of a sharded job can run on a helper thread (cmvae_amd/dist.py: SeedStream).
 * mt_key_host [624] / *mt_pos_host: the MT19937 state as np.random.get_state() returns it, updated in place;
 * out_host [n_episodes, K] int64; scratch_host [N] int64.                                                       */
int cmvae_em_seed_draw_host(uint32_t* mt_key_host, int* mt_pos_host, int n_samples, int k, int n_episodes,
                            int64_t* out_host, int64_t* scratch_host);

#ifdef __cplusplus
}
#endif
#endif /* CMVAE_ABI_H */
