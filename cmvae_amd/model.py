"""GMVAE — the episodic train (`forward`) and few-shot eval (`prediction`) hot path of CMVAE on
MI355X.  Same constructor arguments, return values and ``state_dict`` keys as the reference
(omniglot/model.py:14-87, 431-585; mimgnet/model.py:11-69, 364-511; celeba/model.py).

Where the arithmetic runs:
  * EM loops, causal prior term, (I+CC^T)^-1, mixture log-likelihood (fwd+bwd),
    reparametrisation + log q (fwd+bwd), acyclicity expm trace (fwd+bwd), few-shot posterior
    + argmax: hand-written HIP kernels behind the C-ABI (cmvae_amd/ops.py).
  * conv / conv-transpose / BatchNorm / max-pool / BCE (cmvae_amd/encoder.py, decoder.py), every
    nn.Linear (+ELU/ReLU), the bulk causal layer and Adam (cmvae_amd/linear.py): hand-written HIP
    kernels on the fp32-MFMA gather-GEMM family.
  * the 4-head softmax attention core of the two SABs: MFMA kernels (cmvae_amd/csrc/attention.hip).
  * Still ATen on the GPU: tensor plumbing (cat, index gathers, weight layout permutations) and the final
    `rec + kl + h_augment` add; the scalar tail of `forward` is two HIP kernels (ops.prior_logw, ops.elbo_tail).
There is no CPU path: tensors must live on the GPU and the HIP library must be loadable.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .module import OmniglotEncoder, OmniglotDecoder, SAB, CausalEncoder
from .linear import linear, linear_mse
from .decoder import decoder_bce, nhwc_perm, permute

_LOG_NORM = -0.5 * np.log(2 * np.pi)

_SIDE_STREAMS = {}


def _side_stream(device):
    """One extra HIP stream per device for work issued beside the current stream (`_unsup_prior_begin`)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)      # (a high-priority stream was measured: the gain goes from 0.14 to 0.03 ms)
    return _SIDE_STREAMS[key]


class _GMVAEBase(nn.Module):
    # the training forward issues the prior's EM on a second stream beside the decoder (_unsup_prior_begin)
    em_side_stream = True

    def _init_common(self, unsupervised_em_iters, semisupervised_em_iters, fix_pi, component_size, latent_size,
                     train_mc_sample_size, test_mc_sample_size, linear, laplace, gamma):
        if latent_size != ops.D:
            raise ValueError('the MI355X kernels are built for latent_size == 64 (got %d)' % latent_size)
        if laplace:
            raise NotImplementedError('laplace=True is never used by the reference trainers')
        self.gamma1 = 1 / math.sqrt(gamma)
        self.laplace = laplace
        self.linear = linear
        self.unsupervised_em_iters = unsupervised_em_iters
        self.semisupervised_em_iters = semisupervised_em_iters
        self.fix_pi = fix_pi
        self.component_size = component_size
        self.latent_size = latent_size
        self.train_mc_sample_size = train_mc_sample_size
        self.test_mc_sample_size = test_mc_sample_size

    def _register_constants(self, with_log2):
        self.register_buffer('log_norm_constant', torch.tensor(_LOG_NORM))       # float64 0-dim, as the reference
        if with_log2:
            self.register_buffer('log_2', torch.tensor(np.log(2)))
        self.register_buffer('uniform_pi', torch.ones(self.component_size) / self.component_size)

    def tail_param_count(self):
        """Number of trailing parameters (module order) whose gradients are complete first in the backward pass —
        proj and, for Omniglot, the decoder: the bucket `dist.FlatGrads.overlap_tail` sends early."""
        n = len(list(self.proj.parameters()))
        if hasattr(self, 'decoder'):
            n += len(list(self.decoder.parameters()))
        return n

    # ------------------------------------------------------------------ small pieces
    def _causal_desc(self):
        c = self.causal
        if c.linear:
            return ops.Causal(True, c.A)
        return ops.Causal(False, c.fc1.weight, c.fc1.bias, c.fc2[0].weight, c.fc2[0].bias)

    def gaussian_log_prob(self, x, mean, logvar=None):
        if logvar is None:
            # the reference builds a zeros tensor and runs the general formula (omniglot/model.py:89-95); with
            # logvar = 0 that is 0 + a/1 = a bit for bit, so the four extra elementwise passes are skipped
            return (-0.5 * (x - mean).pow(2) + self.log_norm_constant).sum(dim=-1)
        return (-0.5 * (logvar + (x - mean).pow(2) / logvar.exp()) + self.log_norm_constant).sum(dim=-1)

    def h_a(self, A):
        return ops.expm_trace(A)

    def _draw_eps(self, mean, S):
        # same generator call and shape as the reference's torch.randn_like on the repeated mean
        return torch.randn(mean.shape[0], mean.shape[1], S, mean.shape[2], device=mean.device, dtype=mean.dtype)

    def _draw_init_idx(self, batch_size, sample_size, device):
        # host numpy stream, one draw per episode in episode order (omniglot/model.py:351-356)
        idx = np.stack([np.random.choice(sample_size, self.component_size, replace=False)
                        for _ in range(batch_size)], 0).astype(np.int64)
        return torch.from_numpy(idx).to(device)

    def _posterior_head(self, H):
        """q_z_given_H: two SABs and the (mu, logvar) Linear (omniglot/model.py:62-68)."""
        H = self.q_z_given_H[1](self.q_z_given_H[0](H))
        return linear(H, self.q_z_given_H[2].weight, self.q_z_given_H[2].bias)

    def get_unsupervised_prior(self, z, init_idx=None):
        """(pi_k, mu_k, trace_k) after `unsupervised_em_iters` EM iterations, detached, as a generator like the
        reference's (omniglot/model.py:367)."""
        return (p for p in self._unsup_prior(z, init_idx)[:3])

    def get_semisupervised_prior(self, unsupervised_z, supervised_z, y):
        """(pi_k, mu_k, logvar_k, trace_k), detached (omniglot/model.py:370-412).  y: int64 labels [B,Ns] or the
        reference's one-hot float [B,Ns,K]."""
        return (p for p in self._semi_prior(unsupervised_z, supervised_z, y)[:4])

    def _unsup_prior(self, z, init_idx=None):
        """(pi_k, mu_k, trace_k, logw): the prior plus the E-step log-weights the kernels hand on."""
        if init_idx is None:
            init_idx = self._draw_init_idx(z.shape[0], z.shape[1], z.device)
        elif not isinstance(init_idx, torch.Tensor) or not init_idx.is_cuda:
            host = np.asarray(init_idx.numpy() if isinstance(init_idx, torch.Tensor) else init_idx)
            ops.check_index_range(host, z.shape[1], 'init_idx')      # host-side indices: checked before upload
            init_idx = torch.as_tensor(host, dtype=torch.int64).to(z.device)
        with torch.no_grad():
            return ops.unsup_prior(z.detach().contiguous(), init_idx.contiguous(), self._causal_desc(),
                                   self.unsupervised_em_iters, self.fix_pi, self.gamma1)

    def _unsup_prior_begin(self, z, init_idx=None):
        """The EM of `_unsup_prior` issued on a second HIP stream, so that its chain of small dependent launches (33
        kernels of ~15 us on at most B*K workgroups) runs beside whatever the caller issues next on the current stream
        (the decoder's forward pass, which does not depend on it).  Returns a handle for `_unsup_prior_end`.  Same
        kernels, same results; under HIP graph capture or with `em_side_stream = False` it runs in place (as a parallel
        branch of the feature models' captured step it was measured: 1.436 vs 1.436 ms, the graph runs its branches one
        after the other)."""
        if not self.em_side_stream or not z.is_cuda or torch.cuda.is_current_stream_capturing():
            return self._unsup_prior(z, init_idx), None
        if init_idx is None:
            init_idx = self._draw_init_idx(z.shape[0], z.shape[1], z.device)
        elif not isinstance(init_idx, torch.Tensor) or not init_idx.is_cuda:
            host = np.asarray(init_idx.numpy() if isinstance(init_idx, torch.Tensor) else init_idx)
            ops.check_index_range(host, z.shape[1], 'init_idx')
            init_idx = torch.as_tensor(host, dtype=torch.int64).to(z.device)
        zc, ic = z.detach().contiguous(), init_idx.contiguous()
        side = _side_stream(z.device)
        side.wait_stream(torch.cuda.current_stream(z.device))
        with torch.cuda.stream(side), torch.no_grad():
            out = ops.unsup_prior(zc, ic, self._causal_desc(), self.unsupervised_em_iters, self.fix_pi, self.gamma1)
        # zc / ic belong to the current stream's allocator pool and out to the side stream's: each is used on the other
        for t in (zc, ic):
            t.record_stream(side)
        return out, side

    @staticmethod
    def _unsup_prior_end(handle):
        out, side = handle
        if side is not None:
            cur = torch.cuda.current_stream(out[0].device)
            cur.wait_stream(side)
            for t in out:
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)
        return out

    def _semi_prior(self, unsupervised_z, supervised_z, y):
        """(pi_k, mu_k, logvar_k, trace_k, logw)."""
        if y.dim() == 3:
            y = y.argmax(-1)
        if not y.is_cuda:
            ops.check_index_range(y, self.component_size, 'y (support labels)')
            y = y.to(unsupervised_z.device)
        with torch.no_grad():
            return ops.semi_prior(unsupervised_z.contiguous(), supervised_z.contiguous(), y.contiguous(),
                                  self.component_size, self._causal_desc(), self.semisupervised_em_iters,
                                  self.gamma1)

    # ------------------------------------------------------------------ shared tails
    def _kl_and_h(self, q_z, q_z4_logq, q_eps, A, h_lambda, h_c, a_lambda, init_idx, extras, prior=None):
        B = q_z.shape[0]
        pi_k, mu_k, trace_k, _ = self._unsup_prior(q_z, init_idx) if prior is None else self._unsup_prior_end(prior)
        # prior term with gradient to the causal parameters (mu_k, pi_k, trace_k are constants)
        q_eps_mu_k = self.causal(mu_k)
        # the scalar tail in five launches (forward + backward): prior log-weights, mixture log-likelihood, and
        # the three KL means + h_augment (ops.prior_logw, ops.mixture_loglik, ops.elbo_tail)
        logw = ops.prior_logw(pi_k, trace_k, q_eps_mu_k)
        log_p_z = ops.mixture_loglik(q_z, mu_k, logw)
        h = self.h_a(A)
        kl_loss, h_augment = ops.elbo_tail(q_z4_logq, log_p_z, q_eps, h, A, h_lambda, h_c, a_lambda)
        if extras is not None:
            # exports for the parity tests: what the step computed, plus glp(q_eps, 0) per row, which the fused tail
            # only holds as a sum (computed here on the side, outside the graph); extras never selects arithmetic
            with torch.no_grad():
                log_p_eps = (-0.5 * q_eps.detach().pow(2) + self.log_norm_constant).sum(-1)
            extras.update(pi_k=pi_k, mu_k=mu_k, trace_k=trace_k, log_p_z=log_p_z.view(B, -1),
                          log_q_z=q_z4_logq, log_p_eps=log_p_eps, q_z=q_z, q_eps=q_eps, A=A)
        return kl_loss, h_augment, h

    def _tr_predict_from_posterior(self, q_mean, q_logvar, eps, init_idx):
        """Cluster labels of the items of a training meta-batch under the unsupervised prior
        (GMVAE.tr_prediction, omniglot/model.py:587-629): EM, then the few-shot posterior kernel with unit
        variances and the EM's log-weights."""
        B, S, D = q_mean.shape
        M = self.train_mc_sample_size
        if eps is None:
            eps = self._draw_eps(q_mean, M)
        q_z, _ = ops.reparam_logq(q_mean.contiguous(), q_logvar.contiguous(), eps)
        q_z = q_z.view(B, S * M, D)
        pi_k, mu_k, trace_k, logw = self._unsup_prior(q_z, init_idx)
        y_pred, _ = ops.posterior_argmax(q_z, mu_k, torch.zeros_like(mu_k), logw, S, M, want_post=False)
        return y_pred

    def te_prediction(self, X_tr, y_tr, X_te):
        return self.prediction(X_tr, y_tr, X_te)

    def _predict_from_posterior(self, q_mean, q_logvar, y_tr, tr, te, eps, extras):
        B = q_mean.shape[0]
        D, Mt = self.latent_size, self.test_mc_sample_size
        if eps is None:
            eps = self._draw_eps(q_mean, self.train_mc_sample_size)
        q_z, _ = ops.reparam_logq(q_mean.contiguous(), q_logvar.contiguous(), eps)
        # the reference samples with train_mc but reshapes with test_mc (omniglot/model.py:536 vs 542-547)
        uz = q_z[:, tr:tr + te].reshape(B, te * Mt, D).contiguous()
        sz = q_z[:, :tr].reshape(B, tr * Mt, D).contiguous()
        y = y_tr.reshape(B * tr)[:, None].repeat(1, Mt).view(B, tr * Mt)
        pi_k, mu_k, lv_k, trace, logw = self._semi_prior(uz, sz, y)
        y_pred, post = ops.posterior_argmax(uz, mu_k, lv_k, logw, te, Mt, want_post=extras is not None)
        if extras is not None:
            extras.update(pi_k=pi_k, mu_k=mu_k, logvar_k=lv_k, post_mean=post, q_z=q_z)
        return y_pred


class OmniglotGMVAE(_GMVAEBase):
    """Conv-VAE variant (omniglot/model.py).  `linear` defaults to False (NOTEARS-MLP causal layer)."""

    def __init__(self, input_shape=[1, 28, 28], unsupervised_em_iters=5, semisupervised_em_iters=5, fix_pi=False,
                 hidden_size=64, component_size=20, latent_size=64, train_mc_sample_size=10,
                 test_mc_sample_size=10, linear=False, eval_iter=15, laplace=False, gamma=1):
        super().__init__()
        if hidden_size != latent_size:
            raise ValueError('hidden_size must equal latent_size (the reference splits 2*hidden into 2*latent)')
        self._init_common(unsupervised_em_iters, semisupervised_em_iters, fix_pi, component_size, latent_size,
                          train_mc_sample_size, test_mc_sample_size, linear, laplace, gamma)
        self.eval_iter = eval_iter
        self.input_shape = list(input_shape)
        self.hidden_size = hidden_size
        self.last_hidden_size = 2 * 2 * hidden_size
        # construction order fixes the RNG consumption order of the seeded init
        self.encoder = OmniglotEncoder(hidden_size=hidden_size)
        L = self.last_hidden_size
        self.q_z_given_H = nn.Sequential(SAB(L, L, 4, ln=False), SAB(L, L, 4, ln=False),
                                         nn.Linear(L, 2 * hidden_size))
        self.causal = CausalEncoder(hidden_size, linear=linear)
        self.proj = nn.Sequential(nn.Linear(2 * latent_size, L), nn.ELU(inplace=True),
                                  nn.Linear(L, L), nn.ELU(inplace=True),
                                  nn.Linear(L, L), nn.ELU(inplace=True))
        self.decoder = OmniglotDecoder(hidden_size=hidden_size)
        self._register_constants(with_log2=True)

    def _encode(self, X):
        B, S = X.shape[:2]
        # HIP encoder; its output is channels-last, gather the columns back to the reference's c*4+pix order
        H = self.encoder(X.reshape(B * S, -1)).view(B, S, self.last_hidden_size)
        return self._posterior_head(H).split(self.latent_size, dim=-1)

    def forward(self, X, h_lambda, h_c, a_lambda, eps=None, init_idx=None, extras=None):
        B, S = X.shape[:2]
        M, D = self.train_mc_sample_size, self.latent_size
        q_mean, q_logvar = self._encode(X)
        if eps is None:
            eps = self._draw_eps(q_mean, M)
        q_z4, log_q_z = ops.reparam_logq(q_mean.contiguous(), q_logvar.contiguous(), eps)
        q_z = q_z4.view(B, S * M, D)
        # the prior's EM needs q_z only: it runs on a second stream beside the causal layer, proj and the decoder
        prior = self._unsup_prior_begin(q_z, init_idx)
        q_eps = self.causal(ops.center_rows(q_z))
        A = self.causal.get_w().contiguous()      # one layout copy for both consumers (expm_trace, elbo_tail)
        # proj: the last Linear's rows are permuted so H_rec comes out channels-last ([2,2,64] per image)
        h = torch.cat([q_z, q_eps], -1).view(-1, 2 * D)
        h = linear(h, self.proj[0].weight, self.proj[0].bias, 'elu')
        h = linear(h, self.proj[2].weight, self.proj[2].bias, 'elu')
        perm, inv = nhwc_perm(h.device, self.hidden_size)
        H_rec = linear(h, permute(self.proj[4].weight, 0, perm, inv), permute(self.proj[4].bias, 0, perm, inv), 'elu')
        # decoder + Bernoulli likelihood: one fused HIP pipeline (cmvae_amd/decoder.py)
        p_rec, bce_sum = decoder_bce(self.decoder, H_rec, X.reshape(B * S, -1), M)
        X_rec = p_rec.view(B, S, M, *self.input_shape)
        rec_loss = bce_sum / (B * S * M)
        kl_loss, h_augment, h = self._kl_and_h(q_z, log_q_z, q_eps, A, h_lambda, h_c, a_lambda, init_idx, extras, prior)
        if extras is not None:
            extras.update(q_z_mean=q_mean, q_z_logvar=q_logvar)
        return X_rec, rec_loss, kl_loss, h_augment, h

    def prediction(self, X_tr, y_tr, X_te, eps=None, extras=None):
        tr, te = X_tr.shape[1], X_te.shape[1]
        with torch.no_grad():
            q_mean, q_logvar = self._encode(torch.cat([X_tr, X_te], dim=1))
            return self._predict_from_posterior(q_mean, q_logvar, y_tr, tr, te, eps, extras)


    def tr_prediction(self, X, eps=None, init_idx=None):
        with torch.no_grad():
            q_mean, q_logvar = self._encode(X)
            return self._tr_predict_from_posterior(q_mean, q_logvar, eps, init_idx)

    def _decode(self, z):
        """z [n,64] -> images [n,1,28,28]: causal layer, proj, decoder (no likelihood target)."""
        D = self.latent_size
        h = torch.cat([z, self.causal(z)], -1).view(-1, 2 * D)
        h = linear(h, self.proj[0].weight, self.proj[0].bias, 'elu')
        h = linear(h, self.proj[2].weight, self.proj[2].bias, 'elu')
        perm, inv = nhwc_perm(h.device, self.hidden_size)
        H_rec = linear(h, permute(self.proj[4].weight, 0, perm, inv), permute(self.proj[4].bias, 0, perm, inv), 'elu')
        p, _ = decoder_bce(self.decoder, H_rec, torch.zeros(H_rec.shape[0], 784, device=z.device), 1)
        return p.view(-1, *self.input_shape)

    def generate(self, X_tr, y_tr, X_te, seed=None, eps=None, causes=(0, 58)):
        """Figure helper (omniglot/model.py:634-723) for episode 0: (True_X, Recon_X, Component_X, Recon_X_2) —
        reconstructions of the first MC sample of each query item, of the component means, and of the latents
        with the coordinates `causes` zeroed and re-encoded as z - causal(z)."""
        tr, te = X_tr.shape[1], X_te.shape[1]
        B, D, Mt = X_tr.shape[0], self.latent_size, self.test_mc_sample_size
        with torch.no_grad():
            q_mean, q_logvar = self._encode(torch.cat([X_tr, X_te], dim=1))
            if eps is None:
                eps = self._draw_eps(q_mean, self.train_mc_sample_size)
            q_z, _ = ops.reparam_logq(q_mean.contiguous(), q_logvar.contiguous(), eps)
            uz = q_z[:, tr:tr + te].reshape(B, te * Mt, D).contiguous()
            sz = q_z[:, :tr].reshape(B, tr * Mt, D).contiguous()
            y = y_tr.reshape(B * tr)[:, None].repeat(1, Mt).view(B, tr * Mt)
            pi_k, mu_k, lv_k, trace, logw = self._semi_prior(uz, sz, y)
            z_to = uz.view(B, te, Mt, D)[0, :, 0, :].clone()
            Recon_X = self._decode(z_to)
            Component_X = self._decode(mu_k[0].contiguous())
            z_to[:, list(causes)] = 0
            Recon_X_2 = self._decode(z_to - self.causal(z_to))
        return X_te[0], Recon_X, Component_X, Recon_X_2


class FeatureGMVAE(_GMVAEBase):
    """MLP-VAE variant over pre-extracted SimCLR features (mimgnet/model.py; celeba/model.py with
    linear=False)."""

    def __init__(self, input_shape=512, unsupervised_em_iters=5, semisupervised_em_iters=5, fix_pi=False,
                 component_size=20, latent_size=64, train_mc_sample_size=10, test_mc_sample_size=10, linear=True,
                 laplace=False, gamma=1):
        super().__init__()
        self._init_common(unsupervised_em_iters, semisupervised_em_iters, fix_pi, component_size, latent_size,
                          train_mc_sample_size, test_mc_sample_size, linear, laplace, gamma)
        self.input_shape = Fd = int(input_shape)
        if Fd % 64 or Fd > 1024:
            raise ValueError('the MI355X GEMM kernels need a feature size that is a multiple of 64, <= 1024 (got %d)' % Fd)
        self.q_z_given_H = nn.Sequential(SAB(Fd, Fd, 4, ln=False), SAB(Fd, Fd, 4, ln=False),
                                         nn.Linear(Fd, 2 * latent_size))
        self.causal = CausalEncoder(latent_size, linear=linear)
        self.proj = nn.Sequential(nn.Linear(2 * latent_size, 2 * Fd), nn.ReLU(inplace=True),
                                  nn.Linear(2 * Fd, 2 * Fd), nn.ReLU(inplace=True),
                                  nn.Linear(2 * Fd, Fd), nn.ReLU(inplace=True))
        self._register_constants(with_log2=False)

    def forward(self, H, h_lambda, h_c, a_lambda, eps=None, init_idx=None, extras=None):
        B, S = H.shape[:2]
        M, D = self.train_mc_sample_size, self.latent_size
        q_mean, q_logvar = self._posterior_head(H).split(D, dim=-1)
        if eps is None:
            eps = self._draw_eps(q_mean, M)
        q_z4, log_q_z = ops.reparam_logq(q_mean.contiguous(), q_logvar.contiguous(), eps)
        q_z = q_z4.view(B, S * M, D)
        q_eps = self.causal(q_z)
        A = self.causal.get_w().contiguous()      # one layout copy for both consumers (expm_trace, elbo_tail)
        h = torch.cat([q_z, q_eps], -1).view(-1, 2 * D)
        for i in (0, 2):
            h = linear(h, self.proj[i].weight, self.proj[i].bias, 'relu')
        # the last proj layer and the squared error against the input features leave one GEMM launch
        h, sq_err = linear_mse(h, self.proj[4].weight, self.proj[4].bias, 'relu', H, M)
        H_rec = h.view(B, S, M, self.input_shape)
        rec_loss = sq_err / (B * S * M)
        kl_loss, h_augment, h = self._kl_and_h(q_z, log_q_z, q_eps, A, h_lambda, h_c, a_lambda, init_idx, extras)
        if extras is not None:
            extras.update(q_z_mean=q_mean, q_z_logvar=q_logvar)
        return H_rec, rec_loss, kl_loss, h_augment, h

    def tr_prediction(self, H, eps=None, init_idx=None):
        with torch.no_grad():
            q_mean, q_logvar = self._posterior_head(H).split(self.latent_size, dim=-1)
            return self._tr_predict_from_posterior(q_mean, q_logvar, eps, init_idx)

    def prediction(self, H_tr, y_tr, H_te, eps=None, extras=None):
        tr, te = H_tr.shape[1], H_te.shape[1]
        with torch.no_grad():
            q_mean, q_logvar = self._posterior_head(torch.cat([H_tr, H_te], dim=1)).split(self.latent_size, dim=-1)
            return self._predict_from_posterior(q_mean, q_logvar, y_tr, tr, te, eps, extras)
