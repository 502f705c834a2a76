"""CPU oracle for the CMVAE episodic train / eval hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (GuodongQi/CMVAE) ships no tests, golden vectors or
fixtures, and executing its Python is denied in this pipeline.  This
file is therefore a restatement, written from the reference's source *text*, of the
arithmetic the reference performs with PyTorch ATen ops, in the same operation order
(including the `repeat`-materialised E-step, the per-iteration batched `inverse`, and the
un-fused decoder / BatchNorm / BCE).  It is pinned only by the analytic known-answer
tests in tests/test_oracle_kat.py and by self-generated fixtures in tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product (cmvae_amd/) never does.

Everything is functional: parameters come in as a flat ``state_dict``-style mapping whose
key names are the reference's module paths, so the same weights drive the
HIP path and this oracle.

Reference lines followed (paths under /root/reference):
  glp                     omniglot/model.py:89-95
  reparametrize           omniglot/model.py:111-122
  h_a                     omniglot/model.py:159-166
  unsup_params            omniglot/model.py:168-222
  semi_params             omniglot/model.py:224-330
  unsup_prior             omniglot/model.py:332-368
  semi_prior              omniglot/model.py:370-412
  forward (omniglot)      omniglot/model.py:431-518
  forward (features)      mimgnet/model.py:364-440 (celeba/model.py identical, linear=False)
  prediction              omniglot/model.py:520-585, mimgnet/model.py:442-511
  encoder / decoder       omniglot/module.py:7-40, 43-71
  mab (SAB)               omniglot/module.py:75-112
  causal_mask / get_w     omniglot/module.py:165-196, mimgnet/module.py:77-110
  locally connected       omniglot/module.py:273-280
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _log_norm_constant(P, like):
    c = P.get('log_norm_constant') if hasattr(P, 'get') else None
    if c is None:
        c = torch.tensor(-0.5 * np.log(2 * np.pi))
    # 0-dim buffer: does not promote fp32 operands; in fp64 runs it stays fp64
    return c.to(dtype=like.dtype) if like.dtype == torch.float64 else c


def glp(x, mean, logvar=None, c=None):
    """Diagonal-Gaussian log density summed over the last dim."""
    if c is None:
        c = torch.tensor(-0.5 * np.log(2 * np.pi))
    if logvar is None:
        logvar = torch.zeros_like(mean)
    a = (x - mean).pow(2)
    log_p = -0.5 * (logvar + a / logvar.exp())
    log_p = log_p + c
    return log_p.sum(dim=-1)


# ----------------------------------------------------------------------------- causal layer
def causal_mask(x, P, linear, latent=64, m1=32):
    if linear:
        x_ = x.reshape(-1, x.shape[-2], x.shape[-1])
        x_ = x_.bmm(P['causal.A'].unsqueeze(0).repeat(x_.shape[0], 1, 1))
        return x_.view(x.shape)
    h = F.linear(x, P['causal.fc1.weight'], P['causal.fc1.bias'])
    h = h.view(-1, latent, m1)
    h = F.elu(h)
    # locally connected: [n,d,1,m1] @ [1,d,m1,1] (+ bias [d,1])
    out = torch.matmul(h.unsqueeze(2), P['causal.fc2.0.weight'].unsqueeze(0)).squeeze(2)
    out = out + P['causal.fc2.0.bias']
    return out.squeeze(2).view(x.shape)


def causal(x, P, linear):
    return x - causal_mask(x, P, linear)


def causal_get_w(P, linear, latent=64):
    if linear:
        return P['causal.A']
    w = P['causal.fc1.weight'].view(latent, -1, latent)
    return w.pow(2).sum(1).t()


def h_a(A, latent):
    return torch.trace(torch.matrix_exp(A * A)) - latent


# ----------------------------------------------------------------------------- set attention
def _tap(taps, name, t):
    """Record the input of a non-smooth op (ReLU / max-pool) so a test can compare the masks of two runs."""
    if taps is not None:
        taps.setdefault(name, []).append(t.detach())


def mab(X, P, prefix, heads=4, taps=None):
    q = F.linear(X, P[prefix + '.mab.fc_q.weight'], P[prefix + '.mab.fc_q.bias'])
    k = F.linear(X, P[prefix + '.mab.fc_k.weight'], P[prefix + '.mab.fc_k.bias'])
    v = F.linear(X, P[prefix + '.mab.fc_v.weight'], P[prefix + '.mab.fc_v.bias'])
    dim_v = q.shape[-1]
    ds = dim_v // heads
    q_ = torch.cat(q.split(ds, 2), 0)
    k_ = torch.cat(k.split(ds, 2), 0)
    v_ = torch.cat(v.split(ds, 2), 0)
    a = torch.softmax(q_.bmm(k_.transpose(1, 2)) / math.sqrt(dim_v), 2)
    o = torch.cat((q_ + a.bmm(v_)).split(q.size(0), 0), 2)
    pre = F.linear(o, P[prefix + '.mab.fc_o.weight'], P[prefix + '.mab.fc_o.bias'])
    _tap(taps, 'linear_relu', pre)
    o = o + F.relu(pre)
    return o


def q_z_given_h(H, P, taps=None):
    o = mab(H, P, 'q_z_given_H.0', taps=taps)
    o = mab(o, P, 'q_z_given_H.1', taps=taps)
    return F.linear(o, P['q_z_given_H.2.weight'], P['q_z_given_H.2.bias'])


# ----------------------------------------------------------------------------- conv encoder / decoder
def _bn(x, P, name, training, running=None):
    rm = P[name + '.running_mean']
    rv = P[name + '.running_var']
    if training:
        # F.batch_norm updates running stats in place: work on private copies
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, P[name + '.weight'], P[name + '.bias'],
                     training=training, momentum=0.1, eps=1e-5)
    if training and running is not None:
        running[name + '.running_mean'] = rm
        running[name + '.running_var'] = rv
    return y


def omniglot_encoder(x, P, training, running=None, taps=None):
    h = x
    for conv, bn in ((0, 1), (4, 5), (8, 9), (12, 13)):
        h = F.conv2d(h, P['encoder.encoder.%d.weight' % conv], P['encoder.encoder.%d.bias' % conv],
                     stride=1, padding=1)
        h = _bn(h, P, 'encoder.encoder.%d' % bn, training, running)
        _tap(taps, 'enc_bn', h)
        h = F.relu(h)
        h = F.max_pool2d(h, 2, ceil_mode=True)
    return h.view(h.shape[0], -1)


def omniglot_decoder(h, P, training, hidden=64, running=None, taps=None):
    x = h.view(-1, hidden, 2, 2)
    spec = ((0, 1, 4), (3, 4, 3), (6, 7, 4))
    for conv, bn, k in spec:
        x = F.conv_transpose2d(x, P['decoder.decoder.%d.weight' % conv], P['decoder.decoder.%d.bias' % conv],
                               stride=2, padding=1)
        x = _bn(x, P, 'decoder.decoder.%d' % bn, training, running)
        _tap(taps, 'dec_bn', x)
        x = F.relu(x)
    x = F.conv_transpose2d(x, P['decoder.decoder.9.weight'], P['decoder.decoder.9.bias'], stride=2, padding=1)
    return torch.sigmoid(x)


def proj(x, P, act, taps=None):
    for i in (0, 2, 4):
        x = F.linear(x, P['proj.%d.weight' % i], P['proj.%d.bias' % i])
        if act is F.relu:
            _tap(taps, 'linear_relu', x)
        x = act(x)
    return x


# ----------------------------------------------------------------------------- the model
class OracleGMVAE:
    """Functional restatement of both GMVAE variants.

    variant='omniglot': conv encoder/decoder, ELU proj, BCE, centred bulk causal input.
    variant='feat'    : SimCLR-feature model (mimgnet: linear=True; celeba: linear=False).
    """

    def __init__(self, P, variant='omniglot', input_shape=(1, 28, 28), unsupervised_em_iters=10,
                 semisupervised_em_iters=10, fix_pi=False, hidden_size=64, component_size=20,
                 latent_size=64, train_mc_sample_size=32, test_mc_sample_size=32, linear=None, gamma=1):
        self.P = P
        self.variant = variant
        self.input_shape = list(input_shape) if variant == 'omniglot' else int(input_shape)
        self.unsup_iters = unsupervised_em_iters
        self.semi_iters = semisupervised_em_iters
        self.fix_pi = fix_pi
        self.hidden = hidden_size
        self.last_hidden = 4 * hidden_size
        self.K = component_size
        self.D = latent_size
        self.train_mc = train_mc_sample_size
        self.test_mc = test_mc_sample_size
        self.linear = (variant != 'omniglot') if linear is None else linear
        self.gamma1 = 1 / math.sqrt(gamma)
        self.training = True
        self.running = {}
        self.taps = None          # set to a dict to record the inputs of every ReLU / max-pool of a forward
        any_param = next(iter(P.values()))
        ref = P['q_z_given_H.2.weight']
        self.I = torch.eye(latent_size, dtype=ref.dtype)
        self.c = _log_norm_constant(P, ref)
        self.uniform_pi = torch.ones(self.K, dtype=ref.dtype) / self.K
        del any_param

    # -- helpers
    def _glp(self, x, mean, logvar=None):
        return glp(x, mean, logvar, self.c)

    def _causal(self, x):
        return causal(x, self.P, self.linear)

    def reparametrize(self, mean, logvar, S, eps=None):
        mean = mean.unsqueeze(2).repeat(1, 1, S, 1)
        logvar = logvar.unsqueeze(2).repeat(1, 1, S, 1)
        std = logvar.mul(0.5).exp()
        if eps is None:
            eps = torch.randn_like(mean)
        return eps.mul(std).add(mean)

    def _minv0(self):
        sig = self._causal(self.I)
        sig2 = sig.mm(sig.T)
        return sig2

    # -- EM: one unsupervised iteration
    def unsup_params(self, X, psi):
        N = X.shape[1]
        K, D = self.K, self.D
        pi, mean, trace = psi
        ll = self._glp(
            X[:, :, None, :].repeat(1, 1, K, 1),
            mean[:, None, :, :].repeat(1, N, 1, 1),
        ) + torch.log(pi[:, None, :].repeat(1, N, 1)) + self._glp(
            self._causal(mean),
            torch.log(trace)[:, :, None].repeat(1, 1, D)
        ).unsqueeze(1)
        post = torch.exp(ll - torch.logsumexp(ll, dim=-1, keepdim=True))
        Nk = torch.sum(post, dim=1)
        Nk = torch.where(Nk.eq(0.0), 1e-22 + torch.zeros_like(Nk), Nk)
        trace = self.gamma1 * torch.ones_like(Nk)
        std_causal = self._causal(self.I[None, None, :, :] * trace[:, :, None, None])
        inverse = torch.inverse(self.I[None, None, :, :] + std_causal.matmul(std_causal.transpose(-1, -2)))
        if not self.fix_pi:
            pi = Nk / Nk.sum(dim=-1, keepdim=True)
        denom = Nk[:, :, None].repeat(1, 1, D)
        mean = torch.matmul(post.permute(0, 2, 1).contiguous(), X) / denom
        mean = mean.unsqueeze(2).matmul(inverse).squeeze(2)
        return pi, mean, trace

    # -- EM: one semi-supervised iteration
    def semi_params(self, uX, sX, y, psi):
        Nu = uX.shape[1]
        B = uX.shape[0]
        K, D = self.K, self.D
        pi, mean, logvar, trace = psi
        ll = self._glp(
            uX[:, :, None, :].repeat(1, 1, K, 1),
            mean[:, None, :, :].repeat(1, Nu, 1, 1),
        ) + torch.log(pi[:, None, :].repeat(1, Nu, 1)) + self._glp(
            self._causal(mean),
            torch.log(trace)[:, :, None].repeat(1, 1, D)
        ).unsqueeze(1)
        post = torch.exp(ll - torch.logsumexp(ll, dim=-1, keepdim=True))
        sN = torch.sum(y, dim=1)
        uN = torch.sum(post, dim=1)
        denom = sN[:, :, None].repeat(1, 1, D) + uN[:, :, None].repeat(1, 1, D)
        trace = self.gamma1 * torch.ones_like(sN)
        std = torch.diag_embed((0.5 * logvar).exp()) * trace[:, :, None, None]
        std_causal = self._causal(std.view(B, -1, D)).view(std.shape)
        inverse = torch.inverse(self.I[None, None, :, :] + std_causal.matmul(std_causal.transpose(-1, -2)))
        yT = y.permute(0, 2, 1).contiguous()
        pT = post.permute(0, 2, 1).contiguous()
        s_mean = torch.matmul(yT, sX)
        u_mean = torch.matmul(pT, uX)
        mean = (s_mean + u_mean) / denom
        mean = mean.unsqueeze(2).matmul(inverse).squeeze(2)
        s_x2 = torch.matmul(yT, sX.pow(2.0))
        s_xm = mean * torch.matmul(yT, sX)
        s_m2 = sN[:, :, None].repeat(1, 1, D) * mean.pow(2.0)
        s_var = s_x2 - 2 * s_xm + s_m2
        u_x2 = torch.matmul(pT, uX.pow(2.0))
        u_xm = mean * torch.matmul(pT, uX)
        u_m2 = uN[:, :, None].repeat(1, 1, D) * mean.pow(2.0)
        u_var = u_x2 - 2 * u_xm + u_m2
        var = (s_var + u_var) / denom
        logvar = torch.log(var)
        return pi, mean, logvar, trace

    def _initial_inverse(self, init_trace):
        sig2 = self._minv0()
        return torch.inverse(self.I[None, None, :, :] + 1 / init_trace[:, :, None, None] * sig2[None, None, :, :])

    def unsup_prior(self, z, init_idx=None, iters=None, trajectory=None):
        B, N = z.shape[0], z.shape[1]
        pi0 = self.uniform_pi[None, :].repeat(B, 1)
        tr0 = torch.ones_like(pi0)
        inverse = self._initial_inverse(tr0)
        rows = []
        for i in range(B):
            if init_idx is None:
                idx = torch.from_numpy(np.random.choice(N, self.K, replace=False))
            else:
                idx = torch.as_tensor(init_idx[i], dtype=torch.long)
            rows.append(torch.index_select(z[i], dim=0, index=idx))
        mean0 = torch.stack(rows, dim=0).unsqueeze(2).matmul(inverse).squeeze(2)
        psi = (pi0, mean0, tr0)
        if trajectory is not None:
            trajectory.append(tuple(p.detach().clone() for p in psi))
        for _ in range(self.unsup_iters if iters is None else iters):
            psi = self.unsup_params(z, psi)
            if trajectory is not None:
                trajectory.append(tuple(p.detach().clone() for p in psi))
        return tuple(p.detach() for p in psi)

    def semi_prior(self, uz, sz, y, iters=None, trajectory=None):
        B = uz.shape[0]
        pi0 = self.uniform_pi[None, :].repeat(B, 1)
        tr0 = torch.ones_like(pi0)
        inverse = self._initial_inverse(tr0)
        sN = y.sum(dim=1)
        mean0 = torch.matmul(y.permute(0, 2, 1).contiguous(), sz) / sN[:, :, None].repeat(1, 1, self.D)
        mean0 = mean0.unsqueeze(2).matmul(inverse).squeeze(2)
        lv0 = torch.zeros_like(mean0)
        psi = (pi0, mean0, lv0, tr0)
        if trajectory is not None:
            trajectory.append(tuple(p.detach().clone() for p in psi))
        for _ in range(self.semi_iters if iters is None else iters):
            psi = self.semi_params(uz, sz, y, psi)
            if trajectory is not None:
                trajectory.append(tuple(p.detach().clone() for p in psi))
        return tuple(p.detach() for p in psi)

    # -- encode: X -> (q_z_mean, q_z_logvar)
    def encode(self, X):
        B, S = X.shape[:2]
        if self.variant == 'omniglot':
            H = omniglot_encoder(X.reshape(-1, *self.input_shape), self.P, self.training, self.running, self.taps)
            H = H.view(B, S, self.last_hidden)
        else:
            H = X
        return q_z_given_h(H, self.P, self.taps).split(self.D, dim=-1)

    # -- the training forward
    def forward(self, X, h_lambda, h_c, a_lambda, eps=None, init_idx=None, extras=None):
        B, S = X.shape[:2]
        M, D, K = self.train_mc, self.D, self.K
        q_mean, q_logvar = self.encode(X)
        q_z = self.reparametrize(q_mean, q_logvar, M, eps).view(B, -1, D)
        if self.variant == 'omniglot':
            q_eps = self._causal(q_z - q_z.mean(1, keepdim=True))
        else:
            q_eps = self._causal(q_z)
        A = causal_get_w(self.P, self.linear, D)
        h_in = torch.cat([q_z, q_eps], -1).view(-1, 2 * D)
        if self.variant == 'omniglot':
            H_rec = proj(h_in, self.P, F.elu)
            X_rec = omniglot_decoder(H_rec, self.P, self.training, self.hidden, self.running, self.taps)
            X_rec = X_rec.view(B, S, M, *self.input_shape)
            rec_loss = F.binary_cross_entropy(
                X_rec, X[:, :, None, :, :, :].repeat(1, 1, M, 1, 1, 1), reduction='sum') / (B * S * M)
        else:
            H_rec = proj(h_in, self.P, F.relu, self.taps).view(B, S, M, self.input_shape)
            X_rec = H_rec
            Hrep = X[:, :, None, :].repeat(1, 1, M, 1)
            rec_loss = F.mse_loss(H_rec, Hrep, reduction='sum') / (B * S * M)

        with torch.no_grad():
            pi_k, mu_k, tr_k = self.unsup_prior(q_z, init_idx)
        eps_mu_k = self._causal(mu_k)
        ll = self._glp(
            q_z.view(B, S, M, D)[:, :, :, None, :].repeat(1, 1, 1, K, 1),
            mu_k[:, None, None, :, :].repeat(1, S, M, 1, 1),
        ) + torch.log(pi_k[:, None, None, :].repeat(1, S, M, 1)) + self._glp(
            eps_mu_k, torch.log(tr_k)[:, :, None].repeat(1, 1, D)
        ).unsqueeze(1).unsqueeze(2)
        log_p_z = torch.logsumexp(ll, dim=-1)
        log_q_z = self._glp(
            q_z.view(B, S, M, D),
            q_mean[:, :, None, :].repeat(1, 1, M, 1),
            q_logvar[:, :, None, :].repeat(1, 1, M, 1))
        log_p_eps = self._glp(q_eps, torch.zeros_like(q_eps))
        kl_loss = log_q_z.mean() - log_p_z.mean() - log_p_eps.mean()
        h = h_a(A, D)
        h_aug = h_lambda * h + 0.5 * h_c * h * h + a_lambda * A.abs().sum()
        if extras is not None:
            extras.update(q_z_mean=q_mean, q_z_logvar=q_logvar, q_z=q_z, q_eps=q_eps, pi_k=pi_k, mu_k=mu_k,
                          trace_k=tr_k, log_p_z=log_p_z, log_q_z=log_q_z, log_p_eps=log_p_eps, A=A)
        return X_rec, rec_loss, kl_loss, h_aug, h

    # -- few-shot prediction
    def prediction(self, X_tr, y_tr, X_te, eps=None, extras=None):
        B, tr = X_tr.shape[:2]
        te = X_te.shape[1]
        D, K, Mt = self.D, self.K, self.test_mc
        X = torch.cat([X_tr, X_te], dim=1)
        was_training = self.training
        with torch.no_grad():
            q_mean, q_logvar = self.encode(X)
            q_z = self.reparametrize(q_mean, q_logvar, self.train_mc, eps)
            uz = q_z[:, tr:(tr + te), :, :].reshape(B, te * Mt, D)
            sz = q_z[:, :tr, :, :].reshape(B, tr * Mt, D)
            y = y_tr.reshape(B * tr)[:, None].repeat(1, Mt).view(B, tr * Mt)
            y = F.one_hot(y, K).to(q_z.dtype)
            pi_k, mu_k, lv_k, trace = self.semi_prior(uz, sz, y)
            ll = self._glp(
                uz.view(B, te, Mt, D)[:, :, :, None, :].repeat(1, 1, 1, K, 1),
                mu_k[:, None, None, :, :].repeat(1, te, Mt, 1, 1),
                lv_k[:, None, None, :, :].repeat(1, te, Mt, 1, 1)
            ) + torch.log(pi_k[:, None, None, :].repeat(1, te, Mt, 1)) + self._glp(
                self._causal(mu_k), torch.log(trace)[:, :, None].repeat(1, 1, D)
            ).unsqueeze(1).unsqueeze(2)
            post = torch.exp(ll - torch.logsumexp(ll, dim=-1, keepdim=True))
            pm = post.mean(dim=-2)
            y_pred = pm.argmax(dim=-1)
        self.training = was_training
        if extras is not None:
            extras.update(pi_k=pi_k, mu_k=mu_k, logvar_k=lv_k, post_mean=pm, q_z=q_z)
        return y_pred


    # -- inspection helpers (omniglot/model.py:587-629, 634-723)
    def tr_prediction(self, X, eps=None, init_idx=None):
        """Cluster label of every training item under the unsupervised prior (omniglot/model.py:587-629)."""
        B, S = X.shape[:2]
        M, D, K = self.train_mc, self.D, self.K
        with torch.no_grad():
            q_mean, q_logvar = self.encode(X)
            q_z = self.reparametrize(q_mean, q_logvar, M, eps).view(B, -1, D)
            pi_k, mu_k, tr_k = self.unsup_prior(q_z, init_idx)
            ll = self._glp(
                q_z.view(B, S, M, D)[:, :, :, None, :].repeat(1, 1, 1, K, 1),
                mu_k[:, None, None, :, :].repeat(1, S, M, 1, 1),
            ) + torch.log(pi_k[:, None, None, :].repeat(1, S, M, 1)) + self._glp(
                self._causal(mu_k), torch.log(tr_k)[:, :, None].repeat(1, 1, D)
            ).unsqueeze(1).unsqueeze(2)
            post = torch.exp(ll - torch.logsumexp(ll, dim=-1, keepdim=True))
            return post.mean(dim=-2).argmax(dim=-1)

    def generate(self, X_tr, y_tr, X_te, eps=None, causes=(0, 58)):
        """(True_X, Recon_X, Component_X, Recon_X_2) of episode 0 (omniglot/model.py:634-723): reconstructions of
        the first MC sample of every query item, of the component means, and of the intervened latents (the
        coordinates `causes` set to zero, then z - causal(z) re-encoded).  As in the reference the intervention
        writes through the view, so Recon_X_2 starts from the zeroed z."""
        assert self.variant == 'omniglot'
        B, tr = X_tr.shape[:2]
        te = X_te.shape[1]
        D, K, Mt = self.D, self.K, self.test_mc
        with torch.no_grad():
            q_mean, q_logvar = self.encode(torch.cat([X_tr, X_te], dim=1))
            q_z = self.reparametrize(q_mean, q_logvar, self.train_mc, eps)
            uz = q_z[:, tr:(tr + te)].reshape(B, te * Mt, D)
            sz = q_z[:, :tr].reshape(B, tr * Mt, D)
            y = F.one_hot(y_tr.reshape(B * tr)[:, None].repeat(1, Mt).view(B, tr * Mt), K).to(q_z.dtype)
            pi_k, mu_k, lv_k, trace = self.semi_prior(uz, sz, y)

            def decode(z):
                H = proj(torch.cat([z, self._causal(z)], -1), self.P, F.elu)
                return omniglot_decoder(H, self.P, self.training, self.hidden, self.running)

            z_to = uz.view(B, te, Mt, D)[0, :, 0, :].clone()
            Recon_X = decode(z_to)
            Component_X = decode(mu_k[0])
            z_to[:, list(causes)] = 0
            z_ = z_to - self._causal(z_to)
            Recon_X_2 = decode(z_)
        return X_te[0], Recon_X, Component_X, Recon_X_2


# ----------------------------------------------------------------------------- CPU baseline helper
def make_trainable(P):
    """Clone a state dict into leaf tensors that require grad (floating-point, >0-dim, non-buffer)."""
    out = {}
    for k, v in P.items():
        v = v.detach().clone()
        is_buf = k.endswith(('running_mean', 'running_var', 'num_batches_tracked')) or \
            k in ('log_norm_constant', 'log_2', 'uniform_pi')
        if v.is_floating_point() and not is_buf:
            v.requires_grad_(True)
        out[k] = v
    return out


def train_step(model, opt, X, h_lambda, h_c, a_lambda, eps=None, init_idx=None):
    """One reference-style optimisation step (omniglot/trainer.py:79-95)."""
    opt.zero_grad(set_to_none=True)
    X_rec, rec, kl, h_aug, h = model.forward(X, h_lambda, h_c, a_lambda, eps=eps, init_idx=init_idx)
    loss = rec + kl + h_aug
    loss.backward()
    opt.step()
    for k, v in model.running.items():
        model.P[k].copy_(v)
    return rec.detach(), kl.detach(), h_aug.detach(), h.detach()
