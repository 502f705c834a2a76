"""torch custom ops (`torch.ops.cmvae_amd.*`) over the C-ABI of libcmvae_hip.so.

Each op is a thin shim: validate shapes on the host (a hand-written kernel must never see
operands it does not expect), allocate outputs/workspace with torch (plumbing), pass raw
device pointers and the current HIP stream through ctypes.  No CPU fallback exists: CPU
tensors raise.
"""
import ctypes
import os

import torch

from ._lib import lib, check, CausalDesc

D = 64
_NS = 'cmvae_amd'
_deflib = torch.library.Library(_NS, 'DEF')


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t, name, dtype=torch.float32, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('cmvae_amd: %s must be a tensor on the MI355X device (no CPU path exists)' % name)
    if t.dtype != dtype:
        raise RuntimeError('cmvae_amd: %s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError('cmvae_amd: %s must be contiguous' % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError('cmvae_amd: %s has shape %s, expected %s' % (name, tuple(t.shape), tuple(shape)))
    return t


# CMVAE_CHECK_INDICES=1: range-check EM seeds and few-shot labels that already live on the GPU (one device sync per
# call; off by default).  Indices that originate on the host are always checked before upload (model.py).
CHECK_INDICES = os.environ.get('CMVAE_CHECK_INDICES', '0') not in ('', '0')


def check_index_range(t, hi, name):
    """Raise if any entry of the integer tensor / array lies outside [0, hi)."""
    if (t.numel() if hasattr(t, 'numel') else t.size) == 0:
        return
    lo_v, hi_v = int(t.min()), int(t.max())
    if lo_v < 0 or hi_v >= hi:
        raise RuntimeError('cmvae_amd: %s has entries in [%d, %d], outside [0, %d)' % (name, lo_v, hi_v, hi))


class Causal:
    """Device-side descriptor of a CausalEncoder's parameters (struct cmvae_causal)."""

    def __init__(self, linear, W, b1=None, w2=None, b2=None):
        self.linear = bool(linear)
        if self.linear:
            self.t = (_req(W.detach(), 'causal.A', shape=(D, D)),)
        else:
            self.t = (_req(W.detach(), 'causal.fc1.weight', shape=(D * 32, D)),
                      _req(b1.detach(), 'causal.fc1.bias', shape=(D * 32,)),
                      _req(w2.detach().reshape(D, 32), 'causal.fc2.0.weight', shape=(D, 32)),
                      _req(b2.detach().reshape(D), 'causal.fc2.0.bias', shape=(D,)))
        if self.linear:
            self.desc = CausalDesc(1, self.t[0].data_ptr(), None, None, None)
        else:
            self.desc = CausalDesc(0, *(x.data_ptr() for x in self.t))
        self.ref = ctypes.byref(self.desc)


# ----------------------------------------------------------------------------- plain (no-grad) ops
def causal_rows(x, causal, pi=None, tau=None, want_eps=True):
    """eps = x - mask(x) for a small batch of rows; optionally logw = log pi + glp(eps, log tau)."""
    x = _req(x, 'x')
    rows = x.numel() // D
    if x.shape[-1] != D:
        raise RuntimeError('cmvae_amd: causal rows must have last dim 64')
    eps = torch.empty_like(x) if want_eps else None
    logw = None
    if pi is not None:
        _req(pi, 'pi'); _req(tau, 'tau')
        if pi.numel() != rows or tau.numel() != rows:
            raise RuntimeError('cmvae_amd: pi/tau must have one entry per row')
        logw = torch.empty_like(pi)
    check(lib.cmvae_causal_rows(_p(x), rows, causal.ref, _p(pi), _p(tau), _p(eps), _p(logw), _stream()),
          'cmvae_causal_rows')
    return eps, logw


def gram_inverse(causal, logvar=None, s0=1.0, nmat=1):
    """(I + C C^T)^-1 with C row i = causal(s_i e_i); s = exp(0.5*logvar)*s0 or s0."""
    ref = causal.t[0]
    if logvar is not None:
        _req(logvar, 'logvar')
        if logvar.shape[-1] != D:
            raise RuntimeError('cmvae_amd: logvar last dim must be 64')
        nmat = logvar.numel() // D
    out = torch.empty(nmat, D, D, device=ref.device, dtype=torch.float32)
    check(lib.cmvae_gram_inverse(_p(logvar), float(s0), nmat, causal.ref, _p(out), _stream()), 'cmvae_gram_inverse')
    return out


def unsup_prior(z, init_idx, causal, iters, fix_pi=False, gamma1=1.0):
    """GMVAE.get_unsupervised_prior: returns (pi [B,K], mu [B,K,64], tau [B,K], logw [B,K])."""
    z = _req(z, 'z')
    if z.dim() != 3 or z.shape[-1] != D:
        raise RuntimeError('cmvae_amd: z must be [B,N,64]')
    B, N, _ = z.shape
    if B < 1 or N < 1:
        raise RuntimeError('cmvae_amd: unsup_prior needs at least one episode and one latent')
    init_idx = _req(init_idx, 'init_idx', torch.int64)
    if init_idx.dim() != 2 or init_idx.shape[0] != B:
        raise RuntimeError('cmvae_amd: init_idx must be [B,K]')
    K = init_idx.shape[1]
    if not (1 <= K <= 32):
        raise RuntimeError('cmvae_amd: K=%d outside [1,32]' % K)
    if K > N:
        raise RuntimeError('cmvae_amd: K=%d seeds from N=%d latents' % (K, N))
    if CHECK_INDICES:
        check_index_range(init_idx, N, 'init_idx')
    ws_bytes = lib.cmvae_unsup_prior_workspace_bytes(B, N, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=z.device)
    pi = torch.empty(B, K, device=z.device); tau = torch.empty(B, K, device=z.device)
    logw = torch.empty(B, K, device=z.device); mu = torch.empty(B, K, D, device=z.device)
    check(lib.cmvae_unsup_prior(_p(z), B, N, K, _p(init_idx), causal.ref, int(iters), int(bool(fix_pi)),
                                float(gamma1), _p(ws), ws_bytes, _p(pi), _p(mu), _p(tau), _p(logw), _stream()),
          'cmvae_unsup_prior')
    return pi, mu, tau, logw


def em_pass(z, mu, logw, ws=None):
    """One fused E-step + M-step pass (partials left in the workspace); for measurement and tests."""
    B, N, K = _mix_shapes(z, mu, logw)
    ws_bytes = lib.cmvae_unsup_prior_workspace_bytes(B, N, K)
    if ws is None:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=z.device)
    nblk = ctypes.c_int(0)
    check(lib.cmvae_em_pass(_p(z), _p(mu), _p(logw), B, N, K, _p(ws), ws.numel(), ctypes.byref(nblk), _stream()),
          'cmvae_em_pass')
    return ws, nblk.value


def semi_prior(uz, sz, y, K, causal, iters, gamma1=1.0):
    """GMVAE.get_semisupervised_prior: returns (pi, mu, logvar, tau, logw)."""
    uz = _req(uz, 'unsupervised_z'); sz = _req(sz, 'supervised_z'); y = _req(y, 'y', torch.int64)
    if uz.dim() != 3 or sz.dim() != 3 or uz.shape[-1] != D or sz.shape[-1] != D or uz.shape[0] != sz.shape[0]:
        raise RuntimeError('cmvae_amd: uz/sz must be [B,N,64] with equal B')
    B, Nu, _ = uz.shape
    Ns = sz.shape[1]
    if B < 1 or Nu < 1 or Ns < 1:
        raise RuntimeError('cmvae_amd: semi_prior needs at least one episode, one query and one support latent')
    if tuple(y.shape) != (B, Ns):
        raise RuntimeError('cmvae_amd: y must be [B,Ns] int64 labels')
    if not (1 <= K <= 32):
        raise RuntimeError('cmvae_amd: K=%d outside [1,32]' % K)
    if CHECK_INDICES:
        check_index_range(y, K, 'y (support labels)')
    ws_bytes = lib.cmvae_semi_prior_workspace_bytes(B, Nu, Ns, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=uz.device)
    dev = uz.device
    pi = torch.empty(B, K, device=dev); tau = torch.empty(B, K, device=dev); logw = torch.empty(B, K, device=dev)
    mu = torch.empty(B, K, D, device=dev); logvar = torch.empty(B, K, D, device=dev)
    check(lib.cmvae_semi_prior(_p(uz), _p(sz), _p(y), B, Nu, Ns, K, causal.ref, int(iters), float(gamma1),
                               _p(ws), ws_bytes, _p(pi), _p(mu), _p(logvar), _p(tau), _p(logw), _stream()),
          'cmvae_semi_prior')
    return pi, mu, logvar, tau, logw


def posterior_argmax(z, mu, logvar, logw, te, M, want_post=False):
    """Final few-shot posterior: int64 labels [B,te] (and optionally the MC-averaged posteriors)."""
    z = _req(z, 'z'); mu = _req(mu, 'mu'); logvar = _req(logvar, 'logvar'); logw = _req(logw, 'logw')
    B, K = logw.shape
    if tuple(z.shape) != (B, te * M, D) or tuple(mu.shape) != (B, K, D) or tuple(logvar.shape) != (B, K, D):
        raise RuntimeError('cmvae_amd: posterior_argmax shape mismatch')
    y = torch.empty(B, te, dtype=torch.int64, device=z.device)
    post = torch.empty(B, te, K, device=z.device) if want_post else None
    check(lib.cmvae_posterior_argmax(_p(z), _p(mu), _p(logvar), _p(logw), B, te, M, K, _p(y), _p(post), _stream()),
          'cmvae_posterior_argmax')
    return y, post


# ----------------------------------------------------------------------------- registered kernels with autograd
_deflib.define('mixture_loglik_fwd(Tensor z, Tensor mu, Tensor logw) -> Tensor')
_deflib.define('mixture_loglik_bwd(Tensor z, Tensor mu, Tensor logw, Tensor g) -> (Tensor, Tensor)')
_deflib.define('reparam_logq_fwd(Tensor mean, Tensor logvar, Tensor eps) -> (Tensor, Tensor)')
_deflib.define('reparam_logq_bwd(Tensor logvar, Tensor eps, Tensor gz, Tensor glogq) -> (Tensor, Tensor)')
_deflib.define('expm_trace(Tensor A) -> (Tensor, Tensor)')


def _mix_shapes(z, mu, logw):
    z = _req(z, 'z'); mu = _req(mu, 'mu'); logw = _req(logw, 'logw')
    if z.dim() != 3 or z.shape[-1] != D:
        raise RuntimeError('cmvae_amd: z must be [B,N,64]')
    B, N, _ = z.shape
    K = logw.shape[-1]
    if tuple(mu.shape) != (B, K, D) or tuple(logw.shape) != (B, K) or not (1 <= K <= 32):
        raise RuntimeError('cmvae_amd: mixture parameter shape mismatch')
    return B, N, K


def _mixture_fwd(z, mu, logw):
    B, N, K = _mix_shapes(z, mu, logw)
    lse = torch.empty(B, N, device=z.device)
    check(lib.cmvae_mixture_loglik_fwd(_p(z), _p(mu), _p(logw), B, N, K, _p(lse), _stream()),
          'cmvae_mixture_loglik_fwd')
    return lse


def _mixture_bwd(z, mu, logw, g):
    B, N, K = _mix_shapes(z, mu, logw)
    g = _req(g, 'g', shape=(B, N))
    ws_bytes = lib.cmvae_mixture_workspace_bytes(B, N, K)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=z.device)
    dz = torch.empty_like(z)
    dlogw = torch.empty(B, K, device=z.device)
    check(lib.cmvae_mixture_loglik_bwd(_p(z), _p(mu), _p(logw), _p(g), B, N, K, _p(ws), ws_bytes, _p(dz), _p(dlogw),
                                       _stream()), 'cmvae_mixture_loglik_bwd')
    return dz, dlogw


def _reparam_fwd(mean, logvar, eps):
    mean = _req(mean, 'mean'); logvar = _req(logvar, 'logvar'); eps = _req(eps, 'eps')
    if mean.shape[-1] != D or mean.shape != logvar.shape:
        raise RuntimeError('cmvae_amd: mean/logvar must be [...,64] and equal-shaped')
    rows = mean.numel() // D
    if eps.dim() != mean.dim() + 1 or eps.shape[-1] != D or eps.numel() % (rows * D) != 0 or \
            tuple(eps.shape[:-2]) != tuple(mean.shape[:-1]):
        raise RuntimeError('cmvae_amd: eps must be [...,M,64] over mean [...,64]')
    M = eps.shape[-2]
    z = torch.empty_like(eps)
    logq = torch.empty(eps.shape[:-1], device=eps.device)
    check(lib.cmvae_reparam_logq_fwd(_p(mean), _p(logvar), _p(eps), rows, M, _p(z), _p(logq), _stream()),
          'cmvae_reparam_logq_fwd')
    return z, logq


def _reparam_bwd(logvar, eps, gz, glogq):
    logvar = _req(logvar, 'logvar'); eps = _req(eps, 'eps')
    gz = _req(gz, 'gz', shape=eps.shape); glogq = _req(glogq, 'glogq', shape=eps.shape[:-1])
    rows = logvar.numel() // D
    M = eps.shape[-2]
    dmean = torch.empty_like(logvar)
    dlogvar = torch.empty_like(logvar)
    check(lib.cmvae_reparam_logq_bwd(_p(logvar), _p(eps), _p(gz), _p(glogq), rows, M, _p(dmean), _p(dlogvar),
                                     _stream()), 'cmvae_reparam_logq_bwd')
    return dmean, dlogvar


def _expm_trace(A):
    A = _req(A, 'A', shape=(D, D))
    h = torch.empty(1, device=A.device)
    grad = torch.empty_like(A)
    check(lib.cmvae_expm_trace(_p(A), _p(h), _p(grad), _stream()), 'cmvae_expm_trace')
    return h, grad


_deflib.impl('mixture_loglik_fwd', _mixture_fwd, 'CUDA')
_deflib.impl('mixture_loglik_bwd', _mixture_bwd, 'CUDA')
_deflib.impl('reparam_logq_fwd', _reparam_fwd, 'CUDA')
_deflib.impl('reparam_logq_bwd', _reparam_bwd, 'CUDA')
_deflib.impl('expm_trace', _expm_trace, 'CUDA')

_ops = getattr(torch.ops, _NS)


class _MixtureLogLik(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, mu, logw):
        z = z.contiguous(); mu = mu.contiguous(); logw = logw.contiguous()
        ctx.save_for_backward(z, mu, logw)
        return _ops.mixture_loglik_fwd(z, mu, logw)

    @staticmethod
    def backward(ctx, g):
        z, mu, logw = ctx.saved_tensors
        dz, dlogw = _ops.mixture_loglik_bwd(z, mu, logw, g.contiguous())
        return dz, None, dlogw


def mixture_loglik(z, mu, logw):
    """log p(z) = logsumexp_k(glp(z, mu_k) + logw_k); grads flow to z and logw (mu is the detached EM output)."""
    return _MixtureLogLik.apply(z, mu.detach(), logw)


def _center(x):
    B, N, d = x.shape
    if d != D:
        raise RuntimeError('cmvae_amd: center_rows expects [B,N,64]')
    x = x.contiguous()
    out = torch.empty_like(x)
    part = torch.empty(B, lib.cmvae_center_rows_chunks(N), D, device=x.device)
    check(lib.cmvae_center_rows(_p(x), B, N, _p(part), _p(out), _stream()), 'cmvae_center_rows')
    return out


class _CenterRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _center(_req(x, 'x'))

    @staticmethod
    def backward(ctx, g):
        return _center(g)


def center_rows(x):
    """x - x.mean(1, keepdim=True) for x [B,N,64] (omniglot/model.py:449)."""
    return _CenterRows.apply(x)


class _ReparamLogQ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, logvar, eps):
        mean = mean.contiguous(); logvar = logvar.contiguous(); eps = eps.contiguous()
        z, logq = _ops.reparam_logq_fwd(mean, logvar, eps)
        ctx.save_for_backward(logvar, eps)
        return z, logq

    @staticmethod
    def backward(ctx, gz, glogq):
        logvar, eps = ctx.saved_tensors
        if gz is None:
            gz = torch.zeros_like(eps)
        if glogq is None:
            glogq = torch.zeros(eps.shape[:-1], device=eps.device)
        dmean, dlogvar = _ops.reparam_logq_bwd(logvar, eps, gz.contiguous(), glogq.contiguous())
        return dmean, dlogvar, None


def reparam_logq(mean, logvar, eps):
    """z = eps*exp(0.5 logvar)+mean broadcast over the MC axis, and log q(z|x)."""
    return _ReparamLogQ.apply(mean, logvar, eps)


class _ExpmTrace(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A):
        h, grad = _ops.expm_trace(A.contiguous())
        ctx.save_for_backward(grad)
        return h.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g


def expm_trace(A):
    """h(A) = tr(expm(A∘A)) - 64 with dh/dA = expm(A∘A)^T ∘ 2A."""
    return _ExpmTrace.apply(A)


class _ElboTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_q, log_p, q_eps, h, A, h_lambda, h_c, a_lambda):
        log_q, log_p, q_eps, A = (_req(t.contiguous(), n) for t, n in
                                  ((log_q, 'log_q_z'), (log_p, 'log_p_z'), (q_eps, 'q_eps'), (A, 'A')))
        dim = q_eps.shape[-1]
        rows = q_eps.numel() // dim
        h1 = h.reshape(1).contiguous()
        part = torch.empty(lib.cmvae_elbo_tail_blocks(), 4, device=q_eps.device, dtype=torch.float32)
        out = torch.empty(2, device=q_eps.device, dtype=torch.float32)
        check(lib.cmvae_elbo_tail_fwd(_p(log_q), log_q.numel(), _p(log_p), log_p.numel(), _p(q_eps), rows, dim, _p(h1),
                                      _p(A), A.numel(), h_lambda, h_c, a_lambda, _p(part), _p(out), _stream()),
              'cmvae_elbo_tail_fwd')
        ctx.consts = (float(h_lambda), float(h_c), float(a_lambda), rows, dim)
        ctx.shapes = (log_q.shape, log_p.shape, h.shape)
        ctx.save_for_backward(q_eps, h1, A)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_kl, g_h):
        q_eps, h1, A = ctx.saved_tensors
        h_lambda, h_c, a_lambda, rows, dim = ctx.consts
        sq, sp, sh = ctx.shapes
        dev = q_eps.device
        parts = [None if t is None else t.reshape(()).float() for t in (g_kl, g_h)]
        if any(t is None for t in parts):                  # an unused output: its gradient counts as zero
            zero = torch.zeros((), device=dev)
            parts = [zero if t is None else t for t in parts]
        g = torch.stack(parts)
        d_q = torch.empty(sq, device=dev, dtype=torch.float32)
        d_p = torch.empty(sp, device=dev, dtype=torch.float32)
        d_e = torch.empty_like(q_eps)
        dA = torch.empty_like(A)
        dh = torch.empty(1, device=dev, dtype=torch.float32)
        check(lib.cmvae_elbo_tail_bwd(_p(g), _p(h1), _p(q_eps), rows, dim, _p(A), A.numel(), d_q.numel(), d_p.numel(),
                                      h_lambda, h_c, a_lambda, _p(d_q), _p(d_p), _p(d_e), _p(dA), _p(dh), _stream()),
              'cmvae_elbo_tail_bwd')
        return d_q, d_p, d_e, dh.reshape(sh), dA, None, None, None


def elbo_tail(log_q_z, log_p_z, q_eps, h, A, h_lambda, h_c, a_lambda):
    """(kl_loss, h_augment) of GMVAE.forward (omniglot/model.py:496-516):
    kl = mean(log_q_z) - mean(log_p_z) - mean(glp(q_eps, 0)), h_augment = h_lambda h + 0.5 h_c h^2 + a_lambda |A|_1,
    two launches forward and one backward instead of the chain of scalar ATen kernels."""
    return _ElboTail.apply(log_q_z, log_p_z, q_eps, h, A, float(h_lambda), float(h_c), float(a_lambda))


class _PriorLogW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pi, trace, x):
        pi, trace, x = _req(pi.contiguous(), 'pi_k'), _req(trace.contiguous(), 'trace_k'), _req(x.contiguous(), 'x')
        if x.shape[-1] != D or pi.shape != x.shape[:-1] or trace.shape != pi.shape:
            raise RuntimeError('cmvae_amd: prior_logw takes pi, trace [B,K] and x [B,K,64]')
        logw = torch.empty_like(pi)
        check(lib.cmvae_prior_logw_fwd(_p(pi), _p(trace), _p(x), pi.numel(), _p(logw), _stream()),
              'cmvae_prior_logw_fwd')
        ctx.save_for_backward(trace, x)
        return logw

    @staticmethod
    def backward(ctx, g):
        trace, x = ctx.saved_tensors
        dx = torch.empty_like(x)
        check(lib.cmvae_prior_logw_bwd(_p(trace), _p(x), _p(g.contiguous()), trace.numel(), _p(dx), _stream()),
              'cmvae_prior_logw_bwd')
        return None, None, dx


def prior_logw(pi_k, trace_k, q_eps_mu_k):
    """log(pi_k) + gaussian_log_prob(q_eps_mu_k, mean=log(trace_k)) (omniglot/model.py:481-491 — log(trace_k) in the
    MEAN slot, as the reference has it); the gradient goes to q_eps_mu_k only (pi_k, trace_k: detached EM outputs)."""
    return _PriorLogW.apply(pi_k.detach(), trace_k.detach(), q_eps_mu_k)
