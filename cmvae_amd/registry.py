"""`torch.ops.cmvae_amd.*` for every leaf entry point of the C-ABI that takes tensors: the binding a PyTorch maintainer
of the reference would call.  `cmvae_amd/ops.py` registers the five ops that carry their own
autograd formulas; this module registers the rest on the same library, each as a thin adapter over the host function
the product path itself calls (one implementation, two ways in: the autograd Functions of linear.py / conv.py /
encoder.py / decoder.py call the host functions directly and skip the dispatcher's per-call cost).

Struct arguments travel as flat lists: `int[] geom` is `geom_to_list(ConvGeom)`; `Tensor[] causal` is `[A]` for the
linear causal layer or `[fc1.weight, fc1.bias, fc2.0.weight, fc2.0.bias]` for the NOTEARS-MLP one.  Optional outputs
come back as empty tensors.
"""
import ctypes

import torch

from . import conv, linear, ops
from ._lib import lib, check, ConvGeom

_d = ops._deflib
_p = ops._p
_GEOM_FIELDS = [(n, t) for n, t in ConvGeom._fields_]


def geom_to_list(g):
    """ConvGeom (ctypes) -> flat int list, field order of `struct cmvae_conv_geom` (include/cmvae_abi.h)."""
    out = []
    for name, typ in _GEOM_FIELDS:
        v = getattr(g, name)
        out.extend(int(x) for x in v) if hasattr(v, '__len__') else out.append(int(v))
    return out


_GEOM_LEN = sum(getattr(t, '_length_', 0) or 1 for _, t in _GEOM_FIELDS)


def geom_from_list(vals):
    if len(vals) != _GEOM_LEN:
        raise RuntimeError('cmvae_amd: geometry list has %d entries, the struct takes %d' % (len(vals), _GEOM_LEN))
    g = ConvGeom()
    i = 0
    for name, typ in _GEOM_FIELDS:
        n = getattr(typ, '_length_', 0)
        if n:
            arr = getattr(g, name)
            for k in range(n):
                arr[k] = int(vals[i + k])
            i += n
        else:
            setattr(g, name, int(vals[i]))
            i += 1
    return g


def _causal(ts):
    if len(ts) == 1:
        return ops.Causal(True, ts[0])
    if len(ts) == 4:
        return ops.Causal(False, *ts)
    raise RuntimeError('cmvae_amd: causal takes [A] or [fc1.weight, fc1.bias, fc2.0.weight, fc2.0.bias]')


def _empty(ref):
    return torch.empty(0, device=ref.device, dtype=torch.float32)


# ----------------------------------------------------------------------------- EM / prediction
_d.define('unsup_prior(Tensor z, Tensor init_idx, Tensor[] causal, int iters, bool fix_pi, float gamma1)'
          ' -> (Tensor, Tensor, Tensor, Tensor)')
_d.define('semi_prior(Tensor uz, Tensor sz, Tensor y, int K, Tensor[] causal, int iters, float gamma1)'
          ' -> (Tensor, Tensor, Tensor, Tensor, Tensor)')
_d.define('posterior_argmax(Tensor z, Tensor mu, Tensor logvar, Tensor logw, int te, int M, bool want_post)'
          ' -> (Tensor, Tensor)')
_d.define('causal_rows(Tensor x, Tensor[] causal, Tensor? pi, Tensor? tau) -> (Tensor, Tensor)')
_d.define('gram_inverse(Tensor[] causal, Tensor? logvar, float s0, int nmat) -> Tensor')
_d.define('center_rows(Tensor x) -> Tensor')


def _unsup_prior(z, init_idx, causal, iters, fix_pi, gamma1):
    return ops.unsup_prior(z, init_idx, _causal(causal), iters, fix_pi, gamma1)


def _semi_prior(uz, sz, y, K, causal, iters, gamma1):
    return ops.semi_prior(uz, sz, y, K, _causal(causal), iters, gamma1)


def _posterior_argmax(z, mu, logvar, logw, te, M, want_post):
    y, post = ops.posterior_argmax(z, mu, logvar, logw, te, M, want_post)
    return y, (post if post is not None else _empty(z))


def _causal_rows(x, causal, pi, tau):
    eps, logw = ops.causal_rows(x, _causal(causal), pi, tau)
    return eps, (logw if logw is not None else _empty(x))


def _gram_inverse(causal, logvar, s0, nmat):
    return ops.gram_inverse(_causal(causal), logvar, s0, nmat)


# ----------------------------------------------------------------------------- Linear / attention / Adam
_d.define('linear_fwd(Tensor x, Tensor weight, Tensor? bias, int act) -> Tensor')
_d.define('linear_bwd(Tensor x, Tensor weight, Tensor dz, bool need_dx, bool need_db) -> (Tensor, Tensor, Tensor)')
_d.define('linear_mse_fwd(Tensor x, Tensor weight, Tensor? bias, int act, Tensor target, int M) -> (Tensor, Tensor)')
_d.define('mse_act_bwd(Tensor y, Tensor target, int M, int act, Tensor? g, float gmul) -> Tensor')
_d.define('act_bwd(Tensor dy, Tensor y, int act) -> Tensor')
_d.define('attention_fwd(Tensor q, Tensor k, Tensor v, int heads, float scale) -> (Tensor, Tensor)')
_d.define('attention_bwd(Tensor q, Tensor k, Tensor v, Tensor lse, Tensor do, int heads, float scale)'
          ' -> (Tensor, Tensor, Tensor)')
_d.define('attention_qkv_fwd(Tensor qkv, int heads, float scale) -> (Tensor, Tensor)')
_d.define('attention_qkv_bwd(Tensor qkv, Tensor lse, Tensor do, int heads, float scale) -> Tensor')
_d.define('adam_step(Tensor(a!) param, Tensor grad, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, int step, float lr, '
          'float beta1, float beta2, float eps, float weight_decay) -> ()')


def _rows2(x, name):
    linear._need(x, name)
    if x.dim() != 2 or x.stride(1) != 1 or x.stride(0) % 4:
        raise RuntimeError('cmvae_amd: %s must be a [rows, features] matrix with unit column stride' % name)
    return x


def _linear_fwd(x, weight, bias, act):
    linear._need(weight, 'weight')
    return linear.linear_fwd(_rows2(x, 'x'), weight.contiguous(), None if bias is None else bias.contiguous(), act)


def _linear_bwd(x, weight, dz, need_dx, need_db):
    dx, dW, db = linear.linear_bwd(_rows2(x, 'x'), weight.contiguous(), _rows2(dz, 'dz'), need_dx, need_db)
    return (dx if dx is not None else _empty(x)), dW, (db if db is not None else _empty(x))


def _linear_mse_fwd(x, weight, bias, act, target, M):
    x = _rows2(x, 'x')
    linear._need(weight, 'weight'); linear._need(target, 'target')
    rows, out_f = x.shape[0], weight.shape[0]
    t2 = target.reshape(-1, out_f).contiguous()
    if rows == 0 or rows != t2.shape[0] * M:
        raise RuntimeError('cmvae_amd: linear_mse_fwd: %d rows against %d targets x %d samples' % (rows, t2.shape[0], M))
    y = torch.empty(rows, out_f, device=x.device, dtype=torch.float32)
    part = torch.empty(lib.cmvae_linear_mse_partials(rows, out_f), 128, device=x.device, dtype=torch.float32)
    check(lib.cmvae_linear_mse_fwd(_p(x), x.stride(0), rows, x.shape[1], out_f, _p(weight.contiguous()),
                                   None if bias is None else _p(bias.contiguous()), act, _p(t2), M, _p(y), y.stride(0),
                                   _p(part), ops._stream()), 'cmvae_linear_mse_fwd')
    return y, part.sum()


def _mse_act_bwd(y, target, M, act, g, gmul):
    y = _rows2(y, 'y')
    t2 = target.reshape(-1, y.shape[1]).contiguous()
    dz = torch.empty_like(y)
    check(lib.cmvae_mse_act_bwd(_p(y), _p(t2), y.shape[0], y.shape[1], M, act,
                                None if g is None else _p(g.reshape(1).contiguous()), float(gmul), _p(dz),
                                ops._stream()), 'cmvae_mse_act_bwd')
    return dz


def _act_bwd(dy, y, act):
    return linear.act_bwd(dy.contiguous(), y.contiguous(), act)


def _attention_fwd(q, k, v, heads, scale):
    for t, n in ((q, 'q'), (k, 'k'), (v, 'v')):
        linear._need(t, n)
    B, S, C = q.shape
    if C != 64 * heads or k.shape != q.shape or v.shape != q.shape:
        raise RuntimeError('cmvae_amd attention: heads of 64 channels and equal Q/K/V shapes are required')
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    o = torch.empty_like(q)
    lse = torch.empty(B, heads, S, device=q.device, dtype=torch.float32)
    check(lib.cmvae_attention_fwd(_p(q), _p(k), _p(v), B, S, heads, scale, _p(o), _p(lse), ops._stream()),
          'cmvae_attention_fwd')
    return o, lse


def _attention_bwd(q, k, v, lse, do, heads, scale):
    B, S, _ = q.shape
    q, k, v, do, lse = q.contiguous(), k.contiguous(), v.contiguous(), do.contiguous(), lse.contiguous()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    delta = torch.empty_like(lse)
    check(lib.cmvae_attention_bwd(_p(q), _p(k), _p(v), _p(lse), _p(do), B, S, heads, scale, _p(dq), _p(dk), _p(dv),
                                  _p(delta), ops._stream()), 'cmvae_attention_bwd')
    return dq, dk, dv


def _qkv3(qkv, heads):
    linear._need(qkv, 'qkv')
    if qkv.dim() != 3 or qkv.shape[2] != 192 * heads:
        raise RuntimeError('cmvae_amd attention: the packed projection is [B, S, 3 * 64 * heads], got %s for %d heads'
                           % (tuple(qkv.shape), heads))
    return qkv.contiguous()


def _attention_qkv_fwd(qkv, heads, scale):
    qkv = _qkv3(qkv, heads)
    B, S, _ = qkv.shape
    o = torch.empty(B, S, 64 * heads, device=qkv.device, dtype=torch.float32)
    lse = torch.empty(B, heads, S, device=qkv.device, dtype=torch.float32)
    check(lib.cmvae_attention_qkv_fwd(_p(qkv), B, S, heads, scale, _p(o), _p(lse), ops._stream()),
          'cmvae_attention_qkv_fwd')
    return o, lse


def _attention_qkv_bwd(qkv, lse, do, heads, scale):
    qkv = _qkv3(qkv, heads)
    B, S, _ = qkv.shape
    do, lse = do.contiguous(), lse.contiguous()
    if tuple(do.shape) != (B, S, 64 * heads) or tuple(lse.shape) != (B, heads, S):
        raise RuntimeError('cmvae_amd attention: do / lse do not match the packed projection')
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    check(lib.cmvae_attention_qkv_bwd(_p(qkv), _p(lse), _p(do), B, S, heads, scale, _p(dqkv), _p(delta),
                                      ops._stream()), 'cmvae_attention_qkv_bwd')
    return dqkv


def _adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay):
    for t, n in ((param, 'param'), (grad, 'grad'), (exp_avg, 'exp_avg'), (exp_avg_sq, 'exp_avg_sq')):
        ops._req(t, n)
        if t.numel() != param.numel() or t.numel() % 4:
            raise RuntimeError('cmvae_amd: adam_step takes four flat buffers of one length, a multiple of 4')
    check(lib.cmvae_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), step, lr, beta1, beta2,
                              eps, weight_decay, ops._stream()), 'cmvae_adam_step')


# ----------------------------------------------------------------------------- scalar tail of forward
_d.define('prior_logw_fwd(Tensor pi, Tensor trace, Tensor x) -> Tensor')
_d.define('prior_logw_bwd(Tensor trace, Tensor x, Tensor g) -> Tensor')
_d.define('elbo_tail_fwd(Tensor log_q, Tensor log_p, Tensor q_eps, Tensor h, Tensor A, float h_lambda, float h_c, '
          'float a_lambda) -> Tensor')
_d.define('elbo_tail_bwd(Tensor g, Tensor h, Tensor q_eps, Tensor A, int n1, int n2, float h_lambda, float h_c, '
          'float a_lambda) -> (Tensor, Tensor, Tensor, Tensor, Tensor)')


def _gpu32(t, name):
    t = t.contiguous()
    linear._need(t, name)
    return t


def _prior_logw_fwd(pi, trace, x):
    pi, trace, x = (_gpu32(t, n) for t, n in ((pi, 'pi'), (trace, 'trace'), (x, 'x')))
    if x.shape[-1] != 64 or pi.numel() * 64 != x.numel() or trace.numel() != pi.numel():
        raise RuntimeError('cmvae_amd: prior_logw takes pi, trace [B,K] and x [B,K,64]')
    logw = torch.empty_like(pi)
    check(lib.cmvae_prior_logw_fwd(_p(pi), _p(trace), _p(x), pi.numel(), _p(logw), ops._stream()),
          'cmvae_prior_logw_fwd')
    return logw


def _prior_logw_bwd(trace, x, g):
    trace, x, g = (_gpu32(t, n) for t, n in ((trace, 'trace'), (x, 'x'), (g, 'g')))
    if x.shape[-1] != 64 or trace.numel() * 64 != x.numel() or g.numel() != trace.numel():
        raise RuntimeError('cmvae_amd: prior_logw_bwd takes trace, g [B,K] and x [B,K,64]')
    dx = torch.empty_like(x)
    check(lib.cmvae_prior_logw_bwd(_p(trace), _p(x), _p(g), trace.numel(), _p(dx), ops._stream()),
          'cmvae_prior_logw_bwd')
    return dx


def _tail_args(q_eps, h, A):
    q_eps, A = _gpu32(q_eps, 'q_eps'), _gpu32(A, 'A')
    h1 = _gpu32(h.reshape(1), 'h')
    dim = q_eps.shape[-1]
    if q_eps.numel() == 0 or dim % 4:
        raise RuntimeError('cmvae_amd: elbo_tail takes a non-empty q_eps [.., D] with D a multiple of 4')
    return q_eps, h1, A, q_eps.numel() // dim, dim


def _elbo_tail_fwd(log_q, log_p, q_eps, h, A, h_lambda, h_c, a_lambda):
    log_q, log_p = _gpu32(log_q, 'log_q'), _gpu32(log_p, 'log_p')
    q_eps, h1, A, rows, dim = _tail_args(q_eps, h, A)
    if log_q.numel() == 0 or log_p.numel() == 0:
        raise RuntimeError('cmvae_amd: elbo_tail takes non-empty log_q / log_p')
    part = torch.empty(lib.cmvae_elbo_tail_blocks(), 4, device=q_eps.device, dtype=torch.float32)
    out = torch.empty(2, device=q_eps.device, dtype=torch.float32)
    check(lib.cmvae_elbo_tail_fwd(_p(log_q), log_q.numel(), _p(log_p), log_p.numel(), _p(q_eps), rows, dim, _p(h1),
                                  _p(A), A.numel(), h_lambda, h_c, a_lambda, _p(part), _p(out), ops._stream()),
          'cmvae_elbo_tail_fwd')
    return out


def _elbo_tail_bwd(g, h, q_eps, A, n1, n2, h_lambda, h_c, a_lambda):
    g = _gpu32(g, 'g')
    q_eps, h1, A, rows, dim = _tail_args(q_eps, h, A)
    if g.numel() != 2 or n1 < 1 or n2 < 1:
        raise RuntimeError('cmvae_amd: elbo_tail_bwd takes g = {d kl, d h_aug} and the element counts of log_q / log_p')
    dev = q_eps.device
    d_q, d_p = torch.empty(n1, device=dev), torch.empty(n2, device=dev)
    d_e, dA, dh = torch.empty_like(q_eps), torch.empty_like(A), torch.empty(1, device=dev)
    check(lib.cmvae_elbo_tail_bwd(_p(g), _p(h1), _p(q_eps), rows, dim, _p(A), A.numel(), n1, n2, h_lambda, h_c,
                                  a_lambda, _p(d_q), _p(d_p), _p(d_e), _p(dA), _p(dh), ops._stream()),
          'cmvae_elbo_tail_bwd')
    return d_q, d_p, d_e, dA, dh


# ----------------------------------------------------------------------------- gather-GEMM family
_d.define('conv_gemm(int[] geom, Tensor A, Tensor Wt, Tensor(a!) out, Tensor? bias, Tensor? a_scale, Tensor? a_shift, '
          'int act, int n_col_blocks, bool stats) -> Tensor')
_d.define('conv_wgrad(int[] geom, Tensor A, Tensor G, Tensor(a!) dW, int[] tap_off, int s_ci, int s_co, '
          'Tensor(b!)? db, bool accumulate_db, Tensor? a_scale, Tensor? a_shift, int g_pitch, int g_col) -> ()')
_d.define('conv_gemm_multi(int[] geoms, Tensor A, Tensor Wt, Tensor(a!) out, Tensor? bias, Tensor? a_scale, '
          'Tensor? a_shift, bool stats) -> Tensor')
_d.define('conv_wgrad_multi(int[] geoms, Tensor A, Tensor G, Tensor(a!) dW, int[] tap_off, int s_ci, int s_co, '
          'Tensor(b!)? db, Tensor? a_scale, Tensor? a_shift, int g_pitch) -> ()')
_d.define('conv_gemm_bnbwd(int[] geom, Tensor A, Tensor Wt, Tensor(a!) out, Tensor x, Tensor scale, Tensor shift, '
          'Tensor mean, Tensor invstd) -> Tensor')


def _conv_gemm_bnbwd(geom, A, Wt, out, x, scale, shift, mean, invstd):
    for t, n in ((A, 'A'), (Wt, 'Wt'), (out, 'out'), (x, 'x')):
        linear._need(t, n)
    if x.shape != out.shape or not x.is_contiguous() or any(t.numel() != 64 for t in (scale, shift, mean, invstd)):
        raise RuntimeError('cmvae_amd: conv_gemm_bnbwd takes x in the layout of out and four [64] fold vectors')
    return conv.conv_gemm_bnbwd(geom_from_list(geom), A, Wt, out, x, (scale, shift, mean, invstd))


def _geoms_from_list(vals):
    """1..4 geometries, one int list after the other."""
    if len(vals) == 0 or len(vals) % _GEOM_LEN or len(vals) > 4 * _GEOM_LEN:
        raise RuntimeError('cmvae_amd: %d entries are not 1..4 geometry lists of %d' % (len(vals), _GEOM_LEN))
    return [geom_from_list(vals[i:i + _GEOM_LEN]) for i in range(0, len(vals), _GEOM_LEN)]


def _conv_gemm_multi(geoms, A, Wt, out, bias, a_scale, a_shift, stats):
    for t, n in ((A, 'A'), (Wt, 'Wt'), (out, 'out')):
        linear._need(t, n)
    part = conv.conv_gemm_multi(_geoms_from_list(geoms), A, Wt, out, bias=bias, a_scale=a_scale, a_shift=a_shift,
                                stats=stats)
    return part if part is not None else _empty(A)


def _conv_wgrad_multi(geoms, A, G, dW, tap_off, s_ci, s_co, db, a_scale, a_shift, g_pitch):
    for t, n in ((A, 'A'), (G, 'G'), (dW, 'dW')):
        linear._need(t, n)
    gs = _geoms_from_list(geoms)
    if len(tap_off) != sum(g.ntaps for g in gs):
        raise RuntimeError('cmvae_amd: conv_wgrad_multi takes one tap offset per tap of every geometry')
    offs, i = [], 0
    for g in gs:
        offs.append(list(tap_off[i:i + g.ntaps]))
        i += g.ntaps
    conv.conv_wgrad_multi(gs, A, G, dW, offs, s_ci, s_co, db=db, a_scale=a_scale, a_shift=a_shift, g_pitch=g_pitch)


def _conv_gemm(geom, A, Wt, out, bias, a_scale, a_shift, act, n_col_blocks, stats):
    for t, n in ((A, 'A'), (Wt, 'Wt'), (out, 'out')):
        linear._need(t, n)
    part = conv.conv_gemm(geom_from_list(geom), A, Wt, out, bias=bias, a_scale=a_scale, a_shift=a_shift, act=act,
                          n_col_blocks=n_col_blocks, stats=stats)
    return part if part is not None else _empty(A)


def _conv_wgrad(geom, A, G, dW, tap_off, s_ci, s_co, db, accumulate_db, a_scale, a_shift, g_pitch, g_col):
    for t, n in ((A, 'A'), (G, 'G'), (dW, 'dW')):
        linear._need(t, n)
    conv.conv_wgrad(geom_from_list(geom), A, G, dW, list(tap_off), s_ci, s_co, db=db, accumulate_db=accumulate_db,
                    a_scale=a_scale, a_shift=a_shift, g_pitch=g_pitch, g_col=g_col)


for _name, _fn in (('unsup_prior', _unsup_prior), ('semi_prior', _semi_prior), ('posterior_argmax', _posterior_argmax),
                   ('causal_rows', _causal_rows), ('gram_inverse', _gram_inverse), ('center_rows', ops._center),
                   ('linear_fwd', _linear_fwd), ('linear_bwd', _linear_bwd), ('linear_mse_fwd', _linear_mse_fwd),
                   ('mse_act_bwd', _mse_act_bwd), ('act_bwd', _act_bwd), ('attention_fwd', _attention_fwd),
                   ('attention_bwd', _attention_bwd), ('adam_step', _adam_step), ('conv_gemm', _conv_gemm),
                   ('attention_qkv_fwd', _attention_qkv_fwd), ('attention_qkv_bwd', _attention_qkv_bwd),
                   ('conv_wgrad', _conv_wgrad), ('conv_gemm_multi', _conv_gemm_multi),
                   ('conv_gemm_bnbwd', _conv_gemm_bnbwd), ('prior_logw_fwd', _prior_logw_fwd),
                   ('prior_logw_bwd', _prior_logw_bwd), ('elbo_tail_fwd', _elbo_tail_fwd),
                   ('elbo_tail_bwd', _elbo_tail_bwd),
                   ('conv_wgrad_multi', _conv_wgrad_multi)):
    _d.impl(_name, _fn, 'CUDA')

REGISTERED = ('mixture_loglik_fwd', 'mixture_loglik_bwd', 'reparam_logq_fwd', 'reparam_logq_bwd', 'expm_trace',
              'unsup_prior', 'semi_prior', 'posterior_argmax', 'causal_rows', 'gram_inverse', 'center_rows',
              'linear_fwd', 'linear_bwd', 'linear_mse_fwd', 'mse_act_bwd', 'act_bwd', 'attention_fwd', 'attention_bwd',
              'adam_step', 'conv_gemm', 'conv_wgrad', 'conv_gemm_multi', 'conv_wgrad_multi', 'attention_qkv_fwd',
              'attention_qkv_bwd', 'conv_gemm_bnbwd', 'prior_logw_fwd', 'prior_logw_bwd', 'elbo_tail_fwd', 'elbo_tail_bwd')


# ----------------------------------------------------------------------------- whole encoder / decoder
# `encode_images` and `decoder_bce` are compositions of the leaf kernels with hand-written backward passes
# (encoder.py / decoder.py); registered as composite ops, autograd flows through their autograd.Functions.
# BatchNorm state travels as tensors: params = [conv.weight, conv.bias, bn.weight, bn.bias] per layer (the decoder's
# last layer has no BatchNorm: [weight, bias]); running = [running_mean, running_var, num_batches_tracked] per BatchNorm.
class _BNState:
    """The attributes of nn.BatchNorm2d the HIP path reads, over tensors handed in through the dispatcher."""
    track_running_stats = True

    def __init__(self, weight, bias, running_mean, running_var, num_batches_tracked, eps, momentum):
        self.weight, self.bias = weight, bias
        self.running_mean, self.running_var, self.num_batches_tracked = running_mean, running_var, num_batches_tracked
        self.eps, self.momentum = eps, (None if momentum < 0 else momentum)


_d.define('encode_images(Tensor x, bool training, Tensor[] params, Tensor(a!)[] running, float eps, float momentum)'
          ' -> Tensor')
_d.define('decoder_bce(Tensor h_rec, Tensor target, int M, bool training, Tensor[] params, Tensor(a!)[] running, '
          'float eps, float momentum) -> (Tensor, Tensor)')


def _bn_states(params, running, n_bn, eps, momentum):
    if len(params) < 4 * n_bn or len(running) != 3 * n_bn:
        raise RuntimeError('cmvae_amd: %d BatchNorm layers take 4 parameter tensors and 3 running tensors each' % n_bn)
    return tuple(_BNState(params[4 * l + 2], params[4 * l + 3], running[3 * l], running[3 * l + 1], running[3 * l + 2],
                          eps, momentum) for l in range(n_bn))


def _encode_images(x, training, params, running, eps, momentum):
    from .encoder import _Encoder
    if len(params) != 16:
        raise RuntimeError('cmvae_amd: encode_images takes 16 parameter tensors (4 layers x conv.weight, conv.bias, '
                           'bn.weight, bn.bias)')
    return _Encoder.apply(x, training, _bn_states(params, running, 4, eps, momentum), *params)


def _decoder_bce(h_rec, target, M, training, params, running, eps, momentum):
    from .decoder import _DecoderBCE
    if len(params) != 14:
        raise RuntimeError('cmvae_amd: decoder_bce takes 14 parameter tensors (3 layers x 4, then the last '
                           "layer's weight and bias)")
    return _DecoderBCE.apply(h_rec.contiguous(), target, M, training, _bn_states(params, running, 3, eps, momentum),
                             *params)


_d.impl('encode_images', _encode_images, 'CompositeImplicitAutograd')
_d.impl('decoder_bce', _decoder_bce, 'CompositeImplicitAutograd')
REGISTERED = REGISTERED + ('encode_images', 'decoder_bce')
