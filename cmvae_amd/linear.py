"""nn.Linear (+ReLU/ELU), the bulk NOTEARS-MLP causal layer and Adam on the HIP library.

  * `linear(x, weight, bias, act)`  -- proj (omniglot/model.py:70-77; mimgnet/model.py:59-66), the
    Set-Transformer projections and the (mu, logvar) head (omniglot/module.py:80-86, model.py:65-68):
    forward, data gradient and weight gradient all run on the fp32-MFMA gather-GEMM kernels.
  * `causal_mlp(x, fc1_w, fc1_b, w2, b2)` -- CausalEncoder.forward on bulk rows, nonlinear variant
    (omniglot/module.py:165-188, 232-288): eps = x - LocallyConnected(elu(fc1(x))).
  * `FlatAdam` -- torch.optim.Adam(lr, weight_decay) as one kernel over a flat parameter buffer
    (omniglot/trainer.py:48-52).
"""
import torch

from . import _workspace
from ._lib import lib, check

ACT = {None: 0, 'none': 0, 'relu': 1, 'elu': 2}
_p = lambda t: None if t is None else t.data_ptr()  # noqa: E731


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need(t, name):
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError('cmvae_amd.linear: %s must be a float32 GPU tensor (there is no CPU path)' % name)


TAPS = None          # tests set this to a dict: receives the output of every Linear+ReLU of a forward, in call order


def linear_fwd(x2, weight, bias, act, out=None):
    rows, in_f = x2.shape
    out_f = weight.shape[0]
    y = torch.empty(rows, out_f, device=x2.device, dtype=torch.float32) if out is None else out
    check(lib.cmvae_linear_fwd(_p(x2), x2.stride(0), rows, in_f, out_f, _p(weight), _p(bias), act, _p(y),
                               y.stride(0), _stream()), 'cmvae_linear_fwd')
    return y


def act_bwd(dy, y, act):
    dz = torch.empty_like(y)
    check(lib.cmvae_act_bwd(_p(dy), _p(y), y.numel(), act, _p(dz), _stream()), 'cmvae_act_bwd')
    return dz


def linear_bwd(x2, weight, dz, need_dx, need_db):
    rows, in_f = x2.shape
    out_f = weight.shape[0]
    dx = torch.empty(rows, in_f, device=x2.device, dtype=torch.float32) if need_dx else None
    dW = torch.empty_like(weight)
    db = torch.empty(out_f, device=x2.device, dtype=torch.float32) if need_db else None
    nbytes = lib.cmvae_linear_bwd_workspace_bytes(rows, in_f, out_f)
    ws = _workspace.get(nbytes, x2.device, 'linear_bwd')
    check(lib.cmvae_linear_bwd(_p(x2), x2.stride(0), _p(weight), _p(dz), dz.stride(0), rows, in_f, out_f, _p(dx),
                               in_f, _p(dW), _p(db), _p(ws), ws.numel(), _stream()), 'cmvae_linear_bwd')
    return dx, dW, db


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        _need(x, 'x'); _need(weight, 'weight')
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1 or x2.stride(0) % 4:
            x2 = x2.contiguous()
        w = weight.contiguous()
        y = linear_fwd(x2, w, None if bias is None else bias.contiguous(), act)
        if TAPS is not None and act == 1:
            TAPS.setdefault('linear_relu', []).append(y)
        ctx.act, ctx.shape, ctx.has_bias = act, x.shape, bias is not None
        ctx.save_for_backward(x2, w, y if act else None)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        dy2 = dy.reshape(-1, w.shape[0]).contiguous()
        if x2.shape[0] == 0:
            return torch.zeros(ctx.shape, device=dy.device), torch.zeros_like(w), \
                (torch.zeros(w.shape[0], device=dy.device) if ctx.has_bias else None), None
        dz = act_bwd(dy2, y, ctx.act) if ctx.act else dy2
        dx, dW, db = linear_bwd(x2, w, dz, ctx.needs_input_grad[0], ctx.has_bias)
        return (None if dx is None else dx.view(ctx.shape)), dW, db, None


def linear(x, weight, bias=None, act=None):
    """act(x @ weight^T + bias) on the HIP GEMM kernels; features must be multiples of 64 (<= 2048)."""
    return _Linear.apply(x, weight, bias, ACT[act])


class _LinearMSE(torch.autograd.Function):
    """y = act(x W^T + b) and sum((y - target[row // M])^2) from the GEMM epilogue (cmvae_linear_mse_fwd)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, target, M):
        _need(x, 'x'); _need(weight, 'weight'); _need(target, 'target')
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1 or x2.stride(0) % 4:
            x2 = x2.contiguous()
        w = weight.contiguous()
        rows, out_f = x2.shape[0], w.shape[0]
        t2 = target.reshape(-1, out_f).contiguous()
        if rows == 0 or rows != t2.shape[0] * M:
            raise RuntimeError('cmvae_amd.linear_mse: %d rows against %d targets x %d samples' % (rows, t2.shape[0], M))
        y = torch.empty(rows, out_f, device=x2.device, dtype=torch.float32)
        part = torch.empty(lib.cmvae_linear_mse_partials(rows, out_f), 128, device=x2.device, dtype=torch.float32)
        check(lib.cmvae_linear_mse_fwd(_p(x2), x2.stride(0), rows, x2.shape[1], out_f, _p(w),
                                       None if bias is None else _p(bias.contiguous()), act, _p(t2), M, _p(y),
                                       y.stride(0), _p(part), _stream()), 'cmvae_linear_mse_fwd')
        if TAPS is not None and act == 1:
            TAPS.setdefault('linear_relu', []).append(y)
        ctx.act, ctx.shape, ctx.has_bias, ctx.M = act, x.shape, bias is not None, M
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x2, w, y, t2)
        return y.view(*x.shape[:-1], out_f), part.sum()

    @staticmethod
    def backward(ctx, dy, dloss):
        x2, w, y, t2 = ctx.saved_tensors
        dz = None
        if dloss is not None:
            dz = torch.empty_like(y)
            check(lib.cmvae_mse_act_bwd(_p(y), _p(t2), y.shape[0], y.shape[1], ctx.M, ctx.act,
                                        _p(dloss.reshape(1).contiguous()), 1.0, _p(dz), _stream()), 'cmvae_mse_act_bwd')
        if dy is not None:
            dy2 = dy.reshape(-1, w.shape[0]).contiguous()
            d2 = act_bwd(dy2, y, ctx.act) if ctx.act else dy2
            dz = d2 if dz is None else dz + d2
        if dz is None:
            dz = torch.zeros_like(y)
        dx, dW, db = linear_bwd(x2, w, dz, ctx.needs_input_grad[0], ctx.has_bias)
        return (None if dx is None else dx.view(ctx.shape)), dW, db, None, None, None


def linear_mse(x, weight, bias, act, target, M):
    """(act(x W^T + b), F.mse_loss(., target repeated M times per row, reduction='sum')) — the last proj layer of the
    feature models with their Gaussian likelihood (mimgnet/model.py:59-66, 388-391) in one GEMM launch; the target is
    not differentiated (it is the input features)."""
    return _LinearMSE.apply(x, weight, bias, ACT[act], target.detach(), int(M))


class _CausalMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fc1_w, fc1_b, w2, b2):
        _need(x, 'x')
        x2 = x.reshape(-1, 64).contiguous()
        w1, b1 = fc1_w.contiguous(), fc1_b.contiguous()
        w2c, b2c = w2.reshape(64, 32).contiguous(), b2.reshape(64).contiguous()
        hid = linear_fwd(x2, w1, b1, 2)                                   # [rows, 2048] = elu(fc1(x))
        eps = torch.empty_like(x2)
        check(lib.cmvae_causal_lc_fwd(_p(x2), _p(hid), _p(w2c), _p(b2c), x2.shape[0], _p(eps), _stream()),
              'cmvae_causal_lc_fwd')
        ctx.shape, ctx.w2_shape, ctx.b2_shape = x.shape, w2.shape, b2.shape
        ctx.save_for_backward(x2, w1, w2c, hid)
        return eps.view(x.shape)

    @staticmethod
    def backward(ctx, d_eps):
        x2, w1, w2c, hid = ctx.saved_tensors
        rows = x2.shape[0]
        g = d_eps.reshape(-1, 64).contiguous()
        d_pre = torch.empty_like(hid)
        nblk, L = lib.cmvae_causal_lc_bwd_grid(rows), lib.cmvae_causal_lc_bwd_partial_len()
        part = torch.empty(nblk, L, device=g.device, dtype=torch.float32)
        check(lib.cmvae_causal_lc_bwd(_p(g), _p(hid), _p(w2c), rows, _p(d_pre), _p(part), _stream()),
              'cmvae_causal_lc_bwd')
        red = torch.empty(L, device=g.device, dtype=torch.float32)
        check(lib.cmvae_colsum_reduce(_p(part), nblk, L, _p(red), _stream()), 'cmvae_colsum_reduce')
        dx1, dW1, _ = linear_bwd(x2, w1, d_pre, ctx.needs_input_grad[0], False)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (g + dx1).view(ctx.shape)
        return dx, dW1, red[2112:], red[:2048].view(ctx.w2_shape), red[2048:2112].view(ctx.b2_shape)


def causal_mlp(x, fc1_w, fc1_b, w2, b2):
    """eps = x - LocallyConnected(elu(fc1(x))) for x [..., 64]; w2 [64,32,1], b2 [64,1]."""
    if x.numel() == 0:
        return x.clone()
    return _CausalMLP.apply(x, fc1_w, fc1_b, w2, b2)


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        for t, n in ((q, 'q'), (k, 'k'), (v, 'v')):
            _need(t, n)
        B, S, C = q.shape
        if C != 64 * heads or k.shape != q.shape or v.shape != q.shape:
            raise RuntimeError('cmvae_amd attention: heads of 64 channels and equal Q/K/V shapes are required '
                               '(got %s, %d heads)' % (tuple(q.shape), heads))
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        o = torch.empty_like(q)
        lse = torch.empty(B, heads, S, device=q.device, dtype=torch.float32)
        check(lib.cmvae_attention_fwd(_p(q), _p(k), _p(v), B, S, heads, scale, _p(o), _p(lse), _stream()),
              'cmvae_attention_fwd')
        ctx.heads, ctx.scale = heads, scale
        ctx.save_for_backward(q, k, v, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, lse = ctx.saved_tensors
        B, S, _ = q.shape
        do = do.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        delta = torch.empty_like(lse)
        check(lib.cmvae_attention_bwd(_p(q), _p(k), _p(v), _p(lse), _p(do), B, S, ctx.heads, ctx.scale, _p(dq), _p(dk),
                                      _p(dv), _p(delta), _stream()), 'cmvae_attention_bwd')
        return dq, dk, dv, None, None


def attention(q, k, v, heads, scale):
    """q + softmax(scale * q k^T) v per (episode, head) for [B, S, heads*64] tensors (MAB core)."""
    return _Attention.apply(q, k, v, heads, float(scale))


class _SelfAttention(torch.autograd.Function):
    """fc_q, fc_k, fc_v of a self-attention block as one GEMM over the stacked weights, and the attention core on the
    packed projection (cmvae_attention_qkv_*)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, heads, scale):
        _need(x, 'x'); _need(wq, 'fc_q.weight')
        B, S, Cin = x.shape
        C = 64 * heads
        for w_ in (wq, wk, wv):
            if tuple(w_.shape) != (C, Cin):
                raise RuntimeError('cmvae_amd self_attention: projection weights must be [%d, %d], got %s'
                                   % (C, Cin, tuple(w_.shape)))
        x2 = x.reshape(-1, Cin)
        if x2.stride(1) != 1 or x2.stride(0) % 4:
            x2 = x2.contiguous()
        w = torch.cat([wq, wk, wv])
        b = torch.cat([bq, bk, bv])
        qkv = linear_fwd(x2, w, b, 0)                                     # [B*S, 3C] = [q | k | v]
        o = torch.empty(B, S, C, device=x.device, dtype=torch.float32)
        lse = torch.empty(B, heads, S, device=x.device, dtype=torch.float32)
        check(lib.cmvae_attention_qkv_fwd(_p(qkv), B, S, heads, scale, _p(o), _p(lse), _stream()),
              'cmvae_attention_qkv_fwd')
        ctx.heads, ctx.scale, ctx.shape = heads, scale, x.shape
        ctx.save_for_backward(x2, w, qkv, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        x2, w, qkv, lse = ctx.saved_tensors
        B, S, _ = ctx.shape
        C = 64 * ctx.heads
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        check(lib.cmvae_attention_qkv_bwd(_p(qkv), _p(lse), _p(do), B, S, ctx.heads, ctx.scale, _p(dqkv), _p(delta),
                                          _stream()), 'cmvae_attention_qkv_bwd')
        dx, dW, db = linear_bwd(x2, w, dqkv, ctx.needs_input_grad[0], True)
        return ((None if dx is None else dx.view(ctx.shape)), dW[:C], db[:C], dW[C:2 * C], db[C:2 * C], dW[2 * C:],
                db[2 * C:], None, None)


def self_attention(x, wq, bq, wk, bk, wv, bv, heads, scale):
    """attention(fc_q(x), fc_k(x), fc_v(x)) for x [B, S, C_in] (SAB = MAB(X, X), omniglot/module.py:102-108): the same
    values as the three `linear` calls + `attention`, from one projection GEMM."""
    return _SelfAttention.apply(x, wq, bq, wk, bk, wv, bv, heads, float(scale))


class FlatAdam:
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) over the parameters of a `dist.FlatGrads`.

    The parameters are re-pointed at slices of one flat fp32 buffer (padded to a multiple of 4), the
    gradients are the FlatGrads buffer, and one kernel launch does the whole update."""

    def __init__(self, flat_grads, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = flat_grads.params
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.flat_grads = flat_grads
        n = flat_grads.flat.numel()
        dev = flat_grads.flat.device
        if dev.type != 'cuda':
            raise RuntimeError('FlatAdam: parameters must live on the GPU (there is no CPU path)')
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            off += k
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.t = 0
        self.step_dev = None          # device-side step count: set by device_step(), for steps replayed from a HIP graph

    def device_step(self):
        """Keep the step count on the device from now on (the update launch then holds no per-step host value and can
        be captured into a HIP graph and replayed); idempotent."""
        if self.step_dev is None:
            self.step_dev = torch.full((1,), self.t, dtype=torch.int64, device=self.flat.device)
        return self

    def step(self):
        n = self.flat.numel()
        if self.step_dev is not None:
            self.step_dev += 1
            check(lib.cmvae_adam_step_dev(_p(self.flat), _p(self.flat_grads.flat), _p(self.exp_avg), _p(self.exp_avg_sq),
                                          n, _p(self.step_dev), self.lr, self.betas[0], self.betas[1], self.eps,
                                          self.weight_decay, _stream()), 'cmvae_adam_step_dev')
            return
        self.t += 1
        check(lib.cmvae_adam_step(_p(self.flat), _p(self.flat_grads.flat), _p(self.exp_avg), _p(self.exp_avg_sq), n,
                                  self.t, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                  _stream()), 'cmvae_adam_step')

    def zero_grad(self):
        self.flat_grads.zero()

    def state_dict(self):
        t = self.t if self.step_dev is None else int(self.step_dev)
        return {'step': t, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq}

    def load_state_dict(self, sd):
        self.t = int(sd['step'])
        if self.step_dev is not None:
            self.step_dev.fill_(self.t)
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
