"""Parity of the Linear / bulk causal-MLP / Adam entry points (through the C-ABI) against the oracle's
plain PyTorch fp32 restatement on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


ACTS = {None: lambda v: v, 'relu': F.relu, 'elu': F.elu}


@pytest.mark.parametrize('rows,in_f,out_f,act', [
    (300, 128, 256, 'elu'), (1000, 256, 256, 'relu'), (77, 64, 2048, 'elu'), (513, 2048, 64, None),
    (1, 256, 128, None), (1200, 512, 512, 'relu'), (129, 64, 64, None),
    # 64 -> many column blocks over >= 8192 rows: the weight-stationary kernel (ragged last row tile)
    (8200, 64, 512, 'elu'), (9001, 64, 1024, None),
    # >= 16 384 rows with in_features <= 384 (and the data gradient where out_features <= 384): weight planes resident
    # in LDS, products on the bf16 pipe (ragged last row tile; 1, 2, 4 and 6 taps)
    (20001, 256, 256, 'elu'), (16400, 64, 2048, 'relu'), (17000, 384, 128, None), (20000, 128, 64, None),
    # data gradient with out_features > 384 over few column blocks: split over the contraction (6 and 3 tap groups of
    # the bf16-pipe kernel, partial products added in fixed order; ragged last row tile, two column blocks)
    (25601, 64, 2048, None), (20000, 128, 1024, 'elu')])
def test_linear_fwd_bwd(rows, in_f, out_f, act):
    from cmvae_amd.linear import linear
    g = torch.Generator().manual_seed(rows + in_f)
    x = torch.randn(rows, in_f, generator=g)
    W = torch.randn(out_f, in_f, generator=g) / np.sqrt(in_f)
    b = torch.randn(out_f, generator=g) * 0.1
    gy = torch.randn(rows, out_f, generator=g)
    xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
    yr = ACTS[act](F.linear(xr, Wr, br))
    yr.backward(gy)
    xd, Wd, bd = (t.cuda().requires_grad_(True) for t in (x, W, b))
    yd = linear(xd, Wd, bd, act)
    yd.backward(gy.cuda())
    assert _rel(yd, yr) < 1e-5
    assert _rel(xd.grad, xr.grad) < 1e-5
    assert _rel(Wd.grad, Wr.grad) < 2e-5
    assert _rel(bd.grad, br.grad) < 2e-5


@pytest.mark.parametrize('n_t,M,in_f,out_f,act', [(40, 10, 1024, 512, 'relu'), (7, 3, 128, 64, 'elu'),
                                                   (1, 1, 64, 128, None), (300, 10, 1280, 640, 'relu')])
def test_linear_mse_epilogue(n_t, M, in_f, out_f, act):
    """Last proj layer + F.mse_loss(reduction='sum') of the feature models (mimgnet/model.py:59-66, 388-391) from one
    GEMM launch: value, the gradient of the loss alone, and loss + a use of the output together."""
    from cmvae_amd.linear import linear_mse
    g = torch.Generator().manual_seed(n_t * 31 + M)
    rows = n_t * M
    x = torch.randn(rows, in_f, generator=g)
    W = torch.randn(out_f, in_f, generator=g) / np.sqrt(in_f)
    b = torch.randn(out_f, generator=g) * 0.1
    tgt = torch.randn(n_t, out_f, generator=g).abs()
    gy = torch.randn(rows, out_f, generator=g)
    for use_y in (False, True):
        xr, Wr, br = (t.clone().double().requires_grad_(True) for t in (x, W, b))
        yr = ACTS[act](F.linear(xr, Wr, br))
        lr = F.mse_loss(yr, tgt.double().repeat_interleave(M, 0), reduction='sum')
        (lr * 0.37 + ((yr * gy.double()).sum() if use_y else 0.)).backward()
        xd, Wd, bd = (t.cuda().requires_grad_(True) for t in (x, W, b))
        yd, ld = linear_mse(xd, Wd, bd, act, tgt.cuda(), M)
        (ld * 0.37 + ((yd * gy.cuda()).sum() if use_y else 0.)).backward()
        assert _rel(yd, yr) < 1e-5
        assert abs(float(ld.detach()) - float(lr.detach())) < 2e-6 * abs(float(lr.detach()))
        assert _rel(xd.grad, xr.grad) < 1e-5
        assert _rel(Wd.grad, Wr.grad) < 2e-5
        assert _rel(bd.grad, br.grad) < 2e-5


def test_linear_mse_rejects_mismatched_targets():
    from cmvae_amd.linear import linear_mse
    x, W = torch.randn(30, 64).cuda(), torch.randn(64, 64).cuda()
    with pytest.raises(RuntimeError):
        linear_mse(x, W, None, 'relu', torch.randn(4, 64).cuda(), 10)


def test_linear_batched_shape_no_bias_no_input_grad():
    from cmvae_amd.linear import linear
    x = torch.randn(4, 50, 256)
    W = torch.randn(128, 256) / 16
    yd = linear(x.cuda(), W.cuda().requires_grad_(True), None, None)
    assert yd.shape == (4, 50, 128)
    assert _rel(yd, F.linear(x, W)) < 1e-5
    yd.sum().backward()


def test_linear_rejects_bad_sizes():
    from cmvae_amd.linear import linear
    with pytest.raises(RuntimeError):
        linear(torch.randn(8, 100).cuda(), torch.randn(64, 100).cuda())
    with pytest.raises(RuntimeError):
        linear(torch.randn(8, 64), torch.randn(64, 64))


@pytest.mark.parametrize('rows', [1, 200, 1500])
def test_causal_mlp_bulk(rows):
    from cmvae_amd.linear import causal_mlp
    from oracle.gmvae_oracle import causal
    g = torch.Generator().manual_seed(rows)
    P = {'causal.fc1.weight': torch.randn(2048, 64, generator=g) * 0.05,
         'causal.fc1.bias': torch.randn(2048, generator=g) * 0.1,
         'causal.fc2.0.weight': torch.randn(64, 32, 1, generator=g) * 0.2,
         'causal.fc2.0.bias': torch.randn(64, 1, generator=g) * 0.1}
    x = torch.randn(rows, 64, generator=g)
    gy = torch.randn(rows, 64, generator=g)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    er = causal(xr, Pr, linear=False)
    er.backward(gy)
    Pd = {k: v.cuda().requires_grad_(True) for k, v in P.items()}
    xd = x.cuda().requires_grad_(True)
    ed = causal_mlp(xd, Pd['causal.fc1.weight'], Pd['causal.fc1.bias'], Pd['causal.fc2.0.weight'],
                    Pd['causal.fc2.0.bias'])
    ed.backward(gy.cuda())
    assert _rel(ed, er) < 1e-5
    assert _rel(xd.grad, xr.grad) < 1e-5
    for k in P:
        assert _rel(Pd[k].grad, Pr[k].grad) < 3e-5, k


def test_flat_adam_matches_torch_adam():
    from cmvae_amd.dist import FlatGrads
    from cmvae_amd.linear import FlatAdam
    torch.manual_seed(3)
    shapes = [(64, 64), (7,), (33, 5), (2048, 64), (1,)]
    ref = [torch.randn(s).requires_grad_(True) for s in shapes]
    dev = [r.detach().clone().cuda().requires_grad_(True) for r in ref]
    opt_r = torch.optim.Adam(ref, lr=1e-3, weight_decay=5e-5)
    fg = FlatGrads(dev)
    opt_d = FlatAdam(fg, lr=1e-3, weight_decay=5e-5)
    for it in range(5):
        for r, d in zip(ref, dev):
            gr = torch.randn(r.shape) * (it + 1)
            r.grad = gr.clone()
            d.grad.copy_(gr)
        opt_r.step()
        opt_d.step()
    for r, d in zip(ref, dev):
        assert _rel(d, r) < 1e-6


@pytest.mark.parametrize('B,S,heads', [(2, 5, 4), (3, 37, 4), (4, 300, 4), (1, 16, 1), (2, 600, 2), (2, 1, 4),
                                         (1, 1024, 1), (1, 17, 2)])
def test_attention_core(B, S, heads):
    """O = Q + softmax(QK^T * scale) V per head, forward and all three gradients (omniglot/module.py:92-99)."""
    from cmvae_amd.linear import attention
    g = torch.Generator().manual_seed(S)
    C = 64 * heads
    q, k, v, go = (torch.randn(B, S, C, generator=g) for _ in range(4))
    scale = 1.0 / np.sqrt(C)

    def ref(q, k, v):
        qh, kh, vh = (t.view(B, S, heads, 64).transpose(1, 2) for t in (q, k, v))
        att = torch.softmax(qh.matmul(kh.transpose(-1, -2)) * scale, -1)
        return (qh + att.matmul(vh)).transpose(1, 2).reshape(B, S, C)

    qr, kr, vr = (t.clone().double().requires_grad_(True) for t in (q, k, v))
    orf = ref(qr, kr, vr)
    orf.backward(go.double())
    qd, kd, vd = (t.cuda().requires_grad_(True) for t in (q, k, v))
    od = attention(qd, kd, vd, heads, scale)
    od.backward(go.cuda())
    assert _rel(od, orf) < 1e-5
    for a, b in ((qd, qr), (kd, kr), (vd, vr)):
        assert _rel(a.grad, b.grad) < 2e-5


@pytest.mark.parametrize('B,S,heads,Cin', [(4, 300, 4, 256), (3, 37, 4, 256), (2, 1, 4, 256), (2, 100, 1, 128),
                                             (16, 5, 4, 256)])
def test_self_attention_one_projection_gemm(B, S, heads, Cin):
    """SAB = MAB(X, X) (omniglot/module.py:102-108): fc_q, fc_k, fc_v as one GEMM over the stacked weights and the
    attention core on the packed projection give the three-GEMM path's output bit for bit (same per-column sums) and
    its gradients to rounding (dx is one K = 3C contraction instead of three added up); both against float64."""
    from cmvae_amd.linear import attention, linear, self_attention
    g = torch.Generator().manual_seed(S + Cin)
    C = 64 * heads
    x = torch.randn(B, S, Cin, generator=g)
    ws = [torch.randn(C, Cin, generator=g) / np.sqrt(Cin) for _ in range(3)]
    bs = [torch.randn(C, generator=g) * 0.1 for _ in range(3)]
    go = torch.randn(B, S, C, generator=g)
    scale = 1.0 / np.sqrt(C)

    def leaves():
        return [t.cuda().requires_grad_(True) for t in [x] + ws + bs]

    a = leaves()
    o1 = attention(linear(a[0], a[1], a[4]), linear(a[0], a[2], a[5]), linear(a[0], a[3], a[6]), heads, scale)
    o1.backward(go.cuda())
    f = leaves()
    o2 = self_attention(f[0], f[1], f[4], f[2], f[5], f[3], f[6], heads, scale)
    o2.backward(go.cuda())
    assert torch.equal(o1, o2)
    r = [t.double().requires_grad_(True) for t in [x] + ws + bs]
    qh, kh, vh = ((r[0].matmul(r[1 + i].t()) + r[4 + i]).view(B, S, heads, 64).transpose(1, 2) for i in range(3))
    orf = (qh + torch.softmax(qh.matmul(kh.transpose(-1, -2)) * scale, -1).matmul(vh)).transpose(1, 2).reshape(B, S, C)
    orf.backward(go.double())
    assert _rel(o2, orf) < 1e-5
    for i, (t1, t2, tr) in enumerate(zip(a, f, r)):
        if i == 5:        # fc_k's bias shifts every score of a row alike: its gradient is zero up to rounding
            assert float(t2.grad.abs().max()) < 1e-5 and float(t1.grad.abs().max()) < 1e-5
            continue
        assert _rel(t2.grad, tr.grad) < 3e-5
        assert _rel(t2.grad, t1.grad) < 1e-5
    with pytest.raises(RuntimeError):
        self_attention(f[0], f[1][:, :64], f[4], f[2], f[5], f[3], f[6], heads, scale)


def test_attention_rejects_bad_shapes():
    from cmvae_amd.linear import attention
    with pytest.raises(RuntimeError):
        attention(torch.randn(1, 4, 96).cuda(), torch.randn(1, 4, 96).cuda(), torch.randn(1, 4, 96).cuda(), 2, 0.1)
    with pytest.raises(RuntimeError):
        q = torch.randn(1, 2000, 64).cuda()
        attention(q, q, q, 1, 0.1)


@pytest.mark.parametrize('B,N', [(4, 9600), (3, 257), (2, 1), (1, 256)])
def test_center_rows(B, N):
    """q_z - q_z.mean(1, keepdim=True) (omniglot/model.py:449) and its gradient against torch."""
    from cmvae_amd.ops import center_rows
    g = torch.Generator().manual_seed(N)
    x = torch.randn(B, N, 64, generator=g) + 0.5
    go = torch.randn(B, N, 64, generator=g)
    xr = x.clone().double().requires_grad_(True)
    yr = xr - xr.mean(1, keepdim=True)
    yr.backward(go.double())
    xd = x.cuda().requires_grad_(True)
    yd = center_rows(xd)
    yd.backward(go.cuda())
    assert _rel(yd, yr) < 1e-6 and _rel(xd.grad, xr.grad) < 1e-6


def test_adam_device_step_equals_host_step():
    """cmvae_adam_step_dev (step count read from device memory, replayable from a HIP graph) walks the trajectory of
    cmvae_adam_step and of torch.optim.Adam: bias corrections computed in double on both sides."""
    from cmvae_amd.dist import FlatGrads
    from cmvae_amd.linear import FlatAdam
    g = torch.Generator().manual_seed(12)
    w0 = [torch.randn(37, 64, generator=g), torch.randn(129, generator=g)]
    grads = [[torch.randn(37, 64, generator=g), torch.randn(129, generator=g)] for _ in range(6)]
    pt = [torch.nn.Parameter(t.clone()) for t in w0]
    ref = torch.optim.Adam(pt, lr=3e-3, weight_decay=1e-4)
    runs = []
    for dev_step in (False, True):
        ps = [torch.nn.Parameter(t.clone().cuda()) for t in w0]
        flat = FlatGrads(ps)
        opt = FlatAdam(flat, lr=3e-3, weight_decay=1e-4)
        if dev_step:
            opt.device_step()
        for gs in grads:
            flat.zero()
            for p, gr in zip(ps, gs):
                p.grad = gr.cuda()
            flat.allreduce()
            opt.step()
        assert opt.state_dict()['step'] == 6
        runs.append([p.detach().cpu() for p in ps])
    for gs in grads:
        for p, gr in zip(pt, gs):
            p.grad = gr.clone()
        ref.step()
    for a, b, r in zip(runs[0], runs[1], pt):
        assert torch.equal(a, b)
        assert float((a - r.detach()).abs().max()) < 2e-6
    # a checkpointed step count survives the switch to the device-side counter
    ps = [torch.nn.Parameter(t.clone().cuda()) for t in w0]
    opt = FlatAdam(FlatGrads(ps), lr=1e-3).device_step()
    opt.load_state_dict({'step': 41, 'exp_avg': torch.zeros_like(opt.exp_avg), 'exp_avg_sq': torch.zeros_like(opt.exp_avg_sq)})
    assert opt.state_dict()['step'] == 41
