"""`torch.ops.cmvae_amd.*` (cmvae_amd/registry.py): every registered op is reachable through the dispatcher and returns
what the product path's own host function returns on the same inputs (bit for bit: one implementation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_every_leaf_op_is_registered():
    import cmvae_amd.registry as reg
    for name in reg.REGISTERED:
        assert hasattr(torch.ops.cmvae_amd, name), name


def test_em_and_prediction_ops_equal_the_host_functions(device):
    from cmvae_amd import ops
    O = torch.ops.cmvae_amd
    g = torch.Generator().manual_seed(5)
    B, N, K, Ns = 2, 200, 5, 20
    z = torch.randn(B, N, 64, generator=g).to(device)
    A = (torch.randn(64, 64, generator=g) * 0.05).to(device)
    idx = torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(B)]).to(device)
    want = ops.unsup_prior(z, idx, ops.Causal(True, A), 4, False, 1.0)
    got = O.unsup_prior(z, idx, [A], 4, False, 1.0)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    sz = torch.randn(B, Ns, 64, generator=g).to(device)
    y = (torch.arange(Ns) % K)[None].repeat(B, 1).to(device)
    want = ops.semi_prior(z, sz, y, K, ops.Causal(True, A), 3, 1.0)
    got = O.semi_prior(z, sz, y, K, [A], 3, 1.0)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    pi, mu, logvar, tau, logw = want
    te, M = 50, 4
    yw, _ = ops.posterior_argmax(z, mu, logvar, logw, te, M)
    yg, post = O.posterior_argmax(z, mu, logvar, logw, te, M, False)
    assert torch.equal(yw, yg) and post.numel() == 0
    assert torch.equal(O.center_rows(z), ops.center_rows(z))
    assert torch.equal(O.gram_inverse([A], None, 1.0, 1), ops.gram_inverse(ops.Causal(True, A)))
    eps_w, _ = ops.causal_rows(mu, ops.Causal(True, A))
    eps_g, lw = O.causal_rows(mu, [A], None, None)
    assert torch.equal(eps_w, eps_g) and lw.numel() == 0
    with pytest.raises(RuntimeError):
        O.unsup_prior(z, idx, [A, A], 4, False, 1.0)


def test_linear_attention_adam_ops(device):
    from cmvae_amd.linear import linear, attention
    O = torch.ops.cmvae_amd
    g = torch.Generator().manual_seed(6)
    x = torch.randn(300, 128, generator=g).to(device)
    W = (torch.randn(256, 128, generator=g) / 11).to(device)
    b = torch.randn(256, generator=g).to(device)
    y = O.linear_fwd(x, W, b, 1)
    assert torch.equal(y, linear(x, W, b, 'relu'))
    dz = torch.randn(300, 256, generator=g).to(device)
    dx, dW, db = O.linear_bwd(x, W, dz, True, True)
    ref_dx, ref_dW, ref_db = dz.double() @ W.double(), dz.double().t() @ x.double(), dz.double().sum(0)
    for a, r in ((dx, ref_dx), (dW, ref_dW), (db, ref_db)):
        assert float((a.double() - r).abs().max() / r.abs().max()) < 2e-5
    dx2, _, db2 = O.linear_bwd(x, W, dz, False, False)
    assert dx2.numel() == 0 and db2.numel() == 0
    tgt = torch.randn(30, 256, generator=g).abs().to(device)
    y2, loss = O.linear_mse_fwd(x, W, b, 1, tgt, 10)
    assert torch.equal(y2, y)
    ref = F.mse_loss(y.double(), tgt.double().repeat_interleave(10, 0), reduction='sum')
    assert abs(float(loss) - float(ref)) < 2e-6 * float(ref)
    gl = torch.tensor([0.5], device=device)
    dzm = O.mse_act_bwd(y, tgt, 10, 1, gl, 1.0)
    refd = (y.double() - tgt.double().repeat_interleave(10, 0)) * (y > 0) * 2 * 0.5
    assert float((dzm.double() - refd).abs().max()) < 1e-5 * float(refd.abs().max())
    assert torch.equal(O.act_bwd(dz, y, 1), dz * (y > 0))
    q, k, v = (torch.randn(2, 50, 128, generator=g).to(device) for _ in range(3))
    o, lse = O.attention_fwd(q, k, v, 2, 0.125)
    assert torch.equal(o, attention(q, k, v, 2, 0.125))
    do = torch.randn(2, 50, 128, generator=g).to(device)
    dq, dk, dv = O.attention_bwd(q, k, v, lse, do, 2, 0.125)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    attention(qr, kr, vr, 2, 0.125).backward(do)
    assert torch.equal(dq, qr.grad) and torch.equal(dk, kr.grad) and torch.equal(dv, vr.grad)
    n = 1024
    p0 = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=1e-3, weight_decay=1e-4)
    pt.grad = gr.clone()
    opt.step()
    pd, m, vv = p0.clone().to(device), torch.zeros(n, device=device), torch.zeros(n, device=device)
    O.adam_step(pd, gr.to(device), m, vv, 1, 1e-3, 0.9, 0.999, 1e-8, 1e-4)
    assert float((pd.cpu() - pt.detach()).abs().max()) < 1e-6


def test_conv_ops_take_the_geometry_as_a_list(device):
    from cmvae_amd import conv
    from cmvae_amd.registry import geom_to_list, geom_from_list
    O = torch.ops.cmvae_amd
    g = torch.Generator().manual_seed(7)
    n, H = 32, 7
    geom = conv.geom_conv3x3(n, H, H)
    lst = geom_to_list(geom)
    assert geom_to_list(geom_from_list(lst)) == lst
    x = torch.randn(n * H * H, 64, generator=g).to(device)
    wt = (torch.randn(9, 64, 64, generator=g) * 0.1).to(device)
    b = torch.randn(64, generator=g).to(device)
    want = torch.empty(n * H * H, 64, device=device)
    pw = conv.conv_gemm(geom, x, wt, want, bias=b, stats=True)
    got = torch.empty_like(want)
    pg = O.conv_gemm(lst, x, wt, got, b, None, None, 0, 1, True)
    assert torch.equal(got, want) and torch.equal(pg, pw)
    gy = torch.randn(n * H * H, 64, generator=g).to(device)
    dWw, dWg = torch.zeros(64, 64, 3, 3, device=device), torch.zeros(64, 64, 3, 3, device=device)
    dbw, dbg = torch.zeros(64, device=device), torch.zeros(64, device=device)
    conv.conv_wgrad(geom, x, gy, dWw, list(range(9)), s_ci=9, s_co=64 * 9, db=dbw)
    O.conv_wgrad(lst, x, gy, dWg, list(range(9)), 9, 64 * 9, dbg, False, None, None, 64, 0)
    assert torch.equal(dWw, dWg) and torch.equal(dbw, dbg)
    with pytest.raises(RuntimeError):
        O.conv_gemm(lst[:-1], x, wt, got, b, None, None, 0, 1, False)
