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


def test_scalar_tail_ops_equal_the_host_functions(device):
    from cmvae_amd import ops
    O = torch.ops.cmvae_amd
    g = torch.Generator().manual_seed(11)
    log_q = (torch.randn(2, 7, 3, generator=g) - 50).to(device)
    log_p = (torch.randn(2, 21, generator=g) - 60).to(device)
    q_eps = torch.randn(2, 21, 64, generator=g).to(device)
    A = (torch.randn(64, 64, generator=g) * 0.05).to(device)
    pi = torch.softmax(torch.randn(2, 5, generator=g), -1).to(device)
    trace = (torch.rand(2, 5, generator=g) + 0.5).to(device)
    x = torch.randn(2, 5, 64, generator=g).to(device)
    h = ops.expm_trace(A)
    kl, ha = ops.elbo_tail(log_q, log_p, q_eps, h, A, 0.3, 2.0, 0.01)
    out = O.elbo_tail_fwd(log_q, log_p, q_eps, h, A, 0.3, 2.0, 0.01)
    assert torch.equal(out, torch.stack([kl, ha]))
    gq = torch.tensor([1.5, -0.5], device=device)
    d_q, d_p, d_e, dA, dh = O.elbo_tail_bwd(gq, h, q_eps, A, log_q.numel(), log_p.numel(), 0.3, 2.0, 0.01)
    leaves = [t.clone().requires_grad_(True) for t in (log_q, log_p, q_eps, A)]
    hh = h.clone().requires_grad_(True)
    kl2, ha2 = ops.elbo_tail(leaves[0], leaves[1], leaves[2], hh, leaves[3], 0.3, 2.0, 0.01)
    (1.5 * kl2 - 0.5 * ha2).backward()
    assert torch.equal(d_q.view_as(log_q), leaves[0].grad) and torch.equal(d_p.view_as(log_p), leaves[1].grad)
    assert torch.equal(d_e, leaves[2].grad) and torch.equal(dA, leaves[3].grad) and torch.equal(dh.reshape(()), hh.grad)
    lw = O.prior_logw_fwd(pi, trace, x)
    assert torch.equal(lw, ops.prior_logw(pi, trace, x))
    gw = torch.randn(2, 5, generator=g).to(device)
    xr = x.clone().requires_grad_(True)
    (ops.prior_logw(pi, trace, xr) * gw).sum().backward()
    assert torch.equal(O.prior_logw_bwd(trace, x, gw), xr.grad)
    with pytest.raises(RuntimeError):
        O.prior_logw_fwd(pi, trace, x[:, :, :32])
    with pytest.raises(RuntimeError):
        O.elbo_tail_bwd(gq[:1], h, q_eps, A, 1, 1, 0.3, 2.0, 0.01)


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
    # the packed projection [q | k | v] per row: the same values from column slices
    qkv = torch.cat([q, k, v], 2)
    o2, lse2 = O.attention_qkv_fwd(qkv, 2, 0.125)
    assert torch.equal(o2, o) and torch.equal(lse2, lse)
    dqkv = O.attention_qkv_bwd(qkv, lse2, do, 2, 0.125)
    assert torch.equal(dqkv, torch.cat([dq, dk, dv], 2))
    with pytest.raises(RuntimeError):
        O.attention_qkv_fwd(qkv[:, :, :-64], 2, 0.125)
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
    # the four sub-pixel phases of a transposed conv in one launch: geometries one after the other in one list
    k, Hi = 4, 7
    Ho = conv.convT_out(Hi, k)
    phases = [conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=True) for py in (0, 1) for px in (0, 1)]
    geoms = [ge for ge, _ in phases]
    flat = [v for ge in geoms for v in geom_to_list(ge)]
    wt4 = (torch.randn(k * k, 64, 64, generator=g) * 0.1).to(device)
    want = torch.empty(n * Ho * Ho, 64, device=device)
    pw = conv.conv_gemm_multi(geoms, x, wt4, want, bias=b, stats=True)
    got = torch.empty_like(want)
    pg = O.conv_gemm_multi(flat, x, wt4, got, b, None, None, True)
    assert torch.equal(got, want) and torch.equal(pg, pw)
    gy4 = torch.randn(n * Ho * Ho, 64, generator=g).to(device)
    dWw, dWg = torch.zeros(64, 64, k, k, device=device), torch.zeros(64, 64, k, k, device=device)
    conv.conv_wgrad_multi(geoms, x, gy4, dWw, [t for _, t in phases], 64 * k * k, k * k, db=dbw)
    O.conv_wgrad_multi(flat, x, gy4, dWg, [t for _, ts in phases for t in ts], 64 * k * k, k * k, dbg, None, None, 64)
    assert torch.equal(dWw, dWg) and torch.equal(dbw, dbg)
    # data gradient with the BatchNorm-backward sums of the layer below from its epilogue
    gd, _ = conv.geom_convT_dgrad(n, Hi, Hi, k, pixel_major=True)
    wtd = (torch.randn(k, k, 64, 64, generator=g) * 0.1).to(device)
    fold = [torch.rand(64, generator=g).to(device) + 0.5 for _ in range(4)]
    dw, dg_ = torch.empty(n * Hi * Hi, 64, device=device), torch.empty(n * Hi * Hi, 64, device=device)
    pw = conv.conv_gemm_bnbwd(gd, gy4, wtd, dw, x, fold)
    pg = O.conv_gemm_bnbwd(geom_to_list(gd), gy4, wtd, dg_, x, *fold)
    assert torch.equal(dw, dg_) and torch.equal(pw, pg)
    with pytest.raises(RuntimeError):
        O.conv_gemm_bnbwd(geom_to_list(gd), gy4, wtd, dg_, x[:-1], *fold)
    with pytest.raises(RuntimeError):
        O.conv_gemm_multi(flat[:-1], x, wt4, got, b, None, None, False)
    with pytest.raises(RuntimeError):
        O.conv_wgrad_multi(flat, x, gy4, dWg, [0, 1, 2], 64 * k * k, k * k, dbg, None, None, 64)


def test_encoder_and_decoder_ops_with_autograd(device):
    """torch.ops.cmvae_amd.encode_images / decoder_bce: the module-based calls of the product path, value and gradients,
    with the BatchNorm state handed over as tensors (running statistics updated in place)."""
    import copy
    from cmvae_amd.model import OmniglotGMVAE
    from cmvae_amd.encoder import encode_images
    from cmvae_amd.decoder import decoder_bce
    O = torch.ops.cmvae_amd
    torch.manual_seed(2)
    m1 = OmniglotGMVAE(component_size=5, train_mc_sample_size=4, test_mc_sample_size=4).to(device).train()
    m2 = copy.deepcopy(m1)
    g = torch.Generator().manual_seed(8)
    x = (torch.rand(12, 1, 28, 28, generator=g) < 0.2).float().to(device)
    h1 = encode_images(m1.encoder, x)
    seq = m2.encoder.encoder
    params, running = [], []
    for l in range(4):
        cv, bn = seq[4 * l], seq[4 * l + 1]
        params += [cv.weight, cv.bias, bn.weight, bn.bias]
        running += [bn.running_mean, bn.running_var, bn.num_batches_tracked]
    h2 = O.encode_images(x, True, params, running, seq[1].eps, seq[1].momentum)
    assert torch.equal(h1, h2)
    gy = torch.randn(h1.shape, generator=g).to(device)
    h1.backward(gy); h2.backward(gy)
    for p1, p2 in zip(m1.encoder.parameters(), m2.encoder.parameters()):
        assert torch.equal(p1.grad, p2.grad)
    for b1, b2 in zip(m1.encoder.buffers(), m2.encoder.buffers()):
        assert torch.equal(b1, b2)
    hr = torch.randn(24, 256, generator=g).to(device)
    tgt = (torch.rand(6, 784, generator=g) < 0.2).float().to(device)
    hr1, hr2 = hr.clone().requires_grad_(True), hr.clone().requires_grad_(True)
    p1, l1 = decoder_bce(m1.decoder, hr1, tgt, 4)
    seq = m2.decoder.decoder
    convs, bns = (seq[0], seq[3], seq[6], seq[9]), (seq[1], seq[4], seq[7])
    params, running = [], []
    for l in range(3):
        params += [convs[l].weight, convs[l].bias, bns[l].weight, bns[l].bias]
        running += [bns[l].running_mean, bns[l].running_var, bns[l].num_batches_tracked]
    params += [convs[3].weight, convs[3].bias]
    p2, l2 = O.decoder_bce(hr2, tgt, 4, True, params, running, bns[0].eps, bns[0].momentum)
    assert torch.equal(p1, p2) and torch.equal(l1, l2)
    (l1 / 24).backward(); (l2 / 24).backward()
    assert torch.equal(hr1.grad, hr2.grad)
    for q1, q2 in zip(m1.decoder.parameters(), m2.decoder.parameters()):
        assert torch.equal(q1.grad, q2.grad)
    for b1, b2 in zip(m1.decoder.buffers(), m2.decoder.buffers()):
        assert torch.equal(b1, b2)
