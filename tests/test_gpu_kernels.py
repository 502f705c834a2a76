"""Kernel-level parity: every C-ABI entry point against the CPU oracle on the same seeded inputs.
Tolerance: 1e-4 relative (north_star, fp32); labels bit-exact."""
import numpy as np
import pytest
import torch

from tests.helpers import build_pair, clustered_latents, rel_err, RTOL

pytestmark = pytest.mark.gpu


def _causal_desc(model, device):
    model.to(device)
    return model._causal_desc()


@pytest.mark.parametrize('linear', [False, True])
def test_causal_rows_and_prior_term(device, linear):
    from cmvae_amd import ops
    model, oracle = build_pair('omniglot', linear=linear)
    c = _causal_desc(model, device)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 64, generator=g)
    pi = torch.rand(37, generator=g) + 0.1
    tau = torch.rand(37, generator=g) + 0.5
    eps, logw = ops.causal_rows(x.to(device), c, pi.to(device), tau.to(device))
    ref_eps = oracle._causal(x)
    ref_logw = torch.log(pi) + oracle._glp(ref_eps, torch.log(tau)[:, None].repeat(1, 64))
    assert rel_err(eps, ref_eps) < 1e-5
    assert rel_err(logw, ref_logw) < 1e-5


@pytest.mark.parametrize('linear', [False, True])
def test_gram_inverse(device, linear):
    from cmvae_amd import ops
    model, oracle = build_pair('omniglot', linear=linear)
    c = _causal_desc(model, device)
    I = torch.eye(64)
    # shared (tau = 1) matrix
    out = ops.gram_inverse(c, None, 1.0, 1)
    C = oracle._causal(I)
    ref = torch.inverse(I + C @ C.T)
    assert rel_err(out[0], ref) < 1e-5
    # per-component: C rows = causal(diag(exp(.5 lv)) * tau)
    g = torch.Generator().manual_seed(5)
    lv = torch.randn(2, 3, 64, generator=g) * 0.5
    out = ops.gram_inverse(c, lv.to(device), 0.7)
    std = torch.diag_embed((0.5 * lv).exp()) * 0.7
    Cb = oracle._causal(std.view(2, -1, 64)).view(std.shape)
    ref = torch.inverse(I[None, None] + Cb.matmul(Cb.transpose(-1, -2)))
    assert rel_err(out.view(2, 3, 64, 64), ref) < 1e-5
    # it really is the inverse (fp64 check, independent of the oracle's LU)
    Mx = (I[None, None] + Cb.matmul(Cb.transpose(-1, -2))).double()
    resid = (out.view(2, 3, 64, 64).double().cpu() @ Mx - torch.eye(64, dtype=torch.float64)).abs().max()
    assert resid < 1e-4


@pytest.mark.parametrize('K,N,linear,gamma,fix_pi', [(3, 96, False, 1, False), (5, 200, True, 1, False),
                                                       (20, 333, False, 1, False), (20, 1000, False, 4, True),
                                                       (1, 70, False, 1, False)])
def test_unsup_prior_trajectory(device, K, N, linear, gamma, fix_pi):
    """EM parameters after 0,1,..,iters iterations against the oracle's literal loop."""
    from cmvae_amd import ops
    B = 3
    model, oracle = build_pair('omniglot', K=K, linear=linear, gamma=gamma, fix_pi=fix_pi)
    c = _causal_desc(model, device)
    z, _, _ = clustered_latents(B, N, K, seed=K + N)
    rng = np.random.RandomState(7)
    idx = np.stack([rng.choice(N, K, replace=False) for _ in range(B)], 0)
    traj = []
    oracle.unsup_prior(z, idx, iters=4, trajectory=traj)
    zd = z.to(device)
    idxd = torch.from_numpy(idx).to(device)
    for it in (0, 1, 2, 4):
        pi, mu, tau, logw = ops.unsup_prior(zd, idxd, c, it, fix_pi, model.gamma1)
        rpi, rmu, rtau = traj[it]
        assert rel_err(mu, rmu) < RTOL, 'mu after %d iterations' % it
        assert rel_err(pi, rpi) < RTOL
        assert rel_err(tau, rtau) < 1e-6
        rlogw = torch.log(rpi) + oracle._glp(oracle._causal(rmu), torch.log(rtau)[:, :, None].repeat(1, 1, 64))
        assert rel_err(logw, rlogw) < RTOL


def test_unsup_outlier_seed_stays_finite(device):
    """A far-outlier seed (component that owns a single point) stays finite and matches the oracle.
    (An exactly-zero N_k — the 1e-22 guard of omniglot/model.py:193 — needs every responsibility
    of a component to underflow; the guard is applied literally in em_finalize_kernel.)"""
    from cmvae_amd import ops
    model, oracle = build_pair('omniglot', K=2, linear=True)
    with torch.no_grad():
        model.causal.A.zero_()
        oracle.P['causal.A'].zero_()
    c = _causal_desc(model, device)
    g = torch.Generator().manual_seed(23)
    z = torch.randn(1, 64, 64, generator=g) * 0.01
    z[0, 0] = 30.0
    idx = np.array([[1, 0]])
    traj = []
    oracle.unsup_prior(z, idx, iters=2, trajectory=traj)
    pi, mu, tau, logw = ops.unsup_prior(z.to(device), torch.from_numpy(idx).to(device), c, 2, False, 1.0)
    assert torch.isfinite(mu).all() and torch.isfinite(pi).all() and torch.isfinite(logw).all()
    assert rel_err(mu, traj[2][1]) < RTOL
    assert rel_err(pi, traj[2][0]) < RTOL
    # KAT-3: A = 0 => Minv = I/2: the outlier component's mean is half the outlier
    assert abs(float(mu[0, 1, 0]) - 15.0) < 1e-3


@pytest.mark.parametrize('K,Nu,Ns,linear,gamma', [(3, 120, 24, False, 1), (5, 300, 40, True, 1),
                                                   (20, 480, 160, False, 1), (5, 200, 65, False, 2)])
def test_semi_prior_trajectory(device, K, Nu, Ns, linear, gamma):
    from cmvae_amd import ops
    B = 2
    model, oracle = build_pair('omniglot', K=K, linear=linear, gamma=gamma)
    c = _causal_desc(model, device)
    z, lab, _ = clustered_latents(B, Nu + Ns, K, seed=K + Nu)
    uz, sz = z[:, :Nu].contiguous(), z[:, Nu:].contiguous()
    y = lab[:, Nu:].contiguous()
    y[:, :K] = torch.arange(K)        # every class has at least one labeled sample
    yoh = torch.nn.functional.one_hot(y, K).float()
    traj = []
    oracle.semi_prior(uz, sz, yoh, iters=3, trajectory=traj)
    for it in (0, 1, 3):
        pi, mu, lv, tau, logw = ops.semi_prior(uz.to(device), sz.to(device), y.to(device), K, c, it, model.gamma1)
        rpi, rmu, rlv, rtau = traj[it]
        assert rel_err(mu, rmu) < RTOL, 'mu after %d iterations' % it
        # |d logvar| is the element-wise relative error of the variance: the 1e-4 bar applies to it directly, and
        # norm-wise to the variance and to logvar
        print('  it %d: logvar norm-rel %.2e  max|d logvar| %.2e' % (it, rel_err(lv, rlv), float((lv.cpu() - rlv).abs().max())))
        assert rel_err(lv.exp(), rlv.exp()) < RTOL
        assert float((lv.cpu() - rlv).abs().max()) < RTOL, 'variance element-wise after %d iterations' % it
        assert rel_err(lv, rlv) < RTOL, 'logvar after %d iterations' % it
        assert rel_err(pi, rpi) < 1e-6 and rel_err(tau, rtau) < 1e-6


@pytest.mark.parametrize('K,te,M', [(3, 7, 4), (5, 75, 32), (20, 30, 32), (5, 9, 256), (4, 11, 10)])
def test_posterior_argmax_bit_exact(device, K, te, M):
    from cmvae_amd import ops
    B = 2
    model, oracle = build_pair('omniglot', K=K, M=M)
    z, lab, centres = clustered_latents(B, te * M, K, seed=te)
    g = torch.Generator().manual_seed(9)
    mu = centres + 0.1 * torch.randn(B, K, 64, generator=g)
    lv = torch.randn(B, K, 64, generator=g) * 0.3
    logw = torch.randn(B, K, generator=g)
    y, post = ops.posterior_argmax(z.to(device), mu.to(device), lv.to(device), logw.to(device), te, M, want_post=True)
    ll = oracle._glp(z.view(B, te, M, 64)[:, :, :, None, :].repeat(1, 1, 1, K, 1),
                     mu[:, None, None].repeat(1, te, M, 1, 1), lv[:, None, None].repeat(1, te, M, 1, 1)) \
        + logw[:, None, None, :]
    ref = torch.exp(ll - torch.logsumexp(ll, -1, keepdim=True)).mean(-2)
    assert rel_err(post, ref) < RTOL
    assert torch.equal(y.cpu(), ref.argmax(-1))


@pytest.mark.parametrize('K,N', [(3, 50), (20, 700), (5, 64)])
def test_mixture_loglik_fwd_bwd(device, K, N):
    from cmvae_amd import ops
    B = 2
    g = torch.Generator().manual_seed(11)
    z = torch.randn(B, N, 64, generator=g)
    mu = torch.randn(B, K, 64, generator=g)
    logw = torch.randn(B, K, generator=g)
    up = torch.randn(B, N, generator=g)
    zr = z.clone().double().requires_grad_(True)
    lr = logw.clone().double().requires_grad_(True)
    ll = (-0.5 * (zr[:, :, None, :] - mu.double()[:, None]).pow(2) - 0.5 * np.log(2 * np.pi)).sum(-1) + lr[:, None, :]
    ref = torch.logsumexp(ll, -1)
    (ref * up.double()).sum().backward()
    zd = z.to(device).requires_grad_(True)
    ld = logw.to(device).requires_grad_(True)
    out = ops.mixture_loglik(zd, mu.to(device), ld)
    (out * up.to(device)).sum().backward()
    assert rel_err(out, ref) < 1e-5
    assert rel_err(zd.grad, zr.grad) < 1e-5
    assert rel_err(ld.grad, lr.grad) < 1e-5


@pytest.mark.parametrize('M', [7, 64, 256])
def test_reparam_logq_fwd_bwd(device, M):
    """M >= 64 over a few rows: the backward's sixteen-waves-per-row variant (the feature models' 80 rows x 256)."""
    from cmvae_amd import ops
    g = torch.Generator().manual_seed(13)
    mean = torch.randn(3, 5, 64, generator=g)
    lv = torch.randn(3, 5, 64, generator=g) * 0.5
    eps = torch.randn(3, 5, M, 64, generator=g)
    gz = torch.randn(3, 5, M, 64, generator=g)
    gq = torch.randn(3, 5, M, generator=g)
    mr = mean.clone().double().requires_grad_(True)
    lr = lv.clone().double().requires_grad_(True)
    zr = eps.double() * (0.5 * lr).exp()[:, :, None] + mr[:, :, None]
    qr = (-0.5 * (lr[:, :, None] + (zr - mr[:, :, None]).pow(2) / lr.exp()[:, :, None])
          - 0.5 * np.log(2 * np.pi)).sum(-1)
    ((zr * gz.double()).sum() + (qr * gq.double()).sum()).backward()
    md = mean.to(device).requires_grad_(True)
    ld = lv.to(device).requires_grad_(True)
    z, q = ops.reparam_logq(md, ld, eps.to(device))
    ((z * gz.to(device)).sum() + (q * gq.to(device)).sum()).backward()
    assert rel_err(z, zr) < 1e-6 and rel_err(q, qr) < 1e-5
    assert rel_err(md.grad, mr.grad) < 1e-5 and rel_err(ld.grad, lr.grad) < 1e-5


@pytest.mark.parametrize('scale', [0.0, 0.03, 0.3, 0.5])
def test_expm_trace(device, scale):
    """h and dh/dA against fp64 torch.matrix_exp (the fp32 oracle itself loses ~1e-5 absolute to
    the `trace - 64` cancellation, SURVEY.md §7 'h cancellation')."""
    from cmvae_amd import ops
    g = torch.Generator().manual_seed(17)
    A = torch.randn(64, 64, generator=g) * scale
    Ar = A.clone().double().requires_grad_(True)
    href = torch.trace(torch.matrix_exp(Ar * Ar)) - 64
    href.backward()
    Ad = A.to(device).requires_grad_(True)
    h = ops.expm_trace(Ad)
    h.backward()
    assert abs(float(h.detach()) - float(href.detach())) <= 1e-5 * max(abs(float(href.detach())), 1e-12) + 1e-12
    if scale > 0:
        assert rel_err(Ad.grad, Ar.grad) < 1e-5
    else:
        assert float(Ad.grad.abs().max()) == 0.0
    # the fp32 oracle agrees to its own rounding
    h32 = torch.trace(torch.matrix_exp(A * A)) - 64
    assert abs(float(h.detach()) - float(h32)) <= 1e-4 * abs(float(href.detach())) + 64 * 6e-8 * 8


@pytest.mark.parametrize('B,S,M,K', [(4, 300, 32, 20), (2, 7, 3, 5), (1, 1, 1, 2)])
def test_elbo_tail_and_prior_logw(device, B, S, M, K):
    """The fused scalar tail of forward (omniglot/model.py:481-491, 496-516) against the reference's own chain of
    tensor expressions in float64, values and every gradient."""
    from cmvae_amd import ops
    g = torch.Generator().manual_seed(B * S + K)
    D = 64
    log_q = torch.randn(B, S, M, generator=g) * 3 - 80
    log_p = torch.randn(B, S * M, generator=g) * 3 - 90
    q_eps = torch.randn(B, S * M, D, generator=g)
    A = torch.randn(D, D, generator=g) * 0.05
    A[3, 4] = 0.0
    pi = torch.softmax(torch.randn(B, K, generator=g), -1)
    trace = torch.rand(B, K, generator=g) + 0.5
    x = torch.randn(B, K, D, generator=g)
    h_lambda, h_c, a_lambda = 0.7, 3.0, 0.01
    c = -0.5 * np.log(2 * np.pi)

    def leaves(dtype, dev):
        return [t.to(dtype).to(dev).requires_grad_(True) for t in (log_q, log_p, q_eps, A, x)]

    rq, rp, re, rA, rx = leaves(torch.float64, 'cpu')
    hr = (torch.matrix_exp(rA * rA).diagonal().sum() - D)
    kl_r = rq.mean() - rp.mean() - (-0.5 * re.pow(2) + c).sum(-1).mean()
    ha_r = h_lambda * hr + 0.5 * h_c * hr * hr + a_lambda * rA.abs().sum()
    lw_r = torch.log(pi.double()) + (-0.5 * (rx - torch.log(trace.double())[:, :, None]).pow(2) + c).sum(-1)
    gw = torch.randn(B, K, generator=g)
    (2.0 * kl_r + 0.5 * ha_r + (lw_r * gw.double()).sum()).backward()

    dq, dp, de, dA, dx = leaves(torch.float32, device)
    hd = ops.expm_trace(dA)
    kl_d, ha_d = ops.elbo_tail(dq, dp, de, hd, dA, h_lambda, h_c, a_lambda)
    lw_d = ops.prior_logw(pi.to(device), trace.to(device), dx)
    (2.0 * kl_d + 0.5 * ha_d + (lw_d * gw.to(device)).sum()).backward()
    assert abs(float(kl_d.detach()) - float(kl_r.detach())) < 1e-5 * abs(float(kl_r.detach()))
    assert abs(float(ha_d.detach()) - float(ha_r.detach())) < 1e-5 * abs(float(ha_r.detach()))
    assert rel_err(lw_d.detach(), lw_r.detach().float()) < 1e-6
    for a, r in ((dq, rq), (dp, rp), (de, re), (dA, rA), (dx, rx)):
        assert rel_err(a.grad, r.grad.float()) < 1e-5
    # only one of the two outputs used: the other's gradient is taken as zero
    dq2, dp2, de2, dA2, _ = leaves(torch.float32, device)
    kl2, _ = ops.elbo_tail(dq2, dp2, de2, ops.expm_trace(dA2), dA2, h_lambda, h_c, a_lambda)
    kl2.backward()
    assert float(dA2.grad.abs().max()) == 0.0 and rel_err(de2.grad, re.grad.float() / 2.0) < 1e-5


def test_missing_gpu_tensor_raises(device):
    from cmvae_amd import ops
    with pytest.raises(RuntimeError):
        ops.mixture_loglik(torch.zeros(1, 4, 64), torch.zeros(1, 2, 64), torch.zeros(1, 2))


def test_empty_episodes_are_rejected_not_launched(device):
    """Regression for the round-1 SIGFPE (gpurun_out/t4.log): S=5, K=5 leaves no query items (Nu = 0).  Both the
    Python shims and the C-ABI entry points must refuse N == 0 / Nu == 0 / Ns == 0 with an error, not crash."""
    from cmvae_amd import ops
    from cmvae_amd._lib import lib
    model, _ = build_pair('feat', K=5, linear=True)
    c = _causal_desc(model, device)
    z0 = torch.zeros(2, 0, 64, device=device)
    z = torch.randn(2, 10, 64, device=device)
    y = torch.zeros(2, 10, dtype=torch.int64, device=device)
    with pytest.raises(RuntimeError):
        ops.semi_prior(z0, z, y, 5, c, 2)
    with pytest.raises(RuntimeError):
        ops.semi_prior(z, z0, torch.zeros(2, 0, dtype=torch.int64, device=device), 5, c, 2)
    with pytest.raises(RuntimeError):
        ops.unsup_prior(z0, torch.zeros(2, 5, dtype=torch.int64, device=device), c, 2)
    with pytest.raises(RuntimeError):
        ops.unsup_prior(z[:, :3].contiguous(), torch.zeros(2, 5, dtype=torch.int64, device=device), c, 2)   # K > N
    # straight through the C-ABI: error code and message, nothing launched
    out = [torch.full((2, 5), 3.0, device=device) for _ in range(3)]
    mu, lv = torch.full((2, 5, 64), 3.0, device=device), torch.full((2, 5, 64), 3.0, device=device)
    ws = torch.empty(lib.cmvae_semi_prior_workspace_bytes(2, 0, 10, 5), dtype=torch.uint8, device=device)
    rc = lib.cmvae_semi_prior(z.data_ptr(), z.data_ptr(), y.data_ptr(), 2, 0, 10, 5, c.ref, 2, 1.0, ws.data_ptr(),
                              ws.numel(), out[0].data_ptr(), mu.data_ptr(), lv.data_ptr(), out[1].data_ptr(),
                              out[2].data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b'bad shape' in lib.cmvae_last_error()
    idx = torch.zeros(2, 5, dtype=torch.int64, device=device)
    rc = lib.cmvae_unsup_prior(z.data_ptr(), 2, 0, 5, idx.data_ptr(), c.ref, 2, 0, 1.0, ws.data_ptr(), ws.numel(),
                               out[0].data_ptr(), mu.data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
    assert rc != 0
    torch.cuda.synchronize()
    assert float(mu.min()) == 3.0
    # a model-level call with no query items says so as well
    with pytest.raises(RuntimeError):
        model.to(device).eval()
        H = torch.relu(torch.randn(2, 5, 256, device=device))
        model.prediction(H, torch.arange(5, device=device)[None].repeat(2, 1), H[:, :0])


def test_out_of_range_seeds_and_labels(device):
    """init_idx >= N is clamped inside the kernel (never an out-of-bounds read) and rejected on the host when the
    indices come from the CPU; labels outside [0, K) are rejected on the host path."""
    from cmvae_amd import ops
    model, _ = build_pair('feat', K=3, linear=True)
    c = _causal_desc(model, device)
    model.to(device)
    z = torch.randn(2, 50, 64, device=device)
    bad = torch.tensor([[0, 1, 50], [3, -1, 7]], dtype=torch.int64)
    pi, mu, tau, logw = ops.unsup_prior(z, bad.to(device), c, 1)             # device tensor: clamped, finite
    assert torch.isfinite(mu).all()
    with pytest.raises(RuntimeError):
        model._unsup_prior(z, bad.numpy())                                    # host indices: checked before upload
    with pytest.raises(RuntimeError):
        model._semi_prior(z, z, torch.full((2, 50), 3, dtype=torch.int64))    # label 3 with K = 3
    old = ops.CHECK_INDICES
    ops.CHECK_INDICES = True
    try:
        with pytest.raises(RuntimeError):
            ops.unsup_prior(z, bad.to(device), c, 1)
        with pytest.raises(RuntimeError):
            ops.semi_prior(z, z, torch.full((2, 50), 3, dtype=torch.int64, device=device), 3, c, 1)
    finally:
        ops.CHECK_INDICES = old
