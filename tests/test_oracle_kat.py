"""CPU: the oracle against analytic known answers and against the committed
golden vectors.  The reference has no tests or fixtures and cannot be run here, so these are the pins the
oracle has: "parity unpinned" with respect to the reference, pinned analytically and for regression."""
import math
import os

import numpy as np
import pytest
import scipy.linalg
import torch
import torch.nn.functional as F

from oracle import gmvae_oracle as O
from tests.helpers import build_pair

C = -0.5 * math.log(2 * math.pi)


def test_kat1_glp_at_mean():
    x = torch.randn(5, 64)
    assert torch.allclose(O.glp(x, x), torch.full((5,), 64 * C), atol=1e-5)
    assert abs(64 * C - (-58.81206612)) < 1e-6


def test_kat2_h_of_acyclic_and_2x2():
    assert float(O.h_a(torch.zeros(64, 64), 64)) == 0.0
    A = torch.triu(torch.randn(64, 64), diagonal=1)
    assert abs(float(O.h_a(A.double(), 64))) < 1e-9
    a, b = 0.7, -1.3
    A2 = torch.tensor([[0.0, a], [b, 0.0]], dtype=torch.float64)
    # tr(expm([[0,a^2],[b^2,0]])) = 2 cosh(|ab|)
    assert abs(float(O.h_a(A2, 2)) - (2 * math.cosh(abs(a * b)) - 2)) < 1e-12
    A3 = torch.randn(64, 64, dtype=torch.float64) * 0.1
    ref = np.trace(scipy.linalg.expm((A3 * A3).numpy())) - 64
    assert abs(float(O.h_a(A3, 64)) - ref) < 1e-9


def test_kat3_linear_zero_adjacency_halves_the_mean():
    _, oracle = build_pair('feat', K=4, linear=True)
    oracle.P['causal.A'].zero_()
    x = torch.randn(7, 64)
    assert torch.equal(oracle._causal(x), x)
    z = torch.randn(2, 50, 64)
    idx = np.array([[0, 1, 2, 3], [4, 5, 6, 7]])
    traj = []
    oracle.unsup_prior(z, idx, iters=1, trajectory=traj)
    pi0, mu0, _ = traj[0]
    assert torch.allclose(mu0[0, 0], 0.5 * z[0, 0], atol=1e-6)           # Minv = I/2
    # one iteration by hand
    logw = torch.log(pi0) + (-0.5 * mu0.pow(2) + C).sum(-1)               # prior term: -1/2 |mu|^2 + D c
    ll = (-0.5 * (z[:, :, None] - mu0[:, None]).pow(2) + C).sum(-1) + logw[:, None]
    r = torch.softmax(ll, -1)
    mean = 0.5 * torch.einsum('bnk,bnd->bkd', r, z) / r.sum(1)[:, :, None]
    assert torch.allclose(traj[1][1], mean, atol=1e-5)
    assert torch.allclose(traj[1][0], r.sum(1) / 50, atol=1e-6)


def test_kat4_single_component():
    _, oracle = build_pair('feat', K=1, linear=True)
    z = torch.randn(2, 30, 64)
    pi, mu, tau = oracle.unsup_prior(z, np.array([[3], [4]]), iters=2)
    assert torch.allclose(pi, torch.ones(2, 1))
    assert torch.allclose(tau, torch.ones(2, 1))


def test_kat5_bce_half_and_clamp():
    p = torch.full((1, 784), 0.5)
    x = (torch.rand(1, 784) < 0.3).float()
    assert abs(float(F.binary_cross_entropy(p, x, reduction='sum')) - 784 * math.log(2)) < 1e-3
    assert float(F.binary_cross_entropy(torch.zeros(1), torch.ones(1), reduction='sum')) == 100.0


def test_kat6_sab_scale_and_residual():
    d = 8
    P = {'s.mab.fc_%s.weight' % n: torch.eye(d) for n in 'qkvo'}
    P.update({'s.mab.fc_%s.bias' % n: torch.zeros(d) for n in 'qkvo'})
    X = torch.randn(1, 5, d)
    out = O.mab(X, P, 's', heads=2)
    hd = d // 2
    q = torch.cat(X.split(hd, 2), 0)
    att = torch.softmax(q.bmm(q.transpose(1, 2)) / math.sqrt(d), 2)      # 1/sqrt(dim_V), not 1/sqrt(head dim)
    o = torch.cat((q + att.bmm(q)).split(1, 0), 2)
    assert torch.allclose(out, o + F.relu(o), atol=1e-6)


def test_kat7_nonlinear_get_w():
    model, oracle = build_pair('omniglot', linear=False)
    W = O.causal_get_w(oracle.P, False)
    fc1 = oracle.P['causal.fc1.weight']
    i, j = 5, 9
    assert abs(float(W[i, j]) - float(fc1[32 * j:32 * j + 32, i].pow(2).sum())) < 1e-6
    assert float(W.min()) >= 0.0
    assert torch.allclose(model.causal.get_w(), W)


def test_kat8_semi_em_keeps_pi_and_uses_shrunk_mean():
    _, oracle = build_pair('feat', K=2, linear=True)
    oracle.P['causal.A'].zero_()
    g = torch.Generator().manual_seed(0)
    centres = torch.tensor([[-20.0], [20.0]]).repeat(1, 64)
    lab = torch.arange(40) % 2
    z = (centres[lab] + 0.1 * torch.randn(40, 64, generator=g))[None]
    uz, sz, y = z[:, :30], z[:, 30:], F.one_hot(lab[30:], 2).float()[None]
    traj = []
    oracle.semi_prior(uz, sz, y, iters=1, trajectory=traj)
    pi, mu, lv, _ = traj[1]
    assert torch.allclose(pi, torch.full((1, 2), 0.5))                    # never updated
    # far-apart clusters: hard responsibilities; variance is the second moment about the SHRUNK mean
    for k in range(2):
        pts = z[0][lab == k]
        m = 0.5 * pts.mean(0)                                             # Minv = I/2 at logvar 0
        assert torch.allclose(mu[0, k], m, atol=1e-4)
        var = (pts - m).pow(2).mean(0)
        assert torch.allclose(lv[0, k].exp(), var, rtol=1e-4)


def test_kat9_zero_responsibility_guard():
    _, oracle = build_pair('feat', K=2, linear=True)
    X = torch.zeros(1, 4, 64)
    pi = torch.tensor([[1.0, 0.0]])           # log(0) = -inf: component 1 gets exactly zero responsibility
    mean = torch.zeros(1, 2, 64)
    pi2, mean2, _ = oracle.unsup_params(X, (pi, mean, torch.ones(1, 2)))
    assert torch.isfinite(mean2).all() and torch.isfinite(pi2).all()
    assert float(mean2[0, 1].abs().max()) == 0.0
    assert float(pi2[0, 1]) < 1e-20


@pytest.mark.parametrize('name', ['omniglot_nonlinear', 'omniglot_linear', 'feat_linear', 'feat_nonlinear'])
def test_golden_vectors(name):
    from tools.make_golden import CASES, run_case
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', name + '.npz'))
    got = run_case(CASES[name])
    for k in gold.files:
        if k == 'y_pred':
            assert np.array_equal(gold[k], got[k])
        else:
            np.testing.assert_allclose(got[k], gold[k], rtol=2e-4, atol=1e-5, err_msg=k)


def test_fp64_oracle_agrees_with_fp32():
    _, o32 = build_pair('omniglot', K=3, M=4, iters=3)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in o32.P.items()}
    o64 = O.OracleGMVAE(P64, 'omniglot', component_size=3, train_mc_sample_size=4, test_mc_sample_size=4,
                        unsupervised_em_iters=3, semisupervised_em_iters=3)
    g = torch.Generator().manual_seed(1)
    X = (torch.rand(2, 6, 1, 28, 28, generator=g) < 0.15).float()
    eps = torch.randn(2, 6, 4, 64, generator=g)
    idx = np.array([[0, 5, 9], [1, 2, 3]])
    a = o32.forward(X, 0.1, 0.1, 1e-4, eps=eps, init_idx=idx)
    b = o64.forward(X.double(), 0.1, 0.1, 1e-4, eps=eps.double(), init_idx=idx)
    for i in (1, 2, 3):
        assert abs(float(a[i]) - float(b[i])) <= 1e-4 * abs(float(b[i])) + 1e-7


# --------------------------------------------------------------------------- KATs added in round 2: each pins a piece of
# the oracle against arithmetic written out by hand (loops / einsum on explicit indices), independent of the ATen
# op the oracle calls.  The oracle stays "parity unpinned" (no reference fixtures exist); these harden it.
def _causal_by_hand(x, P):
    """x - LC(ELU(fc1 x)) with explicit loops over the 64 output dims and 32 hidden units per dim
    (omniglot/module.py:181-187, 273-280): hidden unit 32*j + m belongs to output dim j."""
    W1, b1 = P['causal.fc1.weight'].double(), P['causal.fc1.bias'].double()
    w2, b2 = P['causal.fc2.0.weight'].double(), P['causal.fc2.0.bias'].double()
    x = x.double()
    out = torch.empty_like(x)
    for r in range(x.shape[0]):
        for j in range(64):
            acc = 0.0
            for m in range(32):
                u = 32 * j + m
                pre = float(W1[u] @ x[r]) + float(b1[u])
                acc += (pre if pre > 0 else math.expm1(pre)) * float(w2[j, m, 0])
            out[r, j] = x[r, j] - (acc + float(b2[j, 0]))
    return out


def test_kat10_nonlinear_causal_rows_by_hand():
    _, oracle = build_pair('omniglot', linear=False)
    x = torch.randn(3, 64, generator=torch.Generator().manual_seed(0))
    assert torch.allclose(oracle._causal(x).double(), _causal_by_hand(x, oracle.P), atol=2e-6)


def test_kat11_nonlinear_unsup_iteration_by_hand():
    """One E/M iteration of the unsupervised EM with the NOTEARS-MLP causal layer (omniglot/model.py:168-222),
    re-derived in fp64 with explicit sums: responsibilities, guarded N_k, pi, and mean = (r^T z / N_k) Minv with
    Minv = (I + C C^T)^-1, C row i = causal(e_i)."""
    _, oracle = build_pair('omniglot', K=3, linear=False)
    B, N, K, D = 1, 12, 3, 64
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, N, D, generator=g)
    pi = torch.tensor([[0.2, 0.5, 0.3]])
    mu = torch.randn(B, K, D, generator=g) * 0.5
    pi2, mu2, tau2 = oracle.unsup_params(z, (pi, mu, torch.ones(B, K)))
    eps_mu = _causal_by_hand(mu[0], oracle.P)
    ll = torch.empty(N, K, dtype=torch.float64)
    for n in range(N):
        for k in range(K):
            ll[n, k] = (-0.5 * (z[0, n].double() - mu[0, k].double()).pow(2) + C).sum() + math.log(float(pi[0, k])) \
                + (-0.5 * eps_mu[k].pow(2) + C).sum()                      # glp(causal(mu_k), log tau = 0)
    r = torch.softmax(ll, -1)
    Nk = r.sum(0)
    Cm = _causal_by_hand(torch.eye(64), oracle.P)                          # row i = causal(e_i)
    Minv = torch.linalg.inv(torch.eye(64, dtype=torch.float64) + Cm @ Cm.T)
    mean = ((r.T @ z[0].double()) / Nk[:, None]) @ Minv
    assert torch.allclose(pi2[0].double(), Nk / Nk.sum(), atol=1e-6)
    assert torch.allclose(mu2[0].double(), mean, atol=5e-6)
    assert torch.equal(tau2, torch.ones(B, K))


def test_kat12_semi_iteration_by_hand():
    """One semi-supervised iteration, linear causal layer with a non-trivial A (omniglot/model.py:224-330): E-step on
    the unlabeled points only with unit variance, joint denominator, per-component inverse from the incoming
    logvar, variance about the shrunk mean, pi untouched."""
    _, oracle = build_pair('feat', K=2, linear=True)
    A = oracle.P['causal.A'].double()
    g = torch.Generator().manual_seed(4)
    Nu, Ns, D = 9, 4, 64
    uz, sz = torch.randn(1, Nu, D, generator=g), torch.randn(1, Ns, D, generator=g)
    lab = torch.tensor([0, 1, 1, 0])
    y = F.one_hot(lab, 2).float()[None]
    pi = torch.full((1, 2), 0.5)
    mu = torch.randn(1, 2, D, generator=g) * 0.3
    lv = torch.randn(1, 2, D, generator=g) * 0.2
    pi2, mu2, lv2, _ = oracle.semi_params(uz, sz, y, (pi, mu, lv, torch.ones(1, 2)))
    I = torch.eye(D, dtype=torch.float64)
    causal = lambda x: x - x @ A                                           # noqa: E731
    ll = torch.stack([(-0.5 * (uz[0].double() - mu[0, k].double()).pow(2) + C).sum(-1) + math.log(0.5)
                      + (-0.5 * causal(mu[0, k].double()).pow(2) + C).sum() for k in range(2)], -1)
    r = torch.softmax(ll, -1)
    for k in range(2):
        S = torch.diag((0.5 * lv[0, k].double()).exp())
        Cm = causal(S)                                                     # rows of diag(std): causal(s_i e_i)
        Minv = torch.linalg.inv(I + Cm @ Cm.T)
        sup = sz[0][lab == k].double()
        Nk = sup.shape[0] + r[:, k].sum()
        m = ((sup.sum(0) + r[:, k] @ uz[0].double()) / Nk) @ Minv
        var = ((sup - m).pow(2).sum(0) + (r[:, k, None] * (uz[0].double() - m).pow(2)).sum(0)) / Nk
        assert torch.allclose(mu2[0, k].double(), m, atol=5e-6)
        assert torch.allclose(lv2[0, k].double(), var.log(), atol=2e-5)
    assert torch.equal(pi2, pi)


def test_kat13_batchnorm_training_statistics_and_running_update():
    """_bn in training mode: biased batch variance for the normalisation, unbiased variance and momentum 0.1 for the
    running statistics (nn.BatchNorm2d, omniglot/module.py:15), written out by hand."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 64, 3, 3, generator=g) * 2 + 1
    P = {'bn.weight': torch.rand(64, generator=g) + 0.5, 'bn.bias': torch.randn(64, generator=g),
         'bn.running_mean': torch.randn(64, generator=g), 'bn.running_var': torch.rand(64, generator=g) + 0.5}
    running = {}
    y = O._bn(x, P, 'bn', True, running)
    xd = x.double()
    n = 6 * 9
    mean = xd.sum((0, 2, 3)) / n
    var_b = (xd - mean.view(1, -1, 1, 1)).pow(2).sum((0, 2, 3)) / n
    ref = (xd - mean.view(1, -1, 1, 1)) / (var_b + 1e-5).sqrt().view(1, -1, 1, 1) * P['bn.weight'].double().view(1, -1, 1, 1) \
        + P['bn.bias'].double().view(1, -1, 1, 1)
    assert torch.allclose(y.double(), ref, atol=1e-5)
    assert torch.allclose(running['bn.running_mean'].double(), 0.9 * P['bn.running_mean'].double() + 0.1 * mean, atol=1e-6)
    assert torch.allclose(running['bn.running_var'].double(),
                          0.9 * P['bn.running_var'].double() + 0.1 * var_b * n / (n - 1), atol=1e-6)
    # evaluation mode normalises with the running statistics and leaves them alone
    ye = O._bn(x, P, 'bn', False)
    refe = (xd - P['bn.running_mean'].double().view(1, -1, 1, 1)) / (P['bn.running_var'].double() + 1e-5).sqrt().view(1, -1, 1, 1) \
        * P['bn.weight'].double().view(1, -1, 1, 1) + P['bn.bias'].double().view(1, -1, 1, 1)
    assert torch.allclose(ye.double(), refe, atol=1e-5)


@pytest.mark.parametrize('k,Hi', [(4, 2), (3, 4), (4, 7)])
def test_kat14_conv_transpose_by_scatter_loop(k, Hi):
    """ConvTranspose2d(stride 2, padding 1) as the definition: every input pixel scatters w[ci,co,ky,kx] to output
    (2*iy - 1 + ky, 2*ix - 1 + kx); output size (Hi-1)*2 - 2 + k (omniglot/module.py:50-62: 2->4->7->14->28)."""
    g = torch.Generator().manual_seed(k + Hi)
    ci_n, co_n = 5, 4
    x = torch.randn(2, ci_n, Hi, Hi, generator=g, dtype=torch.float64)
    w = torch.randn(ci_n, co_n, k, k, generator=g, dtype=torch.float64)
    b = torch.randn(co_n, generator=g, dtype=torch.float64)
    Ho = (Hi - 1) * 2 - 2 + k
    out = b.view(1, -1, 1, 1).repeat(2, 1, Ho, Ho).clone()
    for iy in range(Hi):
        for ix in range(Hi):
            for ky in range(k):
                for kx in range(k):
                    oy, ox = 2 * iy - 1 + ky, 2 * ix - 1 + kx
                    if 0 <= oy < Ho and 0 <= ox < Ho:
                        out[:, :, oy, ox] += torch.einsum('nc,cd->nd', x[:, :, iy, ix], w[:, :, ky, kx])
    assert torch.allclose(F.conv_transpose2d(x, w, b, stride=2, padding=1), out, atol=1e-10)
    assert Ho == {2: 4, 4: 7, 7: 14}[Hi]


def test_kat15_encoder_block_by_hand():
    """Conv2d(3x3, pad 1) by explicit loops, then MaxPool2d(2, ceil_mode=True) on an odd size (7 -> 4): the last
    window of each row / column holds a single row / column (omniglot/module.py:14-17)."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 2, 7, 7, generator=g, dtype=torch.float64)
    w = torch.randn(3, 2, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(3, generator=g, dtype=torch.float64)
    y = torch.zeros(1, 3, 7, 7, dtype=torch.float64)
    for co in range(3):
        for oy in range(7):
            for ox in range(7):
                acc = float(b[co])
                for ci in range(2):
                    for ky in range(3):
                        for kx in range(3):
                            iy, ix = oy + ky - 1, ox + kx - 1
                            if 0 <= iy < 7 and 0 <= ix < 7:
                                acc += float(x[0, ci, iy, ix]) * float(w[co, ci, ky, kx])
                y[0, co, oy, ox] = acc
    assert torch.allclose(F.conv2d(x, w, b, stride=1, padding=1), y, atol=1e-10)
    pooled = F.max_pool2d(F.relu(y), 2, ceil_mode=True)
    assert tuple(pooled.shape) == (1, 3, 4, 4)
    for co in range(3):
        for py in range(4):
            for px in range(4):
                win = F.relu(y)[0, co, 2 * py:min(2 * py + 2, 7), 2 * px:min(2 * px + 2, 7)]
                assert float(pooled[0, co, py, px]) == float(win.max())


def test_kat16_forward_loss_assembly():
    """kl = mean(log q) - mean(log p_z) - mean(log p_eps); h_aug = h_lambda h + 0.5 h_c h^2 + a_lambda |A|_1; rec =
    BCE_sum / (B S M) — recomputed from the tensors the oracle exposes (omniglot/model.py:463-466, 511-516)."""
    _, oracle = build_pair('omniglot', K=3, M=4, iters=2)
    g = torch.Generator().manual_seed(1)
    X = (torch.rand(2, 6, 1, 28, 28, generator=g) < 0.15).float()
    eps = torch.randn(2, 6, 4, 64, generator=g)
    ex = {}
    with torch.no_grad():
        X_rec, rec, kl, h_aug, h = oracle.forward(X, 0.3, 0.7, 1e-2, eps=eps, init_idx=np.array([[0, 5, 9], [1, 2, 3]]),
                                                  extras=ex)
    kl_hand = ex['log_q_z'].double().mean() - ex['log_p_z'].double().mean() - ex['log_p_eps'].double().mean()
    assert abs(float(kl) - float(kl_hand)) < 1e-4 * abs(float(kl_hand))
    A = ex['A'].double()
    h_hand = np.trace(scipy.linalg.expm((A * A).numpy())) - 64
    assert abs(float(h) - h_hand) < 1e-4 * abs(h_hand) + 64 * 6e-8 * 8
    assert abs(float(h_aug) - (0.3 * h_hand + 0.5 * 0.7 * h_hand ** 2 + 1e-2 * float(A.abs().sum()))) < 1e-4 * float(h_aug)
    t = X[:, :, None].repeat(1, 1, 4, 1, 1, 1).double()
    p = X_rec.double()
    bce = -(t * p.log().clamp_min(-100) + (1 - t) * (1 - p).log().clamp_min(-100)).sum() / (2 * 6 * 4)
    assert abs(float(rec) - float(bce)) < 1e-5 * float(bce)
    # mixture log-likelihood of one latent by hand
    q_z, mu_k, pi_k = ex['q_z'].double(), ex['mu_k'].double(), ex['pi_k'].double()
    eps_mu = oracle._causal(ex['mu_k']).double()
    b_, n_ = 1, 7
    terms = [(-0.5 * (q_z[b_, n_] - mu_k[b_, k]).pow(2) + C).sum() + pi_k[b_, k].log() + (-0.5 * eps_mu[b_, k].pow(2) + C).sum()
             for k in range(3)]
    assert abs(float(torch.logsumexp(torch.stack(terms), 0)) - float(ex['log_p_z'].reshape(2, -1)[b_, n_])) < 1e-3
