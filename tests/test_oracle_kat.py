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
