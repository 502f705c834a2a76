"""Model-level parity on the GPU: GMVAE.forward (ELBO terms + parameter gradients) and
GMVAE.prediction (bit-exact labels) against the CPU oracle on identical weights, inputs,
reparametrisation noise and EM seeds.  Tolerance: 1e-4 relative (north_star)."""
import numpy as np
import pytest
import torch

from tests.helpers import build_pair, omniglot_batch, feature_batch, rel_err, RTOL

pytestmark = pytest.mark.gpu


def _run_forward_pair(model, oracle, X, eps, idx, device, hl=0.1, hc=0.1, al=1e-4):
    # oracle (CPU) with autograd on a private trainable copy of the weights
    from oracle.gmvae_oracle import make_trainable
    oracle.P = make_trainable(oracle.P)
    ex_o = {}
    out_o = oracle.forward(X, hl, hc, al, eps=eps, init_idx=idx, extras=ex_o)
    (out_o[1] + out_o[2] + out_o[3]).backward()
    model.to(device).train()
    ex_g = {}
    out_g = model(X.to(device), hl, hc, al, eps=eps.to(device), init_idx=torch.from_numpy(idx).to(device), extras=ex_g)
    (out_g[1] + out_g[2] + out_g[3]).backward()
    return out_o, ex_o, out_g, ex_g


def _check_forward(model, oracle, out_o, ex_o, out_g, ex_g):
    names = ['rec_loss', 'kl_loss', 'h_augment']
    for i, n in enumerate(names, start=1):
        a, b = float(out_g[i]), float(out_o[i])
        assert abs(a - b) <= RTOL * abs(b) + 1e-7, '%s: gpu %r oracle %r' % (n, a, b)
    # bare h suffers the oracle's own `trace - 64` cancellation: compare at the fp32-ulp(64) floor
    assert abs(float(out_g[4]) - float(out_o[4])) <= RTOL * abs(float(out_o[4])) + 64 * 6e-8 * 8
    assert rel_err(out_g[0], out_o[0]) < RTOL                               # reconstructions
    for key in ('q_z_mean', 'q_z_logvar', 'mu_k', 'pi_k', 'log_p_z', 'log_q_z', 'log_p_eps'):
        assert rel_err(ex_g[key].reshape(-1), ex_o[key].reshape(-1)) < RTOL, key
    # parameter gradients.  Some are analytically zero (conv bias in front of BatchNorm, fc_k bias under
    # softmax): compare every tensor against the scale of the whole gradient, not only its own norm.
    gscale = max(float(g.abs().max()) for g in (oracle.P[n].grad for n, _ in model.named_parameters())
                 if g is not None)
    bad = []
    for name, p in model.named_parameters():
        g_o = oracle.P[name].grad
        if g_o is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        err = float((p.grad.detach().cpu().double() - g_o.double()).norm())
        # gradients pass through BatchNorm over tiny batches: looser than the forward bar
        tol = 2e-3 * float(g_o.double().norm()) + 1e-6 * gscale * np.sqrt(g_o.numel())
        if err > tol:
            bad.append((name, err, tol))
    assert not bad, bad


@pytest.mark.parametrize('linear', [False, True])
def test_omniglot_forward_backward(device, linear):
    B, S, M, K = 2, 6, 4, 3
    model, oracle = build_pair('omniglot', K=K, M=M, iters=3, linear=linear)
    X = omniglot_batch(B, S, seed=1)
    g = torch.Generator().manual_seed(2)
    eps = torch.randn(B, S, M, 64, generator=g)
    rng = np.random.RandomState(3)
    idx = np.stack([rng.choice(S * M, K, replace=False) for _ in range(B)], 0)
    res = _run_forward_pair(model, oracle, X, eps, idx, device)
    _check_forward(model, oracle, *res)


@pytest.mark.parametrize('linear', [True, False])
def test_feature_forward_backward(device, linear):
    B, S, M, K = 3, 5, 16, 5
    model, oracle = build_pair('feat', K=K, M=M, iters=3, linear=linear, feat=256)
    H = feature_batch(B, S, 256, seed=4)
    g = torch.Generator().manual_seed(5)
    eps = torch.randn(B, S, M, 64, generator=g)
    rng = np.random.RandomState(6)
    idx = np.stack([rng.choice(S * M, K, replace=False) for _ in range(B)], 0)
    res = _run_forward_pair(model, oracle, H, eps, idx, device, al=1e-3)
    _check_forward(model, oracle, *res)


@pytest.mark.parametrize('variant,linear,K,shot,query,M', [('omniglot', False, 5, 1, 3, 8),
                                                            ('omniglot', False, 20, 5, 2, 4),
                                                            ('feat', True, 5, 5, 15, 32),
                                                            ('feat', False, 5, 1, 15, 16)])
def test_prediction_bit_exact(device, variant, linear, K, shot, query, M):
    B = 2
    model, oracle = build_pair(variant, K=K, M=M, iters=4, linear=linear)
    tr, te = K * shot, K * query
    if variant == 'omniglot':
        X = omniglot_batch(B, tr + te, seed=7)
    else:
        X = feature_batch(B, tr + te, 256, seed=7)
    y_tr = torch.arange(K).repeat_interleave(shot)[None].repeat(B, 1)
    g = torch.Generator().manual_seed(8)
    eps = torch.randn(B, tr + te, M, 64, generator=g)
    oracle.training = False
    ex_o, ex_g = {}, {}
    y_o = oracle.prediction(X[:, :tr], y_tr, X[:, tr:], eps=eps, extras=ex_o)
    model.to(device).eval()
    y_g = model.prediction(X[:, :tr].to(device), y_tr.to(device), X[:, tr:].to(device), eps=eps.to(device),
                           extras=ex_g)
    assert y_g.dtype == torch.int64 and tuple(y_g.shape) == (B, te)
    assert rel_err(ex_g['mu_k'], ex_o['mu_k']) < RTOL
    assert rel_err(ex_g['logvar_k'].exp(), ex_o['logvar_k'].exp()) < RTOL
    assert rel_err(ex_g['post_mean'], ex_o['post_mean']) < 10 * RTOL
    assert torch.equal(y_g.cpu(), y_o), 'few-shot assignments must be bit-exact'


def test_numpy_stream_matches_reference_order(device):
    """With no seeds supplied, forward draws np.random.choice once per episode in episode order."""
    B, S, M, K = 2, 6, 4, 3
    model, oracle = build_pair('omniglot', K=K, M=M, iters=1)
    model.to(device).train()
    X = omniglot_batch(B, S, seed=1).to(device)
    np.random.seed(123)
    expect = np.stack([np.random.choice(S * M, K, replace=False) for _ in range(B)], 0)
    after = np.random.rand()
    np.random.seed(123)
    got = model._draw_init_idx(B, S * M, device).cpu().numpy()
    assert np.array_equal(got, expect)
    assert np.random.rand() == after
