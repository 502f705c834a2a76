"""End-to-end parity against the CPU oracle at BASELINE.json's own shapes:

  * Omniglot 20-way training step — B=4, S=300, M=32 (N=9600), K=20, 10 EM iterations, NOTEARS-MLP causal layer:
    ELBO terms, EM parameters, reconstructions and every parameter gradient (omniglot/model.py:431-518);
  * Omniglot 20-way `prediction`, 1-shot and 5-shot, query 15 (omniglot/model.py:520-585): labels bit for bit;
  * Mini-ImageNet training step at the real shape — B=16, S=5, M=256, F=256, K=5, linear causal layer
    (mimgnet/main.py:28-45, mimgnet/model.py:364-440);
  * CelebA evaluation call at the sweep shape — 5-way 5-shot, 75 queries x 256 MC samples (N_u = 19200, N_s = 6400),
    NOTEARS-MLP causal layer (celeba/model.py:22, mimgnet/model.py:442-511).

Both sides get identical weights, inputs, reparametrisation noise and EM seeds.  Tolerance: 1e-4 relative on ELBO
terms and EM parameters (north_star), 5e-5 on per-tensor gradients (norm-wise, against the fp32 oracle; a tensor
whose fp32 oracle gradient is itself further than that from the float64 oracle's must be as close to float64 as the
fp32 oracle is), labels bit-exact.  The model is called exactly as bench.py and Trainer call it (`extras=` only exports tensors; it selects no
arithmetic), so the step under test is the fused-tail step that ships.  The measured errors are printed (run with -s to see them)."""
import numpy as np
import pytest
import torch

from tests.helpers import build_pair, omniglot_batch, feature_batch, rel_err, em_elementwise_check, RTOL

pytestmark = pytest.mark.gpu
GTOL = 5e-5          # per-tensor gradient bar, norm-wise against the fp32 oracle (measured worst: 6e-6)


def _scalar_close(name, a, b, tol=RTOL, floor=1e-7):
    a, b = float(a.detach()), float(b.detach())
    print('  %-10s gpu %.8g oracle %.8g rel %.2e' % (name, a, b, abs(a - b) / max(abs(b), 1e-30)))
    assert abs(a - b) <= tol * abs(b) + floor, '%s: gpu %r oracle %r' % (name, a, b)


def _train_step_pair(variant, linear, B, S, M, K, iters, device, al, feat=256, seed=0):
    from oracle.gmvae_oracle import make_trainable
    model, oracle = build_pair(variant, K=K, M=M, iters=iters, linear=linear, seed=seed, feat=feat)
    X = omniglot_batch(B, S, seed=seed + 1) if variant == 'omniglot' else feature_batch(B, S, feat, seed=seed + 1)
    eps = torch.randn(B, S, M, 64, generator=torch.Generator().manual_seed(seed + 2))
    rng = np.random.RandomState(seed + 3)
    idx = np.stack([rng.choice(S * M, K, replace=False) for _ in range(B)], 0)
    oracle.P = make_trainable(oracle.P)
    ex_o, ex_g = {}, {}
    out_o = oracle.forward(X, 0.1, 0.1, al, eps=eps, init_idx=idx, extras=ex_o)
    ex_o['init_idx'] = idx
    (out_o[1] + out_o[2] + out_o[3]).backward()
    model.to(device).train()
    out_g = model(X.to(device), 0.1, 0.1, al, eps=eps.to(device), init_idx=torch.from_numpy(idx).to(device),
                  extras=ex_g)
    (out_g[1] + out_g[2] + out_g[3]).backward()
    torch.cuda.synchronize()

    def grads_fp64():
        """The same step of the oracle in float64: how far fp32 rounding alone moves a gradient of this problem.
        Only run when a tensor misses the bar against the fp32 oracle."""
        from oracle.gmvae_oracle import OracleGMVAE
        P64 = make_trainable({k: (v.detach().double() if v.is_floating_point() else v) for k, v in oracle.P.items()})
        o64 = OracleGMVAE(P64, oracle.variant, input_shape=oracle.input_shape, unsupervised_em_iters=oracle.unsup_iters,
                          semisupervised_em_iters=oracle.semi_iters, fix_pi=oracle.fix_pi, component_size=oracle.K,
                          train_mc_sample_size=oracle.train_mc, test_mc_sample_size=oracle.test_mc,
                          linear=oracle.linear)
        out64 = o64.forward(X.double(), 0.1, 0.1, al, eps=eps.double(), init_idx=idx)
        (out64[1] + out64[2] + out64[3]).backward()
        return {k: v.grad for k, v in P64.items() if getattr(v, 'grad', None) is not None}

    ex_o['grads_fp64'] = grads_fp64
    return model, oracle, out_o, ex_o, out_g, ex_g


def _check_step(model, oracle, out_o, ex_o, out_g, ex_g):
    for i, n in enumerate(('rec_loss', 'kl_loss', 'h_augment'), start=1):
        _scalar_close(n, out_g[i], out_o[i])
    # bare h: the oracle's own `trace - 64` cancels at ulp(64) = 7.6e-6
    hg, ho = float(out_g[4].detach()), float(out_o[4].detach())
    assert abs(hg - ho) <= RTOL * abs(ho) + 64 * 6e-8 * 8
    e = rel_err(out_g[0], out_o[0])
    print('  reconstruction rel %.2e' % e)
    assert e < RTOL
    for key in ('q_z_mean', 'q_z_logvar', 'mu_k', 'pi_k', 'log_p_z', 'log_q_z', 'log_p_eps'):
        e = rel_err(ex_g[key].reshape(-1), ex_o[key].reshape(-1))
        print('  %-10s norm-rel %.2e' % (key, e))
        assert e < RTOL, key
    # element-wise on the EM parameters, anchored on the fp64 EM (tests/helpers.py)
    em_elementwise_check(oracle, ex_g, ex_o, ex_o['init_idx'])
    # every parameter gradient, norm-wise against the fp32 oracle; gradients that are analytically zero (conv bias
    # in front of BatchNorm, fc_k bias under the softmax) are measured against the scale of the whole gradient
    grads = {n: oracle.P[n].grad for n, _ in model.named_parameters() if oracle.P[n].grad is not None}
    gscale = max(float(g.abs().max()) for g in grads.values())
    worst = ('', 0.0)
    g64 = None
    for name, p in model.named_parameters():
        g_o = oracle.P[name].grad
        if g_o is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        g_h = p.grad.detach().cpu().double()
        err = float((g_h - g_o.double()).norm())
        nrm = float(g_o.double().norm())
        floor = 1e-6 * gscale * np.sqrt(g_o.numel())
        r = max(err - floor, 0.0) / max(nrm, 1e-30)
        if r > worst[1]:
            worst = (name, r)
        if err <= GTOL * nrm + floor:
            continue
        # a sum of many cancelling terms (a weight gradient that is small against its summands): the fp32 oracle
        # itself sits that far from the exact gradient.  Then the HIP gradient must be as close to the float64
        # result as the fp32 oracle is (x2), and still within 4 GTOL of the fp32 oracle.
        if g64 is None:
            g64 = ex_o['grads_fp64']()
        err64 = float((g_h - g64[name]).norm())
        ref64 = float((g_o.double() - g64[name]).norm())
        print('  %s: |hip - oracle32| %.2e  |hip - oracle64| %.2e  |oracle32 - oracle64| %.2e  norm %.2e'
              % (name, err, err64, ref64, nrm))
        assert err64 <= 2 * ref64 + floor and err <= 4 * GTOL * nrm + floor, \
            '%s: gradient error %.3e of norm %.3e (fp64: hip %.3e, oracle32 %.3e)' % (name, err, nrm, err64, ref64)
    print('  worst gradient: %s %.2e' % worst)


def test_omniglot_20way_training_step_against_oracle(device):
    """The metric's own configuration, one whole step, forward and backward."""
    res = _train_step_pair('omniglot', False, 4, 300, 32, 20, 10, device, al=1e-4)
    _check_step(*res)


def test_omniglot_5way_training_step_against_oracle(device):
    """BASELINE.json configs[1]: Omniglot 5-way, sample-size 200 (B=4, S=200, M=32, K=5)."""
    res = _train_step_pair('omniglot', False, 4, 200, 32, 5, 10, device, al=1e-4)
    _check_step(*res)


def test_mimgnet_training_step_real_shape_against_oracle(device):
    res = _train_step_pair('feat', True, 16, 5, 256, 5, 10, device, al=1e-3)
    _check_step(*res)


def test_celeba_training_step_real_shape_against_oracle(device):
    res = _train_step_pair('feat', False, 16, 5, 256, 5, 10, device, al=1e-4)
    _check_step(*res)


def _prediction_pair(variant, linear, B, K, shot, query, M, iters, device, feat=256, seed=0):
    model, oracle = build_pair(variant, K=K, M=M, iters=iters, linear=linear, seed=seed, feat=feat)
    tr, te = K * shot, K * query
    X = omniglot_batch(B, tr + te, seed=seed + 7) if variant == 'omniglot' else \
        feature_batch(B, tr + te, feat, seed=seed + 7)
    y_tr = torch.arange(K).repeat_interleave(shot)[None].repeat(B, 1)
    eps = torch.randn(B, tr + te, M, 64, generator=torch.Generator().manual_seed(seed + 8))
    oracle.training = False
    ex_o, ex_g = {}, {}
    y_o = oracle.prediction(X[:, :tr], y_tr, X[:, tr:], eps=eps, extras=ex_o)
    model.to(device).eval()
    y_g = model.prediction(X[:, :tr].to(device), y_tr.to(device), X[:, tr:].to(device), eps=eps.to(device),
                           extras=ex_g)
    assert y_g.dtype == torch.int64 and tuple(y_g.shape) == (B, te)
    e_mu = rel_err(ex_g['mu_k'], ex_o['mu_k'])
    e_lv = rel_err(ex_g['logvar_k'], ex_o['logvar_k'])
    e_var = rel_err(ex_g['logvar_k'].exp(), ex_o['logvar_k'].exp())
    e_pm = rel_err(ex_g['post_mean'], ex_o['post_mean'])
    n_diff = int((y_g.cpu() != y_o).sum())
    print('  mu_k %.2e  logvar_k %.2e  var_k %.2e  post_mean %.2e  labels differing %d of %d'
          % (e_mu, e_lv, e_var, e_pm, n_diff, y_o.numel()))
    assert e_mu < RTOL and e_var < RTOL and e_pm < RTOL
    assert n_diff == 0, 'few-shot assignments must be bit-exact'


@pytest.mark.parametrize('shot', [1, 5])
def test_omniglot_20way_prediction_against_oracle(device, shot):
    _prediction_pair('omniglot', False, 4, 20, shot, 15, 32, 10, device)


@pytest.mark.parametrize('variant_linear,shot', [(True, 1), (False, 5)])
def test_feature_prediction_sweep_shape_against_oracle(device, variant_linear, shot):
    """Mini-ImageNet (linear) 1-shot and CelebA (NOTEARS-MLP) 5-shot at N_u = 75 x 256 = 19200."""
    _prediction_pair('feat', variant_linear, 4, 5, shot, 15, 256, 10, device)
