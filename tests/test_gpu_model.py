"""Model-level parity on the GPU: GMVAE.forward (ELBO terms + parameter gradients) and
GMVAE.prediction (bit-exact labels) against the CPU oracle on identical weights, inputs,
reparametrisation noise and EM seeds.  Tolerance: 1e-4 relative (north_star)."""
import numpy as np
import pytest
import torch

from tests.helpers import build_pair, omniglot_batch, feature_batch, rel_err, max_rel, em_elementwise_check, RTOL

pytestmark = pytest.mark.gpu


def _run_forward_pair(model, oracle, X, eps, idx, device, hl=0.1, hc=0.1, al=1e-4):
    # oracle (CPU) with autograd on a private trainable copy of the weights
    from oracle.gmvae_oracle import make_trainable, OracleGMVAE
    from cmvae_amd import encoder as _enc, decoder as _dec, linear as _lin
    # the same oracle in fp64: tells how far fp32 rounding alone moves the gradients of this problem
    P64 = make_trainable({k: (v.double() if v.is_floating_point() else v) for k, v in oracle.P.items()})
    o64 = OracleGMVAE(P64, oracle.variant, input_shape=oracle.input_shape, unsupervised_em_iters=oracle.unsup_iters,
                      semisupervised_em_iters=oracle.semi_iters, fix_pi=oracle.fix_pi, component_size=oracle.K,
                      train_mc_sample_size=oracle.train_mc, test_mc_sample_size=oracle.test_mc, linear=oracle.linear)
    out64 = o64.forward(X.double(), hl, hc, al, eps=eps.double(), init_idx=idx)
    (out64[1] + out64[2] + out64[3]).backward()
    oracle.P64 = P64
    oracle.P = make_trainable(oracle.P)
    ex_o = {}
    oracle.taps = {}
    out_o = oracle.forward(X, hl, hc, al, eps=eps, init_idx=idx, extras=ex_o)
    ex_o['init_idx'] = idx
    (out_o[1] + out_o[2] + out_o[3]).backward()
    model.to(device).train()
    ex_g = {}
    taps_g = {}
    _enc.TAPS = _dec.TAPS = _lin.TAPS = taps_g
    try:
        out_g = model(X.to(device), hl, hc, al, eps=eps.to(device), init_idx=torch.from_numpy(idx).to(device),
                      extras=ex_g)
    finally:
        _enc.TAPS = _dec.TAPS = _lin.TAPS = None
    (out_g[1] + out_g[2] + out_g[3]).backward()
    ex_g['taps'], ex_o['taps'] = taps_g, oracle.taps
    return out_o, ex_o, out_g, ex_g


def _nchw(y, fold, n):
    """Pre-BatchNorm channels-last rows [n*H*W,64] + fold -> BatchNorm output [n,64,H,W] on the CPU."""
    hw = y.shape[0] // n
    H = int(round(hw ** 0.5))
    pre = y * fold[0] + fold[1]
    return pre.view(n, H, H, 64).permute(0, 3, 1, 2).cpu()


def _mask_flips(taps_g, taps_o):
    """Number of ReLU sign decisions and max-pool winners on which the HIP run and the oracle run differ.
    A unit whose pre-activation lies within fp32 rounding of zero (or a pooling window whose two largest entries
    lie that close together) can legitimately go either way; everything upstream of it then sees a different
    gradient."""
    import torch.nn.functional as F
    flips = 0
    if 'enc' in taps_g:
        ys, folds = taps_g['enc']
        n = taps_o['enc_bn'][0].shape[0]
        for y, fold, pre_o in zip(ys, folds, taps_o['enc_bn']):
            pre_g = _nchw(y, fold, n)
            flips += int(((pre_g > 0) != (pre_o > 0)).sum())
            pg, ig = F.max_pool2d(F.relu(pre_g), 2, ceil_mode=True, return_indices=True)
            po, io = F.max_pool2d(F.relu(pre_o), 2, ceil_mode=True, return_indices=True)
            flips += int(((ig != io) & ((pg > 0) | (po > 0))).sum())
    if 'dec' in taps_g:
        acts, folds = taps_g['dec']
        n = taps_o['dec_bn'][0].shape[0]
        for y, fold, pre_o in zip(acts[1:], folds[1:], taps_o['dec_bn']):
            flips += int(((_nchw(y, fold, n) > 0) != (pre_o > 0)).sum())
    lg, lo = taps_g.get('linear_relu', []), taps_o.get('linear_relu', [])
    assert len(lg) == len(lo)
    for y_g, pre_o in zip(lg, lo):
        flips += int(((y_g.cpu() > 0) != (pre_o.reshape(y_g.shape) > 0)).sum())
    return flips


def _check_forward(model, oracle, out_o, ex_o, out_g, ex_g):
    names = ['rec_loss', 'kl_loss', 'h_augment']
    for i, n in enumerate(names, start=1):
        a, b = float(out_g[i].detach()), float(out_o[i].detach())
        assert abs(a - b) <= RTOL * abs(b) + 1e-7, '%s: gpu %r oracle %r' % (n, a, b)
    # bare h suffers the oracle's own `trace - 64` cancellation: compare at the fp32-ulp(64) floor
    hg, ho = float(out_g[4].detach()), float(out_o[4].detach())
    assert abs(hg - ho) <= RTOL * abs(ho) + 64 * 6e-8 * 8
    assert rel_err(out_g[0], out_o[0]) < RTOL                               # reconstructions
    for key in ('q_z_mean', 'q_z_logvar', 'mu_k', 'pi_k', 'log_p_z', 'log_q_z', 'log_p_eps'):
        assert rel_err(ex_g[key].reshape(-1), ex_o[key].reshape(-1)) < RTOL, key
    # element-wise on the EM parameters, anchored on the fp64 EM (tests/helpers.py)
    em_elementwise_check(oracle, ex_g, ex_o, ex_o['init_idx'], verbose=False)
    strict_ok, worst = _grad_report(model, oracle)
    return strict_ok, worst, _mask_flips(ex_g['taps'], ex_o['taps'])


def _grad_report(model, oracle):
    """Per-tensor gradient errors.  Some gradients are analytically zero (conv bias in front of BatchNorm,
    fc_k bias under softmax): errors are measured against the scale of the whole gradient as well.
    Returns (strict_ok, worst_relative_error)."""
    gscale = max(float(g.abs().max()) for g in (oracle.P[n].grad for n, _ in model.named_parameters())
                 if g is not None)
    strict_ok, worst = True, 0.0
    for name, p in model.named_parameters():
        g_o = oracle.P[name].grad
        if g_o is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        g64 = oracle.P64[name].grad
        g_h = p.grad.detach().cpu().double()
        floor = 1e-6 * gscale * np.sqrt(g_o.numel())
        err32 = float((g_h - g_o.double()).norm())
        err64 = float((g_h - g64).norm())
        ref64 = float((g_o.double() - g64).norm())
        nrm = float(g_o.double().norm())
        if err32 > 2e-4 * nrm + floor and err64 > 4 * ref64 + floor:
            strict_ok = False
        worst = max(worst, (err32 - floor) / max(nrm, 1e-30))
    return strict_ok, worst


def _grad_parity_over_seeds(run):
    """Forward parity must hold for every seed.  Gradients: every seed on which the HIP run and the oracle take the
    same ReLU / max-pool decisions must match per tensor within 2e-4 of the fp32 oracle (or be as close to the fp64
    oracle as the fp32 oracle is, x4).  On these tiny batches a pre-activation within fp32 rounding of zero can
    legitimately fall on either side in two correct fp32 implementations, which moves every upstream gradient by
    ~1/N (seen: 4e-3 for one flip among 48 images): a seed may miss the strict bar only if such a flip is actually
    observed in the exported masks, and then stays within 2e-2.  At least two seeds must be flip-free so the strict
    bar is exercised."""
    results = [run(seed) for seed in (1, 2, 3, 4)]
    print('  (strict_ok, worst, mask flips) per seed:', results)
    for strict_ok, worst, flips in results:
        assert worst < 2e-2, results
        assert strict_ok or flips > 0, 'gradient mismatch without any ReLU / max-pool decision differing: %r' % (results,)
    assert sum(1 for _, _, flips in results if flips == 0) >= 2, results


@pytest.mark.parametrize('linear', [False, True])
def test_omniglot_forward_backward(device, linear):
    B, S, M, K = 2, 6, 4, 3

    def run(seed):
        model, oracle = build_pair('omniglot', K=K, M=M, iters=3, linear=linear, seed=seed)
        X = omniglot_batch(B, S, seed=seed)
        g = torch.Generator().manual_seed(seed + 1)
        eps = torch.randn(B, S, M, 64, generator=g)
        rng = np.random.RandomState(seed + 2)
        idx = np.stack([rng.choice(S * M, K, replace=False) for _ in range(B)], 0)
        res = _run_forward_pair(model, oracle, X, eps, idx, device)
        return _check_forward(model, oracle, *res)

    _grad_parity_over_seeds(run)


@pytest.mark.parametrize('linear', [True, False])
def test_feature_forward_backward(device, linear):
    B, S, M, K = 3, 5, 16, 5

    def run(seed):
        model, oracle = build_pair('feat', K=K, M=M, iters=3, linear=linear, feat=256, seed=seed)
        H = feature_batch(B, S, 256, seed=seed + 3)
        g = torch.Generator().manual_seed(seed + 4)
        eps = torch.randn(B, S, M, 64, generator=g)
        rng = np.random.RandomState(seed + 5)
        idx = np.stack([rng.choice(S * M, K, replace=False) for _ in range(B)], 0)
        res = _run_forward_pair(model, oracle, H, eps, idx, device, al=1e-3)
        return _check_forward(model, oracle, *res)

    _grad_parity_over_seeds(run)


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
    assert rel_err(ex_g['logvar_k'], ex_o['logvar_k']) < RTOL
    assert rel_err(ex_g['post_mean'], ex_o['post_mean']) < RTOL
    assert max_rel(ex_g['mu_k'], ex_o['mu_k'], floor=float(ex_o['mu_k'].abs().max()) * 1e-2) < 5 * RTOL
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


@pytest.mark.parametrize('name', ['omniglot_nonlinear', 'omniglot_linear', 'feat_linear', 'feat_nonlinear'])
def test_hip_path_matches_golden_vectors(device, name):
    """The committed fixtures (tests/golden/*.npz, made by tools/make_golden.py from the oracle)."""
    import os
    from tools.make_golden import CASES, case_inputs
    c = CASES[name]
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', name + '.npz'))
    model, _ = build_pair(c['variant'], K=c['K'], M=c['M'], iters=3, linear=c['linear'], seed=5)
    X, eps, idx, y_tr = case_inputs(c)
    model.to(device).train()
    ex = {}
    out = model(X.to(device), 0.1, 0.1, 1e-4, eps=eps.to(device), init_idx=torch.from_numpy(idx).to(device),
                extras=ex)
    for i, k in ((1, 'rec'), (2, 'kl'), (3, 'h_aug')):
        assert abs(float(out[i].detach()) - float(gold[k])) <= RTOL * abs(float(gold[k])) + 1e-7, k
    assert rel_err(ex['mu_k'], torch.from_numpy(gold['mu_k'])) < RTOL
    assert rel_err(ex['pi_k'], torch.from_numpy(gold['pi_k'])) < RTOL
    assert rel_err(ex['log_p_z'].reshape(-1), torch.from_numpy(gold['log_p_z']).reshape(-1)) < RTOL
    # the oracle's fixture was made with BatchNorm running statistics at their initial values
    model2, _ = build_pair(c['variant'], K=c['K'], M=c['M'], iters=3, linear=c['linear'], seed=5)
    model2.to(device).eval()
    K = c['K']
    y = model2.prediction(X[:, :K].to(device), y_tr.to(device), X[:, K:].to(device), eps=eps.to(device))
    assert np.array_equal(y.cpu().numpy(), gold['y_pred'])


@pytest.mark.parametrize('variant,linear', [('omniglot', False), ('feat', True)])
def test_tr_prediction_bit_exact(device, variant, linear):
    """GMVAE.tr_prediction (omniglot/model.py:587-629): cluster labels of the training items."""
    B, S, K, M = 2, 24, 4, 6
    model, oracle = build_pair(variant, K=K, M=M, iters=3, linear=linear)
    X = omniglot_batch(B, S, seed=5) if variant == 'omniglot' else feature_batch(B, S, 256, seed=5)
    eps = torch.randn(B, S, M, 64, generator=torch.Generator().manual_seed(6))
    rng = np.random.RandomState(2)
    idx = np.stack([rng.choice(S * M, K, replace=False) for _ in range(B)], 0)
    oracle.training = False
    y_o = oracle.tr_prediction(X, eps=eps, init_idx=idx)
    model.to(device).eval()
    y_g = model.tr_prediction(X.to(device), eps=eps.to(device), init_idx=torch.from_numpy(idx).to(device))
    assert y_g.dtype == torch.int64 and tuple(y_g.shape) == (B, S)
    assert torch.equal(y_g.cpu(), y_o)


def test_generate_matches_oracle(device):
    """GMVAE.generate (omniglot/model.py:634-723): reconstructions, component images, intervened reconstructions."""
    B, K, shot, query, M = 2, 5, 2, 3, 4
    model, oracle = build_pair('omniglot', K=K, M=M, iters=3, linear=False)
    tr, te = K * shot, K * query
    X = omniglot_batch(B, tr + te, seed=9)
    y_tr = torch.arange(K).repeat_interleave(shot)[None].repeat(B, 1)
    eps = torch.randn(B, tr + te, M, 64, generator=torch.Generator().manual_seed(10))
    oracle.training = False
    outs_o = oracle.generate(X[:, :tr], y_tr, X[:, tr:], eps=eps)
    model.to(device).eval()
    outs_g = model.generate(X[:, :tr].to(device), y_tr.to(device), X[:, tr:].to(device), eps=eps.to(device))
    assert tuple(outs_g[1].shape) == (te, 1, 28, 28) and tuple(outs_g[2].shape) == (K, 1, 28, 28)
    for a, b in zip(outs_g, outs_o):
        assert rel_err(a, b) < RTOL
