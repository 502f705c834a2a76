"""Generate tests/golden/*.npz from the CPU oracle (oracle/gmvae_oracle.py).

The reference ships no fixtures and cannot be executed in this pipeline, so these
vectors are self-generated regression pins of the oracle ("parity unpinned" with respect to the
reference itself).  Inputs are regenerated from seeds by tests/helpers.py; only the expected outputs
and the noise / EM seeds are stored.
Run:  python tools/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import build_pair, omniglot_batch, feature_batch   # noqa: E402

CASES = {
    'omniglot_nonlinear': dict(variant='omniglot', linear=False, K=3, M=4, B=2, S=6),
    'omniglot_linear': dict(variant='omniglot', linear=True, K=3, M=4, B=2, S=6),
    'feat_linear': dict(variant='feat', linear=True, K=5, M=8, B=2, S=9),
    'feat_nonlinear': dict(variant='feat', linear=False, K=5, M=8, B=2, S=9),
}


def case_inputs(c, seed=11):
    g = torch.Generator().manual_seed(seed)
    B, S, M, K = c['B'], c['S'], c['M'], c['K']
    X = omniglot_batch(B, S, seed=seed) if c['variant'] == 'omniglot' else feature_batch(B, S, 256, seed=seed)
    eps = torch.randn(B, S, M, 64, generator=g)
    rng = np.random.RandomState(seed)
    idx = np.stack([rng.choice(S * M, K, replace=False) for _ in range(B)], 0)
    # few-shot task carved from the same items: first K items are the 1-shot support set
    y_tr = torch.arange(K)[None].repeat(B, 1)
    return X, eps, idx, y_tr


def run_case(c):
    model, oracle = build_pair(c['variant'], K=c['K'], M=c['M'], iters=3, linear=c['linear'], seed=5)
    X, eps, idx, y_tr = case_inputs(c)
    ex = {}
    out = oracle.forward(X, 0.1, 0.1, 1e-4, eps=eps, init_idx=idx, extras=ex)
    oracle.training = False
    K = c['K']
    exp = {}
    y = oracle.prediction(X[:, :K], y_tr, X[:, K:], eps=eps, extras=exp)
    return dict(rec=out[1].detach().numpy(), kl=out[2].detach().numpy(), h_aug=out[3].detach().numpy(),
                h=out[4].detach().numpy(), pi_k=ex['pi_k'].numpy(), mu_k=ex['mu_k'].numpy(),
                log_p_z=ex['log_p_z'].detach().numpy(), y_pred=y.numpy(), sup_mu=exp['mu_k'].numpy(),
                sup_logvar=exp['logvar_k'].numpy())


if __name__ == '__main__':
    os.makedirs(os.path.join(ROOT, 'tests', 'golden'), exist_ok=True)
    for name, c in CASES.items():
        np.savez(os.path.join(ROOT, 'tests', 'golden', name + '.npz'), **run_case(c))
        print('wrote', name)
