"""Shared builders for the parity tests: seeded models, synthetic episodes, tolerances."""
import numpy as np
import torch

from cmvae_amd.model import OmniglotGMVAE, FeatureGMVAE
from oracle.gmvae_oracle import OracleGMVAE

# north_star tolerance: ELBO terms, EM parameters, causal loss within 1e-4 relative (fp32)
RTOL = 1e-4


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a, b, floor=1e-6):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float(((a - b).abs() / b.abs().clamp_min(floor)).max())


def randomize_causal(model, scale=1.0, seed=1):
    """Give a linear adjacency non-trivial values (the reference initialises A = 0)."""
    g = torch.Generator().manual_seed(seed)
    if model.causal.linear:
        with torch.no_grad():
            model.causal.A.copy_(torch.randn(64, 64, generator=g) * 0.05 * scale)


def build_pair(variant='omniglot', K=3, M=4, iters=3, linear=None, seed=0, feat=256, gamma=1, fix_pi=False,
               dtype=torch.float32):
    """A product model (CPU-constructed) and an oracle sharing the same seeded weights."""
    torch.manual_seed(seed)
    kw = dict(component_size=K, train_mc_sample_size=M, test_mc_sample_size=M, unsupervised_em_iters=iters,
              semisupervised_em_iters=iters, gamma=gamma, fix_pi=fix_pi)
    if variant == 'omniglot':
        lin = False if linear is None else linear
        model = OmniglotGMVAE(linear=lin, **kw)
        okw = dict(input_shape=(1, 28, 28))
    else:
        lin = True if linear is None else linear
        model = FeatureGMVAE(input_shape=feat, linear=lin, **kw)
        okw = dict(input_shape=feat)
    randomize_causal(model)
    sd = {k: v.detach().clone().to(dtype if v.is_floating_point() else v.dtype) for k, v in model.state_dict().items()}
    oracle = OracleGMVAE(sd, variant, linear=lin, **okw, **kw)
    return model, oracle


def omniglot_batch(B, S, seed=0, p=0.15):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, S, 1, 28, 28, generator=g) < p).float()


def feature_batch(B, S, F=256, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.relu(torch.randn(B, S, F, generator=g))


def clustered_latents(B, N, K, D=64, spread=3.0, noise=0.5, seed=0):
    """Latents drawn around K well-separated centres (+ labels)."""
    g = torch.Generator().manual_seed(seed)
    centres = torch.randn(B, K, D, generator=g) * spread / np.sqrt(D) * 4
    lab = torch.randint(0, K, (B, N), generator=g)
    z = torch.gather(centres, 1, lab[:, :, None].expand(-1, -1, D)) + noise * torch.randn(B, N, D, generator=g)
    return z, lab, centres
