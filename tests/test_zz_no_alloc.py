"""cmvae_unsup_prior_noalloc (the entry point that allocates nothing: its workspace is a static device array)
against cmvae_unsup_prior through the C-ABI: same kernels on the same partition, so every output must agree bit for
bit — on two calls in a row, the second with a smaller input (the array keeps the first call's contents), and
shapes beyond the array are rejected before anything is launched."""
import ctypes

import numpy as np
import pytest
import torch

from tests.helpers import build_pair, clustered_latents

pytestmark = pytest.mark.gpu


def _both(lib, z, idx, desc, iters, fix_pi, gamma1):
    B, N, _ = z.shape
    K = idx.shape[1]
    dev = z.device
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for noalloc in (False, True):
        pi, tau, logw = (torch.full((B, K), float('nan'), device=dev) for _ in range(3))
        mu = torch.full((B, K, 64), float('nan'), device=dev)
        if noalloc:
            rc = lib.cmvae_unsup_prior_noalloc(z.data_ptr(), B, N, K, idx.data_ptr(), desc.ref, iters, fix_pi, gamma1,
                                               pi.data_ptr(), mu.data_ptr(), tau.data_ptr(), logw.data_ptr(), st)
        else:
            nbytes = lib.cmvae_unsup_prior_workspace_bytes(B, N, K)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = lib.cmvae_unsup_prior(z.data_ptr(), B, N, K, idx.data_ptr(), desc.ref, iters, fix_pi, gamma1,
                                       ws.data_ptr(), nbytes, pi.data_ptr(), mu.data_ptr(), tau.data_ptr(),
                                       logw.data_ptr(), st)
        assert rc == 0, lib.cmvae_last_error()
        torch.cuda.synchronize()
        outs.append((pi, mu, tau, logw))
    return outs


@pytest.mark.parametrize('linear', [False, True])
def test_noalloc_matches_workspace_entry_point(device, linear):
    from cmvae_amd._lib import lib
    from tests.test_gpu_kernels import _causal_desc
    model, _ = build_pair('omniglot', K=20, linear=linear)
    desc = _causal_desc(model, device)
    assert lib.cmvae_unsup_prior_noalloc_max_bytes() == 8 << 20
    # first call: the metric's shape (B=4, N=9600, K=20); second call: smaller and ragged, K=5
    for (B, N, K, iters, seed) in ((4, 9600, 20, 10, 1), (3, 1000, 5, 4, 2)):
        _, oracle = build_pair('omniglot', K=K, linear=linear)          # same seeded weights, K components
        z, _, _ = clustered_latents(B, N, K, seed=seed)
        rng = np.random.RandomState(seed)
        idx = torch.from_numpy(np.stack([rng.choice(N, K, replace=False) for _ in range(B)], 0)).to(device)
        zd = z.to(device)
        ref, got = _both(lib, zd, idx, desc, iters, 0, 1.0)
        for a, b in zip(ref, got):
            assert torch.isfinite(a).all()
            assert torch.equal(a, b)
        # and against the oracle, to the 1e-4 bar of every EM test
        if (B, N) == (3, 1000):
            rpi, rmu, rtau = oracle.unsup_prior(z, idx.cpu().numpy(), iters=iters)
            assert float((got[1].cpu() - rmu).norm() / rmu.norm()) < 1e-4
            assert float((got[0].cpu() - rpi).norm() / rpi.norm()) < 1e-4


def test_noalloc_rejects_shapes_beyond_the_static_buffer(device):
    from cmvae_amd._lib import lib
    from tests.test_gpu_kernels import _causal_desc
    model, _ = build_pair('omniglot', K=32, linear=True)
    desc = _causal_desc(model, device)
    B, N, K = 2048, 32, 32
    assert lib.cmvae_unsup_prior_workspace_bytes(B, N, K) > lib.cmvae_unsup_prior_noalloc_max_bytes()
    z = torch.zeros(B, N, 64, device=device)
    idx = torch.arange(K, device=device)[None].repeat(B, 1).contiguous()
    pi, tau, logw = (torch.full((B, K), 7.0, device=device) for _ in range(3))
    mu = torch.full((B, K, 64), 7.0, device=device)
    rc = lib.cmvae_unsup_prior_noalloc(z.data_ptr(), B, N, K, idx.data_ptr(), desc.ref, 2, 0, 1.0, pi.data_ptr(),
                                       mu.data_ptr(), tau.data_ptr(), logw.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b'scratch' in lib.cmvae_last_error()
    torch.cuda.synchronize()
    assert float(mu.min()) == 7.0 and float(pi.max()) == 7.0        # nothing was launched
