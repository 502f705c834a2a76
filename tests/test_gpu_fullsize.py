"""Parity at BASELINE.json's full sizes (Omniglot 20-way: B=4, S=300, M=32, N=9600, K=20; 38400 decoder images).

The CPU oracle cannot run a whole O20 step in seconds, so these tests use what the domain offers:
  * per-image independence of the convolutions: a slice of images of the full-size launch against the oracle;
  * linearity of the weight gradient in the rows it contracts over;
  * the EM loop itself IS affordable on the CPU at full size (3 iterations) and is compared directly;
  * fixed-order reductions: two identical full-size steps must agree bit for bit;
  * mixture weights sum to one, labels stay in range.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import build_pair, clustered_latents, rel_err, RTOL

pytestmark = pytest.mark.gpu
B, S, M, K = 4, 300, 32, 20
N = S * M


def _nhwc(x):          # [n,C,H,W] -> [n*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def test_full_size_decoder_layer_image_slices(device):
    """Layer 3 (ConvTranspose2d 64->64, k4 s2 p1 on 7x7) over all 38400 images, BatchNorm+ReLU of the previous layer
    folded into the operand load; first and last images against torch on the CPU (omniglot/module.py:57-59)."""
    from cmvae_amd import decoder
    n = B * S * M
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 64, 4, 4, generator=g) * 0.05
    b = torch.randn(64, generator=g) * 0.1
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    head = torch.randn(3, 64, 7, 7, generator=g)
    tail = torch.randn(2, 64, 7, 7, generator=g)
    x = torch.empty(n * 49, 64, device=device).normal_(generator=torch.Generator(device=device).manual_seed(1))
    x[:3 * 49] = _nhwc(head).to(device)
    x[-2 * 49:] = _nhwc(tail).to(device)
    y, part = decoder.convT_forward(x, n, 7, 4, w.to(device), b.to(device), (sc.to(device), sh.to(device)), want_stats=True)
    sums = decoder.colsum(part)
    for imgs, sl in ((head, slice(0, 3 * 196)), (tail, slice(n * 196 - 2 * 196, n * 196))):
        a = F.relu(imgs * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        ref = F.conv_transpose2d(a, w, b, stride=2, padding=1)
        assert rel_err(y[sl], _nhwc(ref)) < 1e-5
    # the BatchNorm statistics of the epilogue equal the column sums of what was written
    y64 = y.double()
    assert rel_err(sums[:64], y64.sum(0)) < 1e-5
    assert rel_err(sums[64:], (y64 * y64).sum(0)) < 1e-5


def test_full_size_weight_gradient_is_additive_over_images(device):
    """dW over 38400 images = dW over the first 30000 + dW over the rest (contraction over rows is linear)."""
    from cmvae_amd import conv
    n, n1 = B * S * M, 30000
    gen = torch.Generator(device=device).manual_seed(2)
    x = torch.empty(n * 16, 64, device=device).normal_(generator=gen)
    gy = torch.empty(n * 49, 64, device=device).normal_(generator=gen)

    def wgrad(xs, gs, cnt):
        dW = torch.zeros(64, 64, 3, 3, device=device)
        db = torch.zeros(64, device=device)
        first = True
        for py in (0, 1):
            for px in (0, 1):
                geom, tap_idx = conv.geom_convT_phase(cnt, 4, 4, 3, py, px)
                conv.conv_wgrad(geom, xs, gs, dW, tap_idx, s_ci=64 * 9, s_co=9, db=db, accumulate_db=not first)
                first = False
        return dW, db

    dW, db = wgrad(x, gy, n)
    dWa, dba = wgrad(x[:n1 * 16], gy[:n1 * 49], n1)
    dWb, dbb = wgrad(x[n1 * 16:].contiguous(), gy[n1 * 49:].contiguous(), n - n1)
    assert rel_err(dW, dWa + dWb) < 2e-5
    assert rel_err(db, dba + dbb) < 2e-5


def test_full_size_linear_row_slices(device):
    """proj layer 2 (256 -> 256, ELU) over all 38400 rows; a slice of rows against torch on the CPU."""
    from cmvae_amd.linear import linear
    n = B * S * M
    g = torch.Generator().manual_seed(3)
    W = torch.randn(256, 256, generator=g) / 16
    b = torch.randn(256, generator=g) * 0.1
    probe = torch.randn(300, 256, generator=g)
    x = torch.empty(n, 256, device=device).normal_(generator=torch.Generator(device=device).manual_seed(4))
    x[:150] = probe[:150].to(device)
    x[-150:] = probe[150:].to(device)
    y = linear(x, W.to(device), b.to(device), 'elu')
    ref = F.elu(F.linear(probe, W, b))
    assert rel_err(y[:150], ref[:150]) < 1e-5 and rel_err(y[-150:], ref[150:]) < 1e-5


def test_full_size_em_against_oracle(device):
    """Unsupervised EM at B=4, N=9600, K=20 (3 iterations): parameters against the oracle's literal loop, mixture
    weights sum to one, and a second run is bit-identical (per-block partials reduced in a fixed order)."""
    from cmvae_amd import ops
    from tests.test_gpu_kernels import _causal_desc
    model, oracle = build_pair('omniglot', K=K, linear=False)
    c = _causal_desc(model, device)
    z, _, _ = clustered_latents(B, N, K, seed=11)
    rng = np.random.RandomState(5)
    idx = np.stack([rng.choice(N, K, replace=False) for _ in range(B)], 0)
    rpi, rmu, rtau = oracle.unsup_prior(z, idx, iters=3)
    zd, idxd = z.to(device), torch.from_numpy(idx).to(device)
    pi, mu, tau, logw = ops.unsup_prior(zd, idxd, c, 3, False, model.gamma1)
    assert rel_err(mu, rmu) < RTOL and rel_err(pi, rpi) < RTOL and rel_err(tau, rtau) < 1e-6
    assert float((pi.sum(-1) - 1).abs().max()) < 1e-5
    pi2, mu2, tau2, logw2 = ops.unsup_prior(zd, idxd, c, 3, False, model.gamma1)
    assert torch.equal(mu, mu2) and torch.equal(pi, pi2) and torch.equal(logw, logw2)


def test_full_size_training_step_is_reproducible(device):
    """Two O20 training steps from the same weights, images, noise and EM seeds: losses and every gradient agree
    bit for bit (no float atomics anywhere on the path); losses are finite; prediction labels are in range."""
    import bench
    from cmvae_amd import dist as cdist
    torch.manual_seed(10086)
    model = bench.build_model(device)
    flat = cdist.FlatGrads(model.parameters())
    X = bench.synth_images(B, S, 'cpu', 100).to(device)
    eps = torch.randn(B, S, M, 64, device=device)
    rng = np.random.RandomState(3)
    idx = torch.from_numpy(np.stack([rng.choice(N, K, replace=False) for _ in range(B)], 0)).to(device)
    outs = []
    for _ in range(2):
        model.train()
        flat.zero()
        _, rec, kl, h_aug, h = model(X, 0.1, 0.1, 1e-4, eps=eps, init_idx=idx)
        (rec + kl + h_aug).backward()
        flat.pack()
        outs.append((rec.detach().clone(), kl.detach().clone(), h_aug.detach().clone(), flat.flat.clone()))
    # (training mode normalises with batch statistics: the running-statistics update between the steps is inert)
    for a, b in zip(outs[0], outs[1]):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)
    model.eval()
    Xtr, Xte = bench.synth_images(B, K, device, 1), bench.synth_images(B, K * 15, device, 2)
    ytr = torch.arange(K, device=device)[None].repeat(B, 1)
    y = model.prediction(Xtr, ytr, Xte)
    assert y.shape == (B, K * 15) and y.dtype == torch.int64 and int(y.min()) >= 0 and int(y.max()) < K


@pytest.mark.parametrize('K_,N_', [(1, 64), (32, 640), (2, 33)])
def test_component_count_limits(device, K_, N_):
    """K = 1 and K = 32 (the kernel limit) and a ragged N against the oracle."""
    from cmvae_amd import ops
    from tests.test_gpu_kernels import _causal_desc
    model, oracle = build_pair('omniglot', K=K_, linear=True)
    c = _causal_desc(model, device)
    z, _, _ = clustered_latents(2, N_, K_, seed=K_)
    rng = np.random.RandomState(K_)
    idx = np.stack([rng.choice(N_, K_, replace=False) for _ in range(2)], 0)
    rpi, rmu, rtau = oracle.unsup_prior(z, idx, iters=2)
    pi, mu, tau, _ = ops.unsup_prior(z.to(device), torch.from_numpy(idx).to(device), c, 2, False, model.gamma1)
    assert rel_err(mu, rmu) < RTOL and rel_err(pi, rpi) < RTOL
    with pytest.raises(RuntimeError):
        ops.unsup_prior(z.to(device), torch.zeros(2, 33, dtype=torch.int64, device=device), c, 1, False, 1.0)


def test_em_sweep_batch_is_independent_of_the_tile_partition(device):
    """Evaluation-sweep shape (hundreds of episodes per launch: 3 workgroups per episode, many tiles each) against the
    same episodes run 4 at a time (one tile per workgroup): the semi-supervised EM must not depend on how an
    episode's tiles are dealt to workgroups (only the order of the partial sums changes)."""
    from cmvae_amd import ops
    from tests.test_gpu_kernels import _causal_desc
    Bs, Nu, Ns, Ks = 260, 2500, 320, 5
    model, _ = build_pair('feat', K=Ks, linear=True)
    c = _causal_desc(model, device)
    g = torch.Generator().manual_seed(17)
    centers = torch.randn(Bs, Ks, 64, generator=g) * 2.0
    lab_u = torch.randint(0, Ks, (Bs, Nu), generator=g)
    uz = (centers.gather(1, lab_u[..., None].expand(-1, -1, 64)) + torch.randn(Bs, Nu, 64, generator=g)).to(device)
    y = torch.arange(Ns).remainder(Ks)[None].repeat(Bs, 1)
    sz = (centers.gather(1, y[..., None].expand(-1, -1, 64)) + torch.randn(Bs, Ns, 64, generator=g)).to(device)
    y = y.to(device)
    big = ops.semi_prior(uz, sz, y, Ks, c, 4, model.gamma1)
    for b0 in (0, 128, 256):
        small = ops.semi_prior(uz[b0:b0 + 4].contiguous(), sz[b0:b0 + 4].contiguous(), y[b0:b0 + 4].contiguous(), Ks, c,
                               4, model.gamma1)
        for t_big, t_small in zip(big, small):
            assert rel_err(t_big[b0:b0 + 4], t_small) < 1e-5
    # the mixture log-likelihood pass shares the partition
    logw = big[-1]
    lp_big = ops.mixture_loglik(uz, big[1], logw)
    lp_small = ops.mixture_loglik(uz[:4].contiguous(), big[1][:4].contiguous(), logw[:4].contiguous())
    assert rel_err(lp_big[:4], lp_small) < 1e-6
