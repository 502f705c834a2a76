"""Where does the EM's element-wise error come from?  Gram inverse, prior term and one E/M iteration of the HIP path and
of the fp32 oracle, each against the fp64 oracle (nonlinear and linear causal layer)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from tests.helpers import build_pair, rel_err
from oracle.gmvae_oracle import OracleGMVAE
from cmvae_amd import ops
dev = torch.device('cuda:0')
for linear in (False, True):
    K, B, N = 20, 4, 9600
    model, o32 = build_pair('omniglot', K=K, M=32, iters=10, linear=linear, seed=0)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in o32.P.items()}
    o64 = OracleGMVAE(P64, 'omniglot', component_size=K, linear=linear)
    model.to(dev)
    c = model._causal_desc()
    I = torch.eye(64)
    C64 = o64._causal(I.double()); inv64 = torch.linalg.inv(I.double() + C64 @ C64.T)
    C32 = o32._causal(I); inv32 = torch.inverse(I + C32 @ C32.T)
    invg = ops.gram_inverse(c, None, 1.0, 1)[0].cpu()
    print('linear=%s  Minv: hip rel %.2e max %.2e | oracle32 rel %.2e max %.2e' % (
        linear, rel_err(invg, inv64), float((invg.double() - inv64).abs().max()), rel_err(inv32, inv64),
        float((inv32.double() - inv64).abs().max())))
    g = torch.Generator().manual_seed(1)
    z = torch.randn(B, N, 64, generator=g) + 0.3 * torch.randn(B, 1, 64, generator=g)
    rng = np.random.RandomState(2)
    idx = np.stack([rng.choice(N, K, replace=False) for _ in range(B)], 0)
    # prior term of arbitrary means
    mu = torch.randn(B * K, 64, generator=g) * 0.4
    pi = torch.rand(B * K, generator=g) + 0.1; tau = torch.ones(B * K)
    eps_g, logw_g = ops.causal_rows(mu.to(dev), c, pi.to(dev), tau.to(dev))
    lw64 = torch.log(pi.double()) + o64._glp(o64._causal(mu.double()), torch.zeros(B * K, 64, dtype=torch.float64))
    lw32 = torch.log(pi) + o32._glp(o32._causal(mu), torch.zeros(B * K, 64))
    print('   logw: hip max abs %.2e | oracle32 max abs %.2e   (|logw| ~ %.1f)' % (
        float((logw_g.cpu().double() - lw64).abs().max()), float((lw32.double() - lw64).abs().max()), float(lw64.abs().mean())))
    for iters in (0, 1, 2, 10):
        pg, mg, tg, lg = ops.unsup_prior(z.to(dev), torch.from_numpy(idx).to(dev), c, iters, False, 1.0)
        p32, m32, _ = o32.unsup_prior(z, idx, iters=iters)
        p64, m64, _ = o64.unsup_prior(z.double(), idx, iters=iters)
        print('   iters %2d  mu: hip max %.2e rel %.2e | oracle32 max %.2e rel %.2e    pi: hip %.2e | oracle32 %.2e' % (
            iters, float((mg.cpu().double() - m64).abs().max()), rel_err(mg, m64), float((m32.double() - m64).abs().max()),
            rel_err(m32, m64), float((pg.cpu().double() - p64).abs().max()), float((p32.double() - p64).abs().max())))
