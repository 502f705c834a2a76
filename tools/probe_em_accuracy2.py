"""The CelebA real-shape training step: the model's EM parameters against the fp64 EM on the model's own latents, per
iteration count, next to the fp32 oracle EM on the same latents."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from tests.helpers import build_pair, feature_batch, rel_err
from oracle.gmvae_oracle import OracleGMVAE
from cmvae_amd import ops
dev = torch.device('cuda:0')
B, S, M, K = 16, 5, 256, 5
model, o32 = build_pair('feat', K=K, M=M, iters=10, linear=False, seed=0, feat=256)
X = feature_batch(B, S, 256, seed=1)
eps = torch.randn(B, S, M, 64, generator=torch.Generator().manual_seed(2))
rng = np.random.RandomState(3)
idx = np.stack([rng.choice(S * M, K, replace=False) for _ in range(B)], 0)
P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in o32.P.items()}
o64 = OracleGMVAE(P64, 'feat', input_shape=256, component_size=K, train_mc_sample_size=M, linear=False)
model.to(dev).train()
ex = {}
with torch.no_grad():
    model(X.to(dev), 0.1, 0.1, 1e-4, eps=eps.to(dev), init_idx=torch.from_numpy(idx).to(dev), extras=ex)
qz = ex['q_z'].detach()
c = model._causal_desc()
pg, mg, tg, lg = ops.unsup_prior(qz.contiguous(), torch.from_numpy(idx).to(dev), c, 10, False, 1.0)
print('direct op == model EM:', bool(torch.equal(mg, ex['mu_k'])))
qc = qz.cpu()
for iters in (0, 1, 2, 3, 5, 10):
    pg, mg, tg, lg = ops.unsup_prior(qz.contiguous(), torch.from_numpy(idx).to(dev), c, iters, False, 1.0)
    p32, m32, _ = o32.unsup_prior(qc, idx, iters=iters)
    p64, m64, _ = o64.unsup_prior(qc.double(), idx, iters=iters)
    d = (mg.cpu().double() - m64).abs()
    b, k, j = np.unravel_index(int(d.argmax()), d.shape)
    print('iters %2d  mu: hip max %.2e rel %.2e | oracle32 max %.2e rel %.2e   pi: hip %.2e | oracle32 %.2e   worst at (b=%d,k=%d,d=%d) pi64=%.4f N_k=%.1f'
          % (iters, float(d.max()), rel_err(mg, m64), float((m32.double() - m64).abs().max()), rel_err(m32, m64),
             float((pg.cpu().double() - p64).abs().max()), float((p32.double() - p64).abs().max()), b, k, j,
             float(p64[b, k]), float(p64[b, k]) * S * M))
