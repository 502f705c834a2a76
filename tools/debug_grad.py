import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.helpers import build_pair, omniglot_batch, rel_err
from oracle.gmvae_oracle import make_trainable
lin = sys.argv[1] == '1'
dev = torch.device('cuda:0')
B, S, M, K = 2, 6, 4, 3
model, oracle = build_pair('omniglot', K=K, M=M, iters=3, linear=lin)
X = omniglot_batch(B, S, seed=1)
g = torch.Generator().manual_seed(2)
eps = torch.randn(B, S, M, 64, generator=g)
rng = np.random.RandomState(3)
idx = np.stack([rng.choice(S * M, K, replace=False) for _ in range(B)], 0)
oracle.P = make_trainable(oracle.P)
ex_o = {}
out_o = oracle.forward(X, .1, .1, 1e-4, eps=eps, init_idx=idx, extras=ex_o)
for k in ('q_z', 'q_eps'): ex_o[k].retain_grad()
terms = {'rec': out_o[1], 'kl': out_o[2], 'h': out_o[3]}
which = sys.argv[2] if len(sys.argv) > 2 else 'all'
loss_o = sum(terms.values()) if which == 'all' else terms[which]
loss_o.backward()
model.to(dev).train()
ex_g = {}
out_g = model(X.to(dev), .1, .1, 1e-4, eps=eps.to(dev), init_idx=torch.from_numpy(idx).to(dev), extras=ex_g)
for k in ('q_z', 'q_eps'): ex_g[k].retain_grad()
tg = {'rec': out_g[1], 'kl': out_g[2], 'h': out_g[3]}
loss_g = sum(tg.values()) if which == 'all' else tg[which]
loss_g.backward()
for k in ('q_z', 'q_eps'):
    print(k, 'val', rel_err(ex_g[k], ex_o[k]), 'grad', rel_err(ex_g[k].grad, ex_o[k].grad))
for name, p in model.named_parameters():
    go = oracle.P[name].grad
    if go is None or p.grad is None: 
        print(name, 'none', go is None, p.grad is None); continue
    print('%-40s %.3e  |g| %.3e' % (name, rel_err(p.grad, go), float(go.norm())))
