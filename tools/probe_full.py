import sys, time, torch, numpy as np
sys.path.insert(0, '.')
t0 = time.time()
def note(m):
    torch.cuda.synchronize(); print('[%6.1fs] %s' % (time.time()-t0, m), flush=True)
from cmvae_amd import ops
from cmvae_amd.model import OmniglotGMVAE
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = OmniglotGMVAE(component_size=20, train_mc_sample_size=32, test_mc_sample_size=32, unsupervised_em_iters=10, semisupervised_em_iters=10).to(dev)
note('model built')
z = torch.randn(4, 9600, 64, device=dev)
idx = torch.stack([torch.randperm(9600)[:20] for _ in range(4)]).to(dev)
c = m._causal_desc()
out = ops.unsup_prior(z, idx, c, 10); note('unsup_prior done')
lw = out[3]
l = ops.mixture_loglik(z.requires_grad_(True), out[1], lw); l.sum().backward(); note('mixture done')
X = (torch.rand(4, 300, 1, 28, 28, device=dev) < 0.15).float()
with torch.no_grad():
    H = m.encoder(X.view(-1, 1, 28, 28)); note('encoder fwd done')
    q = m.q_z_given_H(H.view(4, 300, 256)); note('SAB done')
    hr = torch.randn(38400, 128, device=dev)
    pr = m.proj(hr); note('proj done')
    for i, layer in enumerate(m.decoder.decoder):
        pr = layer(pr.view(-1, 64, 2, 2) if i == 0 else pr); note('decoder layer %d done %s' % (i, tuple(pr.shape)))
