import sys, torch
sys.path.insert(0, '.')
from cmvae_amd import ops
from cmvae_amd.model import OmniglotGMVAE
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = OmniglotGMVAE(component_size=20, train_mc_sample_size=32, test_mc_sample_size=32, unsupervised_em_iters=10, semisupervised_em_iters=10).to(dev)
c = m._causal_desc()
z = torch.randn(4, 9600, 64, device=dev)
idx = torch.stack([torch.randperm(9600)[:20] for _ in range(4)]).to(dev)
sz = torch.randn(4, 3200, 64, device=dev)
y = torch.arange(20, device=dev).repeat_interleave(160)[None].repeat(4, 1).contiguous()
for it in range(2):
    out = ops.unsup_prior(z, idx, c, 10)
    o2 = ops.semi_prior(z, sz, y, 20, c, 10)
    yp, _ = ops.posterior_argmax(z, o2[1], o2[2], o2[4], 300, 32)
torch.cuda.synchronize()
