import sys, time, torch
sys.path.insert(0, '.')
from cmvae_amd.model import OmniglotGMVAE
from cmvae_amd.decoder import decoder_bce
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = OmniglotGMVAE(component_size=20, train_mc_sample_size=32, test_mc_sample_size=32).to(dev).train()
n = 38400
h = torch.randn(n, 256, device=dev).requires_grad_(True)
X = (torch.rand(1200, 784, device=dev) < 0.15).float()
for it in range(2):
    p, loss = decoder_bce(m.decoder, torch.nn.functional.elu(h), X, 32)
    (loss / n).backward()
torch.cuda.synchronize()
print('done')
