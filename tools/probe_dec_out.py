"""Last decoder layer + BCE, forward and backward, at the O20 shape: time of the decoder_bce call pair (events)."""
import sys, torch
sys.path.insert(0, '.')
from cmvae_amd.model import OmniglotGMVAE
from cmvae_amd.decoder import decoder_bce
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = OmniglotGMVAE(component_size=20, train_mc_sample_size=32, test_mc_sample_size=32).to(dev).train()
n = 38400
h = torch.randn(n, 256, device=dev).requires_grad_(True)
X = (torch.rand(1200, 784, device=dev) < 0.15).float()
hh = torch.nn.functional.elu(h).detach().requires_grad_(True)
def run():
    p, loss = decoder_bce(m.decoder, hh, X, 32)
    (loss / n).backward()
for _ in range(2):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 3)
print('decoder fwd+bwd: %s ms' % ['%.3f' % t for t in ts])
