"""Feature-model (Mini-ImageNet / CelebA shapes) train step and batched eval throughput."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from cmvae_amd.model import FeatureGMVAE
from cmvae_amd import dist as cdist
from cmvae_amd.linear import FlatAdam
dev = torch.device('cuda:0')
for linear, name in ((True, 'mimgnet'), (False, 'celeba')):
    torch.manual_seed(0)
    m = FeatureGMVAE(input_shape=256, unsupervised_em_iters=10, semisupervised_em_iters=10, component_size=5,
                     train_mc_sample_size=256, test_mc_sample_size=256, linear=linear).to(dev)
    flat = cdist.FlatGrads(m.parameters()); opt = FlatAdam(flat, lr=1e-4, weight_decay=5e-5)
    B, S = 16, 5
    H = torch.relu(torch.randn(B, S, 256, device=dev))
    seeds = cdist.SeedUploader((B, 5), dev)
    def step():
        m.train(); flat.zero()
        idx = seeds.upload(cdist.draw_em_init(S * 256, 5, B, 0, 1))
        _, rec, kl, h, _ = m(H, 0.1, 0.1, 1e-3, init_idx=idx)
        (rec + kl + h).backward(); flat.allreduce(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print('%s train: %.2f ms/step  %.0f episodes/s' % (name, dt * 1e3, B / dt))
    m.eval()
    for Be in (16, 256):
        for shot in (1, 5):
            tr, te = 5 * shot, 75
            Htr = torch.relu(torch.randn(Be, tr, 256, device=dev)); Hte = torch.relu(torch.randn(Be, te, 256, device=dev))
            ytr = torch.arange(5, device=dev).repeat_interleave(shot)[None].repeat(Be, 1)
            for _ in range(2): m.prediction(Htr, ytr, Hte)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): m.prediction(Htr, ytr, Hte)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            print('%s eval B=%d %d-shot: %.2f ms/call  %.0f episodes/s' % (name, Be, shot, dt * 1e3, Be / dt))
