"""Host enqueue time vs device time of one training step (is the step launch-bound?)."""
import sys, time, cProfile, pstats, io
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from cmvae_amd import dist as cdist
import os
cfg = bench.CONFIGS[os.environ.get('CMVAE_PROBE_CONFIG', 'o20')]
dev = torch.device('cuda:0')
model = bench.build_model(dev, cfg)
flat = cdist.FlatGrads(model.parameters())
from cmvae_amd.linear import FlatAdam
opt = FlatAdam(flat, lr=cfg['lr'], weight_decay=5e-5)
np.random.seed(1); torch.manual_seed(1)
X = bench.synth_batch(cfg, cfg['batch'], cfg['sample'], 'cpu', 100).to(dev)
N = cfg['sample'] * cfg['mc']

def step():
    model.train(); flat.zero()
    idx = torch.from_numpy(cdist.draw_em_init(N, cfg['way'], cfg['batch'], 0, 1)).pin_memory().to(dev, non_blocking=True)
    _, rec, kl, h, _ = model(X, *cfg['consts'], init_idx=idx)
    (rec + kl + h).backward()
    flat.allreduce(); opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
hs, ts = [], []
for _ in range(10):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append(t1 - t0); ts.append(t2 - t0)
print('host enqueue ms %.2f   total ms %.2f' % (1e3 * np.median(hs), 1e3 * np.median(ts)))
if len(sys.argv) > 1:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        step()
    torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45); print(s.getvalue()[:9000])
