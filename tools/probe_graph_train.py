"""Does the whole O20 training step (forward, backward, gradient pack, Adam) capture into one HIP graph, and what does a
replay cost next to the eager step?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from cmvae_amd import dist as cdist
from cmvae_amd.linear import FlatAdam
dev = torch.device('cuda:0')
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'o20']
model = bench.build_model(dev, cfg)
flat = cdist.FlatGrads(model.parameters())
opt = FlatAdam(flat, lr=cfg['lr'], weight_decay=5e-5)
np.random.seed(1); torch.manual_seed(1)
way, S, M, B = cfg['way'], cfg['sample'], cfg['mc'], cfg['batch']
N = S * M
pool = [bench.synth_batch(cfg, B, S, dev, 100 + i) for i in range(4)]
X_static = pool[0].clone()
idx_static = torch.zeros(B, way, dtype=torch.int64, device=dev)
seeds = cdist.SeedUploader((B, way), dev)
hl, hc, al = cfg['consts']

def body(X, idx):
    model.train(); flat.zero()
    _, rec, kl, h, _ = model(X, hl, hc, al, init_idx=idx)
    (rec + kl + h).backward()
    flat.allreduce(); opt.step()
    return rec, kl, h

def eager(i):
    return body(pool[i % 4], seeds.upload(cdist.draw_em_init(N, way, B, 0, 1)))

for i in range(5):
    eager(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    eager(i)
torch.cuda.synchronize()
print('eager  %.3f ms/step' % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3):
        body(X_static, idx_static)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body(X_static, idx_static)
torch.cuda.synchronize()
print('captured', flush=True)

def replay(i):
    X_static.copy_(pool[i % 4], non_blocking=True)
    idx_static.copy_(seeds.upload(cdist.draw_em_init(N, way, B, 0, 1)), non_blocking=True)
    g.replay()

for i in range(5):
    replay(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    replay(i)
torch.cuda.synchronize()
print('graph  %.3f ms/step   losses %s' % ((time.perf_counter() - t0) / 20 * 1e3, [float(x) for x in out]), flush=True)
# same state, same inputs: one eager step vs one replay from identical weights
