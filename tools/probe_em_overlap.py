"""Training step with the prior's EM on the second stream (model.em_side_stream) and in place, alternately in ONE process
on one box: ms per step over 20 steps each, four rounds; losses of the two settings must agree bit for bit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from cmvae_amd import dist as cdist
from cmvae_amd.linear import FlatAdam
cfg = bench.CONFIGS[os.environ.get('CMVAE_PROBE_CONFIG', 'o20')]
dev = torch.device('cuda:0')
model = bench.build_model(dev, cfg)
flat = cdist.FlatGrads(model.parameters())
opt = FlatAdam(flat, lr=cfg['lr'], weight_decay=5e-5)
np.random.seed(1); torch.manual_seed(1)
X = bench.synth_batch(cfg, cfg['batch'], cfg['sample'], 'cpu', 100).to(dev)
N = cfg['sample'] * cfg['mc']

def step(update=True):
    model.train(); flat.zero()
    idx = torch.from_numpy(cdist.draw_em_init(N, cfg['way'], cfg['batch'], 0, 1)).pin_memory().to(dev, non_blocking=True)
    _, rec, kl, h, _ = model(X, *cfg['consts'], init_idx=idx)
    loss = rec + kl + h
    loss.backward()
    if update:
        flat.allreduce(); opt.step()
    return loss

# same step twice, weights untouched: identical losses and gradients
vals = []
for side in (True, False):
    model.em_side_stream = side
    np.random.seed(5); torch.manual_seed(5)
    l = step(update=False)
    vals.append((l.item(), flat.flat.clone() if hasattr(flat, 'flat') else None))
print('loss side / in place: %r %r  equal %s' % (vals[0][0], vals[1][0], vals[0][0] == vals[1][0]))
if vals[0][1] is not None:
    print('gradients equal: %s' % torch.equal(vals[0][1], vals[1][1]))
for _ in range(3):
    step()
for rnd in range(4):
    for side in (True, False):
        model.em_side_stream = side
        step(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            step()
        e1.record(); e1.synchronize()
        print('em_side_stream=%-5s  %.3f ms/step' % (side, e0.elapsed_time(e1) / 20))
