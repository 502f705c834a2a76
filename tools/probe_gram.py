"""Time of cmvae_gram_inverse for nmat per-component matrices (nonlinear causal layer)."""
import sys
import torch
sys.path.insert(0, '.')
import bench
from cmvae_amd import ops
dev = torch.device('cuda:0')
model = bench.build_model(dev).eval()
c = model._causal_desc()
for nmat in (1, 80, 640, 1280):
    lv = torch.randn(nmat, 64, device=dev) * 0.3
    for _ in range(3):
        ops.gram_inverse(c, lv, 1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gram_inverse(c, lv, 1.0)
    e1.record(); torch.cuda.synchronize()
    print('nmat %5d: %.1f us per launch' % (nmat, 1e3 * e0.elapsed_time(e1) / 20), flush=True)
