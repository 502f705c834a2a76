"""Layer-3 weight gradient (4 phases x 4 taps, 38400 images, BatchNorm fold): time per phase launch, in isolation.
CMVAE_WGRAD_STREAM=0 -> the LDS-staged kernel; CMVAE_HIP_LIB=<ablation build> -> operand loads that always hit L1."""
import os
import sys
import torch
sys.path.insert(0, '.')
from cmvae_amd import conv
dev = torch.device('cuda:0')
n, Hi, k = 38400, 7, 4
g = torch.Generator(device=dev).manual_seed(0)
x = torch.empty(n * 49, 64, device=dev).normal_(generator=g)
gy = torch.empty(n * 196, 64, device=dev).normal_(generator=g)
sc = torch.rand(64, device=dev) + 0.5
sh = torch.randn(64, device=dev) * 0.3
if os.environ.get('NO_FOLD'):
    sc = sh = None
dW = torch.zeros(64, 64, k, k, device=dev)
db = torch.zeros(64, device=dev)
for pm in (True, False):
    geoms = [conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=pm) for py in (0, 1) for px in (0, 1)]

    def run():
        for i, (geom, tap_idx) in enumerate(geoms):
            conv.conv_wgrad(geom, x, gy, dW, tap_idx, s_ci=64 * k * k, s_co=k * k, db=db, accumulate_db=i > 0,
                            a_scale=sc, a_shift=sh)
    run(); torch.cuda.synchronize()
    ts = []
    for rnd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print('pixel_major=%s: %s us per phase launch (incl. reduce)' % (pm, ['%.1f' % t for t in ts]))
