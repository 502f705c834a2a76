"""Layer-3 data gradient (16 taps, 38400 images): image-major vs pixel-major row order, interleaved on one box."""
import sys
import torch
sys.path.insert(0, '.')
from cmvae_amd import conv
dev = torch.device('cuda:0')
n, Hi, k = 38400, 7, 4
g = torch.Generator(device=dev).manual_seed(0)
dy = torch.empty(n * 196, 64, device=dev).normal_(generator=g)
w = torch.empty(4, 4, 64, 64, device=dev).normal_(generator=g) * 0.05
out = torch.empty(n * 49, 64, device=dev)
res = {}
for rnd in range(4):
    for pm in (False, True):
        gd, _ = conv.geom_convT_dgrad(n, Hi, Hi, k, pixel_major=pm)
        conv.conv_gemm(gd, dy, w, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            conv.conv_gemm(gd, dy, w, out)
        e1.record(); e1.synchronize()
        res.setdefault(pm, []).append(e0.elapsed_time(e1) / 5)
for pm, v in res.items():
    print('pixel_major=%s: %s ms' % (pm, ['%.3f' % x for x in v]))
