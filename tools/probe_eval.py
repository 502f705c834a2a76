"""Where the time of GMVAE.prediction goes (Omniglot 20-way, B episodes per call)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for shot in (1, 5):
    tr, te = bench.WAY * shot, bench.WAY * bench.QUERY
    Xtr = bench.synth_images(B, tr, dev, 200 + shot); Xte = bench.synth_images(B, te, dev, 300 + shot)
    ytr = torch.arange(bench.WAY, device=dev).repeat_interleave(shot)[None].repeat(B, 1)
    for _ in range(3):
        model.prediction(Xtr, ytr, Xte)
    torch.cuda.synchronize()
    hs, ts = [], []
    for _ in range(20):
        t0 = time.perf_counter(); model.prediction(Xtr, ytr, Xte); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        hs.append(t1 - t0); ts.append(t2 - t0)
    print('B %d shot %d: host %.3f ms  total %.3f ms  -> %.1f episodes/s' % (B, shot, 1e3 * np.median(hs), 1e3 * np.median(ts), B / np.median(ts)))
