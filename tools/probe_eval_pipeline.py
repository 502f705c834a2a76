"""Evaluation throughput at B episodes per call on ONE box: sequential eager loop, eager calls on n streams, graph
replays on n streams (cmvae_amd/pipeline.py)."""
import sys, time
import torch
sys.path.insert(0, '.')
import bench
from cmvae_amd.pipeline import PredictionPipeline
dev = torch.device('cuda:0')
model = bench.build_model(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for shot in (1, 5):
    tr, te = bench.WAY * shot, bench.WAY * bench.QUERY
    Xtr = bench.synth_images(B, tr, dev, 200 + shot); Xte = bench.synth_images(B, te, dev, 300 + shot)
    ytr = torch.arange(bench.WAY, device=dev).repeat_interleave(shot)[None].repeat(B, 1)
    for n, graphs in ((1, False), (2, False), (4, False), (1, True), (2, True), (4, True), (8, True)):
        pipe = PredictionPipeline(model, n, use_graphs=graphs)
        for _ in range(2 * n):
            pipe.submit(Xtr, ytr, Xte)
        pipe.results(); torch.cuda.synchronize()
        calls = 40
        t0 = time.perf_counter()
        for _ in range(calls):
            pipe.submit(Xtr, ytr, Xte)
        t1 = time.perf_counter()
        pipe.results(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('B %d shot %d  streams %d graphs %-5s: %.3f ms/call (host %.3f)  -> %.0f episodes/s'
              % (B, shot, n, graphs, (t2 - t0) / calls * 1e3, (t1 - t0) / calls * 1e3, B * calls / (t2 - t0)), flush=True)
