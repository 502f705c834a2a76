"""Evaluation throughput (4 episodes per call, graph replays) over 4 / 8 / 12 / 16 streams; run it once more with
GPU_MAX_HW_QUEUES=8 in the environment to see the hardware queues' part (measured: 8 streams on the default 4 queues is
the best point, 6457 / 5700 episodes/s; 8 queues 5707 / 5188)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from cmvae_amd.pipeline import PredictionPipeline
dev = torch.device('cuda:0')
model = bench.build_model(dev).eval()
B = 4
for shot in (1, 5):
    tr, te = bench.WAY * shot, bench.WAY * bench.QUERY
    Xtr = bench.synth_images(B, tr, dev, 200 + shot); Xte = bench.synth_images(B, te, dev, 300 + shot)
    ytr = torch.arange(bench.WAY, device=dev).repeat_interleave(shot)[None].repeat(B, 1)
    for n in (4, 8, 12, 16):
        pipe = PredictionPipeline(model, n, use_graphs=True)
        for _ in range(2 * n):
            pipe.submit(Xtr, ytr, Xte)
        pipe.results(); torch.cuda.synchronize()
        calls = 96
        t0 = time.perf_counter()
        for _ in range(calls):
            pipe.submit(Xtr, ytr, Xte)
        t1 = time.perf_counter()
        pipe.results(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('shot %d streams %2d: %.3f ms/call (host %.3f) -> %.0f episodes/s' % (shot, n, (t2 - t0) / calls * 1e3, (t1 - t0) / calls * 1e3, B * calls / (t2 - t0)), flush=True)
