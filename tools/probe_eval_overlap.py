"""Few-shot evaluation on 8 streams + HIP graphs under `rocprofv3 --kernel-trace`: how busy is the GPU, and with how
many kernels at once?  (tools/trace_eval_overlap.sh runs it and evaluates the trace.)"""
import sys, time
import torch
sys.path.insert(0, '.')
import bench
from cmvae_amd.pipeline import PredictionPipeline
dev = torch.device('cuda:0')
model = bench.build_model(dev).eval()
B, shot, n = 4, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 8
tr, te = bench.WAY * shot, bench.WAY * bench.QUERY
Xtr = bench.synth_images(B, tr, dev, 201); Xte = bench.synth_images(B, te, dev, 301)
ytr = torch.arange(bench.WAY, device=dev).repeat_interleave(shot)[None].repeat(B, 1)
pipe = PredictionPipeline(model, n)
for _ in range(2 * n):
    pipe.submit(Xtr, ytr, Xte)
pipe.results(); torch.cuda.synchronize()
time.sleep(0.05)                       # a visible pause in the trace: the timed calls come after it
calls = 80
t0 = time.perf_counter()
for _ in range(calls):
    pipe.submit(Xtr, ytr, Xte)
pipe.results(); torch.cuda.synchronize()
print('ms/call %.3f' % ((time.perf_counter() - t0) / calls * 1e3))
