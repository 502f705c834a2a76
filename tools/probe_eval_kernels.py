"""Sum of kernel durations of ONE few-shot prediction call (run under `rocprofv3 --kernel-trace --stats`): what the
streams + graphs pipeline can approach at best when the kernels fill the GPU.  usage: probe_eval_kernels.py [B] [shot]"""
import sys
import torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
model = bench.build_model(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shot = int(sys.argv[2]) if len(sys.argv) > 2 else 1
CALLS = 20
tr, te = bench.WAY * shot, bench.WAY * bench.QUERY
Xtr = bench.synth_images(B, tr, dev, 200 + shot); Xte = bench.synth_images(B, te, dev, 300 + shot)
ytr = torch.arange(bench.WAY, device=dev).repeat_interleave(shot)[None].repeat(B, 1)
with torch.no_grad():
    for _ in range(CALLS):
        model.prediction(Xtr, ytr, Xte)
torch.cuda.synchronize()
print('calls', CALLS)
