"""Per-launch time and useful TFLOP/s of every conv GEMM / weight-gradient call of one training step."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from cmvae_amd import conv, dist as cdist
dev = torch.device('cuda:0')
model = bench.build_model(dev)
flat = cdist.FlatGrads(model.parameters())
np.random.seed(1); torch.manual_seed(1)
X = bench.synth_images(bench.B_PER_GPU, bench.SAMPLE, 'cpu', 100).to(dev)
N = bench.SAMPLE * bench.MC
idx = torch.from_numpy(cdist.draw_em_init(N, bench.WAY, bench.B_PER_GPU, 0, 1)).to(dev)

def step():
    model.train(); flat.zero()
    _, rec, kl, h, _ = model(X, 0.1, 0.1, 1e-4, init_idx=idx)
    (rec + kl + h).backward()

for _ in range(2):
    step()
torch.cuda.synchronize()
rec = []
og, ow = conv.conv_gemm, conv.conv_wgrad
def tg(g, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = og(g, *a, **k); e1.record()
    ncb = k.get('n_col_blocks', 1)
    rec.append(('gemm ', g.rows_total, g.Hg, g.Hi, g.sy, g.ntaps, ncb, e0, e1, bench.useful_flops(g, ncb), k.get('a_scale') is not None))
    return r
def tw(g, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = ow(g, *a, **k); e1.record()
    rec.append(('wgrad', g.rows_total, g.Hg, g.Hi, g.sy, g.ntaps, 1, e0, e1, bench.useful_flops(g, 1), k.get('a_scale') is not None))
    return r
conv.conv_gemm, conv.conv_wgrad = tg, tw
import cmvae_amd.decoder as D, cmvae_amd.encoder as E
step(); torch.cuda.synchronize()
tot = {}
for kind, rows, Hg, Hi, sy, nt, ncb, e0, e1, fl, tr in rec:
    ms = e0.elapsed_time(e1)
    print('%s rows %8d grid %2d in %2d s%d taps %2d ncb %d T=%d  %8.1f us  %6.1f TF/s' % (kind, rows, Hg, Hi, sy, nt, ncb, tr, ms * 1e3, fl / ms / 1e9))
    a = tot.setdefault(kind, [0, 0]); a[0] += ms; a[1] += fl
for k, (ms, fl) in tot.items():
    print(k, 'total %.2f ms  %.1f TF/s' % (ms, fl / ms / 1e9))
