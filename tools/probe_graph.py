"""Does GMVAE.prediction capture into a HIP graph, and what does a replay cost next to the eager call?"""
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
        ref = model.prediction(Xtr, ytr, Xte)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); model.prediction(Xtr, ytr, Xte); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    eager = np.median(ts)
    eps = torch.randn(B, tr + te, model.train_mc_sample_size, model.latent_size, device=dev)
    ref = model.prediction(Xtr, ytr, Xte, eps=eps)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model.prediction(Xtr, ytr, Xte, eps=eps)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model.prediction(Xtr, ytr, Xte, eps=eps)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    same = bool((out == ref).all())
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    rep = np.median(ts)
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    thr = (time.perf_counter() - t0) / 50
    print('B %d shot %d: eager %.3f ms  replay %.3f ms (back-to-back %.3f ms)  labels identical %s  -> %.1f episodes/s'
          % (B, shot, 1e3 * eager, 1e3 * rep, 1e3 * thr, same, B / thr), flush=True)
