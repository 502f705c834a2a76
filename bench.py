#!/usr/bin/env python
"""bench.py — CMVAE episodic hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

Workload (BASELINE.json metric "train episodes/sec + eval EM episodes/sec, Omniglot 20-way"):
Omniglot 20-way training step — meta-batch of 4 episodes per GPU, sample-size 300, 32 MC samples,
10 unsupervised EM iterations, conv VAE + NOTEARS-MLP causal layer, Adam(lr 1e-3, wd 5e-5) —
i.e. the reference's `python main.py --way 20 --sample-size 300` step (omniglot/trainer.py:79-95).
One "step" = forward + backward + gradient all-reduce + optimizer step over one synthetic
meta-batch that is already resident in HBM.  Weak scaling: per-GPU work fixed.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WAY, SAMPLE, MC, B_PER_GPU, EM_ITERS, QUERY = 20, 300, 32, 4, 10, 15
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32-input MFMA dense peak


def synth_images(n, s, device, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.rand(n, s, 1, 28, 28, generator=g) < 0.15).float().to(device)


def build_model(device):
    from cmvae_amd.model import OmniglotGMVAE
    torch.manual_seed(10086)
    m = OmniglotGMVAE(input_shape=[1, 28, 28], unsupervised_em_iters=EM_ITERS, semisupervised_em_iters=EM_ITERS,
                      fix_pi=False, hidden_size=64, component_size=WAY, latent_size=64,
                      train_mc_sample_size=MC, test_mc_sample_size=MC)
    return m.to(device)


def event_time_ms(fn, iters, stream):
    """Average duration of fn() measured with HIP events on the stream the kernels run on."""
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record(stream)
    for _ in range(iters):
        fn()
    end.record(stream)
    end.synchronize()
    return start.elapsed_time(end) / iters


def roofline_em_pass(device):
    """HBM-bound kernel of the EM loop: the fused E-step/M-step pass (em_pass_kernel), at the training shape
    (Omniglot 20-way: latency-bound, 10 MB per launch) and at the evaluation-sweep shape of the feature models
    (256 episodes x 75 queries x 256 MC samples, 5-way: 1.26 GB per launch).
    Algorithmic bytes per launch = B*(N*D*4 + K*(2D+2)*4)  (SURVEY.md §8d)."""
    from cmvae_amd import ops
    out = []
    for tag, (B, N, K) in (('train O20', (B_PER_GPU, SAMPLE * MC, WAY)), ('eval sweep 5-way', (256, 75 * 256, 5))):
        g = torch.Generator(device='cpu').manual_seed(1)
        z = torch.randn(B, N, 64, generator=g).to(device)
        mu = torch.randn(B, K, 64, generator=g).to(device)
        logw = torch.randn(B, K, generator=g).to(device)
        ws, _ = ops.em_pass(z, mu, logw)
        stream = torch.cuda.current_stream()
        ms = event_time_ms(lambda: ops.em_pass(z, mu, logw, ws), 50, stream)
        alg_bytes = B * (N * 64 * 4 + K * (2 * 64 + 2) * 4)
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        name = 'em_pass_kernel<%d, false, false>' % ((K + 1) // 2)     # the instantiation this shape launches
        traffic, src = pmc_traffic(lambda k: k == name)
        out.append({'kernel': 'em_pass_kernel', 'shape': tag, 'bound': 'hbm', 'achieved': round(achieved, 1),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                    'traffic_source': src,
                    'launch_us': round(ms * 1e3, 2), 'algorithmic_bytes': alg_bytes,
                    'note': 'back-to-back launches from Python: includes host launch gaps'})
        del z
    return out


def useful_flops(g, ncb):
    """Algorithmic FLOPs of one gather-GEMM launch: 2 * 64 * 64 * (in-image taps over all rows) * column blocks
    (taps that fall in the zero padding are not counted)."""
    n_img = g.rows_total // (g.Hg * g.Wg)
    taps = 0
    for t in range(g.ntaps):
        ny = sum(1 for y in range(g.Hg) if 0 <= y * g.sy + g.dy[t] < g.Hi)
        nx = sum(1 for x in range(g.Wg) if 0 <= x * g.sx + g.dx[t] < g.Wi)
        taps += ny * nx
    return 2.0 * 64 * 64 * taps * n_img * ncb


def pmc_traffic(select):
    """HBM bytes per launch, launch-weighted over the kernels whose rocprof name satisfies `select`, from the newest
    committed rocprofv3 PMC summary (tools/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes over this same
    workload, gfx950 correction applied).  PMC counters cannot be read from inside this process."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return None, None
    ks = json.load(open(files[-1]))['kernels']
    sel = [v for k, v in ks.items() if select(k.rstrip())]
    n = sum(v['launches'] for v in sel)
    if not n:
        return None, None
    return sum(v['hbm_bytes_per_launch'] * v['launches'] for v in sel) / n, os.path.basename(files[-1])


def pmc_traffic_conv_gemm():
    """The conv launches of conv_gemm_wstat_kernel / conv_gemm_direct_kernel / conv_gemm_kernel (the last two with
    template argument LINEAR = false): the launches roofline_conv_gemm times."""
    return pmc_traffic(lambda k: k.startswith('conv_gemm_wstat_kernel') or
                       (k.startswith(('conv_gemm_kernel', 'conv_gemm_direct_kernel')) and k.endswith('false>')))


def roofline_conv_gemm(step_fn):
    """Dominant kernel of the step: conv_gemm_kernel (fp32 MFMA gather-GEMM: all decoder / encoder convs and
    their data gradients).  Every launch of one extra, untimed training step is bracketed by HIP events on
    the launch stream; achieved = sum of algorithmic FLOPs / sum of launch durations."""
    from cmvae_amd import conv
    rec = []
    orig = conv.conv_gemm

    def timed(g, A, Wt, out, bias=None, a_scale=None, a_shift=None, act=0, n_col_blocks=1, stats=False):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(g, A, Wt, out, bias=bias, a_scale=a_scale, a_shift=a_shift, act=act, n_col_blocks=n_col_blocks,
                 stats=stats)
        e1.record()
        rec.append((e0, e1, useful_flops(g, n_col_blocks)))
        return r

    conv.conv_gemm = timed
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        conv.conv_gemm = orig
    total_ms = sum(a.elapsed_time(b) for a, b, _ in rec)
    total_flops = sum(f for _, _, f in rec)
    achieved = total_flops / (total_ms * 1e-3) / 1e12
    traffic, src = pmc_traffic_conv_gemm()
    return {'kernel': 'conv_gemm_wstat_kernel + conv_gemm_direct_kernel (conv / conv-transpose / data-gradient launches)', 'bound': 'mfma',
            'achieved': round(achieved, 2),
            'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
            'traffic': traffic, 'traffic_unit': 'HBM bytes per launch (PMC)', 'traffic_source': src,
            'launches_per_step': len(rec), 'avg_launch_us': round(total_ms * 1e3 / len(rec), 1),
            'algorithmic_gflop_per_launch': round(total_flops / len(rec) / 1e9, 2),
            'share_of_step_ms': round(total_ms, 2)}


def cpu_baseline(threads):
    """The CPU oracle (a restatement of the reference's PyTorch op sequence) on the same workload,
    bounded to ONE training step of 4 episodes (about 10-30 s of CPU work)."""
    from cmvae_amd.model import OmniglotGMVAE
    from oracle.gmvae_oracle import OracleGMVAE, make_trainable, train_step
    torch.set_num_threads(threads)
    torch.manual_seed(10086)
    np.random.seed(10086)
    m = OmniglotGMVAE(unsupervised_em_iters=EM_ITERS, semisupervised_em_iters=EM_ITERS, component_size=WAY,
                      train_mc_sample_size=MC, test_mc_sample_size=MC)
    P = make_trainable({k: v.detach().clone() for k, v in m.state_dict().items()})
    oracle = OracleGMVAE(P, 'omniglot', unsupervised_em_iters=EM_ITERS, semisupervised_em_iters=EM_ITERS,
                         component_size=WAY, train_mc_sample_size=MC, test_mc_sample_size=MC)
    opt = torch.optim.Adam([p for p in P.values() if p.requires_grad], lr=1e-3, weight_decay=5e-5)
    X = synth_images(B_PER_GPU, SAMPLE, 'cpu', 7)
    # tiny warm-up (thread pool, allocator) on 1 episode of 20 items; not part of the sample
    small = OracleGMVAE(P, 'omniglot', unsupervised_em_iters=1, component_size=WAY, train_mc_sample_size=2)
    small.forward(X[:1, :24], 0.1, 0.1, 1e-4)
    t0 = time.perf_counter()
    train_step(oracle, opt, X, 0.1, 0.1, 1e-4)
    dt = time.perf_counter() - t0
    return {'value': round(B_PER_GPU / dt, 4), 'unit': 'train episodes/s', 'cores': threads, 'kind': 'port',
            'sample': '1 training step (4 episodes) of the same Omniglot 20-way workload, oracle/gmvae_oracle.py',
            'seconds': round(dt, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eval-calls', type=int, default=10)
    args = ap.parse_args()

    t_start = time.perf_counter()
    from cmvae_amd import dist as cdist
    rank, world = cdist.init_from_env()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run' % args.gpus)
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)

    model = build_model(device)
    flat = cdist.FlatGrads(model.parameters())
    from cmvae_amd.linear import FlatAdam
    opt = FlatAdam(flat, lr=1e-3, weight_decay=5e-5)          # Adam as in omniglot/trainer.py:48-52
    np.random.seed(10086)
    torch.manual_seed(10086 + rank)

    # synthetic meta-batches, resident in HBM before the timed region; each rank owns its shard
    n_pool = 4
    pool = [synth_images(B_PER_GPU * world, SAMPLE, 'cpu', 100 + i)[cdist.shard_slice(B_PER_GPU * world, rank, world)]
            .to(device) for i in range(n_pool)]
    h_lambda, h_c, a_lambda = 0.1, 0.1, 1e-4
    N = SAMPLE * MC

    seeds = cdist.SeedUploader((B_PER_GPU, WAY), device)

    def step(i):
        model.train()
        flat.zero()
        X = pool[i % n_pool]
        idx = seeds.upload(cdist.draw_em_init(N, WAY, B_PER_GPU * world, rank, world))
        _, rec, kl, h_aug, _ = model(X, h_lambda, h_c, a_lambda, init_idx=idx)
        (rec + kl + h_aug).backward()
        flat.allreduce()
        opt.step()
        return rec, kl, h_aug

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def note(msg):
        if rank == 0:
            print('[bench %7.1fs] %s' % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        note('warmup step %d done' % i)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(i)
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    losses = [float(x) for x in last]
    note('timed region done: %.2f ms/step' % (dt / args.steps * 1e3))
    ms_per_step = dt / args.steps * 1e3
    train_eps = B_PER_GPU * world * args.steps / dt

    # eval: GMVAE.prediction, 20-way, query 15; B=4 episodes per call (the reference's default batch size) and
    # B=32 per call (same episodes, larger calls: what Trainer.eval uses with --eval-batch-size 32)
    eval_rates, eval_rates_b32 = {}, {}
    model.eval()
    for B_eval, rates in ((B_PER_GPU, eval_rates), (32, eval_rates_b32)):
        for shot in (1, 5):
            tr, te = WAY * shot, WAY * QUERY
            Xtr = synth_images(B_eval, tr, device, 200 + shot)
            Xte = synth_images(B_eval, te, device, 300 + shot)
            ytr = torch.arange(WAY, device=device).repeat_interleave(shot)[None].repeat(B_eval, 1)
            for _ in range(2):
                model.prediction(Xtr, ytr, Xte)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.eval_calls):
                y = model.prediction(Xtr, ytr, Xte)
            barrier()
            torch.cuda.synchronize()
            edt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([edt], device=device, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                edt = float(t)
            rates['%dshot' % shot] = round(B_eval * world * args.eval_calls / edt, 2)
            note('eval B=%d %d-shot done' % (B_eval, shot))

    # roofline of the dominant kernel: one extra, untimed, instrumented training step (all ranks take part:
    # the step contains the BatchNorm-statistics and gradient all-reduces)
    roof = roofline_conv_gemm(lambda: step(0))
    if rank == 0:
        out = {
            'metric': 'train episodes/sec (whole node) + eval EM episodes/sec, Omniglot 20-way',
            'value': round(train_eps, 3), 'unit': 'train episodes/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'Omniglot 20-way train step: 4 episodes/GPU x sample-size 300 x 32 MC, '
                                   '10 EM iters, conv VAE + NOTEARS-MLP causal, Adam (configs[2]; the metric\'s config)',
                       'global_batch_episodes': B_PER_GPU * world, 'parallelism': 'dp%d' % world},
            'eval_episodes_per_s': eval_rates, 'eval_episodes_per_s_batch32': eval_rates_b32,
            'final_losses': {'rec': losses[0], 'kl': losses[1], 'h_aug': losses[2]},
        }
        out['roofline'] = roof
        out['roofline_other'] = roofline_em_pass(device)
        note('roofline measurements done; starting cpu baseline')
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, int(os.environ.get('OMP_NUM_THREADS', '16')))
            cb = cpu_baseline(threads)
            out['cpu_baseline'] = cb
            out['speedup_vs_cpu'] = round(train_eps / cb['value'], 1)
        print(json.dumps(out), flush=True)
    barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
