#!/usr/bin/env python
"""bench.py — CMVAE episodic hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config o20|o5|mi|ce_sweep]

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its own N ranks (a fresh
`python -m torch.distributed.run --nproc-per-node N bench.py ...` child, started before anything touches the GPU)
and relays rank 0's JSON line; under torch.distributed.run it is one rank of the job (one process per GPU, RCCL).

Workloads (BASELINE.json; --config):
  o20 (default, the metric's config)  Omniglot 20-way training step — 4 episodes per GPU, sample-size 300, 32 MC
        samples, 10 unsupervised EM iterations, conv VAE + NOTEARS-MLP causal layer, Adam(lr 1e-3, wd 5e-5), i.e.
        `python main.py --way 20 --sample-size 300` (omniglot/trainer.py:79-95) — plus GMVAE.prediction, 1-/5-shot;
  o5    the same with --way 5 --sample-size 200 (configs[1]);
  mi    Mini-ImageNet 5-way training step on synthetic SimCLR features: 16 episodes per GPU x 5 items x 256 MC,
        linear causal layer, lr 1e-4 (mimgnet/main.py:28-45), plus prediction at 1-/5-shot (75 queries);
  ce_sweep  CelebA 5-way evaluation sweep: NOTEARS-MLP causal layer, 4096 test episodes per shot in calls of 256
        episodes (semi-supervised EM batched over the call); `value` is eval episodes/s there.
One "step" = forward + backward + gradient all-reduce + optimizer step over one synthetic meta-batch that is already
resident in HBM.  --scaling weak (default): per-GPU work fixed (the config's episodes per GPU); --scaling strong
--global-batch 32: a fixed global meta-batch split over the GPUs.  Prints ONE JSON line on rank 0; next to `value`
it carries host_ms_per_step (the loop that issues the steps), the slowest / fastest rank's own step time and
steps_long_ms (a second timed region of --long-steps steps).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32-input MFMA dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # bf16 MFMA dense peak (MI355X_MICROARCH.md, matrix cores)
# fp32 products on the bf16 pipe cost six bf16 MFMA products each (three-piece split, csrc/conv_gemm.hip): the pipe's peak
# in fp32-equivalent (algorithmic) FLOP/s
BF16_PIPE_F32_EQUIV_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0
EM_ITERS, QUERY = 10, 15
PMC_CONFIG = 'o20'             # which committed profiles/r*_<config>_pmc_traffic.json the roofline traffic is read from
WAY, SAMPLE, MC, B_PER_GPU = 20, 300, 32, 4        # the default (o20) shape, for the probes under tools/

CONFIGS = {
    'o20': dict(variant='omniglot', way=20, sample=300, mc=32, batch=4, lr=1e-3, consts=(0.1, 0.1, 1e-4),
                workload='Omniglot 20-way train step: 4 episodes/GPU x sample-size 300 x 32 MC, 10 EM iters, conv VAE '
                         '+ NOTEARS-MLP causal, Adam (configs[2]; the metric\'s config)'),
    'o5': dict(variant='omniglot', way=5, sample=200, mc=32, batch=4, lr=1e-3, consts=(0.1, 0.1, 1e-4),
               workload='Omniglot 5-way train step: 4 episodes/GPU x sample-size 200 x 32 MC, 10 EM iters, conv VAE + '
                        'NOTEARS-MLP causal, Adam (configs[1])'),
    'mi': dict(variant='feat', linear=True, way=5, sample=5, mc=256, batch=16, lr=1e-4, consts=(0.1, 0.1, 1e-3), feat=256,
               workload='Mini-ImageNet 5-way train step on synthetic SimCLR features: 16 episodes/GPU x 5 items x 256 '
                        'MC, MLP VAE + linear causal layer, Adam (configs[3])'),
    'ce_sweep': dict(variant='feat', linear=False, way=5, sample=5, mc=256, batch=16, lr=1e-4, consts=(1.0, 1.0, 1e-4),
                     feat=256, eval_only=True, sweep_call=256, sweep_episodes=4096,
                     workload='CelebA 5-way evaluation sweep on synthetic SimCLR features: 4096 test episodes per shot '
                              'in calls of 256 episodes, 75 queries x 256 MC, NOTEARS-MLP causal layer (configs[4])'),
}


def synth_images(n, s, device, seed):
    import torch
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.rand(n, s, 1, 28, 28, generator=g) < 0.15).float().to(device)


def synth_features(n, s, f, device, seed):
    import torch
    g = torch.Generator(device='cpu').manual_seed(seed)
    return torch.relu(torch.randn(n, s, f, generator=g)).to(device)       # SimCLR features are post-ReLU


def synth_batch(cfg, n, s, device, seed):
    if cfg['variant'] == 'omniglot':
        return synth_images(n, s, device, seed)
    return synth_features(n, s, cfg['feat'], device, seed)


def model_kwargs(cfg):
    return dict(unsupervised_em_iters=EM_ITERS, semisupervised_em_iters=EM_ITERS, fix_pi=False,
                component_size=cfg['way'], latent_size=64, train_mc_sample_size=cfg['mc'],
                test_mc_sample_size=cfg['mc'])


def build_model(device, cfg=None):
    import torch
    from cmvae_amd.model import OmniglotGMVAE, FeatureGMVAE
    cfg = cfg or CONFIGS['o20']
    torch.manual_seed(10086)
    if cfg['variant'] == 'omniglot':
        m = OmniglotGMVAE(input_shape=[1, 28, 28], hidden_size=64, **model_kwargs(cfg))
    else:
        m = FeatureGMVAE(input_shape=cfg['feat'], linear=cfg['linear'], **model_kwargs(cfg))
    return m.to(device)


def event_time_ms(fn, iters, stream):
    """Average duration of fn() measured with HIP events on the stream the kernels run on."""
    import torch
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


def roofline_em_pass(device, shapes):
    """HBM-bound kernel of the EM loop: the fused E-step/M-step pass (em_pass_kernel).
    Algorithmic bytes per launch = B*(N*D*4 + K*(2D+2)*4)  (SURVEY.md §8d)."""
    import torch
    from cmvae_amd import ops
    out = []
    for tag, (B, N, K) in shapes:
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
        # shapes that share an instantiation share its PMC record: the largest takes the record's largest launch, the
        # others report no traffic (the launch-weighted mean would mix the shapes)
        same = [b * (n * 64 * 4 + k * 130 * 4) for _, (b, n, k) in shapes if (k + 1) // 2 == (K + 1) // 2]
        if len(same) == 1:
            traffic, src = pmc_traffic(lambda k: k == name)
        elif alg_bytes == max(same):
            traffic, src = pmc_traffic(lambda k: k == name, largest=True)
        else:
            traffic, src = None, None
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


def pmc_traffic(select, largest=False):
    """HBM bytes per launch, launch-weighted over the kernels whose rocprof name satisfies `select`, from the newest
    committed rocprofv3 PMC summary (tools/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes over this same
    workload, gfx950 correction applied).  PMC counters cannot be read from inside this process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_%s_pmc_traffic.json' % PMC_CONFIG))) or \
        (sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json'))) if PMC_CONFIG == 'o20' else [])
    if not files:
        return None, None
    ks = json.load(open(files[-1]))['kernels']
    sel = [v for k, v in ks.items() if select(k.rstrip())]
    n = sum(v['launches'] for v in sel)
    if not n:
        return None, None
    if largest:
        return max(v['max_launch_read_bytes'] + v['max_launch_write_bytes'] for v in sel), os.path.basename(files[-1])
    return sum(v['hbm_bytes_per_launch'] * v['launches'] for v in sel) / n, os.path.basename(files[-1])


def _template_args(kernel_name):
    i, j = kernel_name.find('<'), kernel_name.rfind('>')
    if i < 0 or j < i:
        return []
    return [a.strip() for a in kernel_name[i + 1:j].split(',')]


def _is_bf16_pipe(k):
    """Kernel names of the launches whose fp32 products run on the bf16 matrix pipe: the BF3 instantiations of the
    weight-stationary kernels (last template argument) and the bf16-pipe weight gradient."""
    a = _template_args(k)
    if k.startswith('conv_wgrad_bf3_multi_kernel'):
        return True
    if k.startswith('conv_gemm_wstat_multi_kernel'):
        return len(a) == 4 and a[3] == 'true'
    if k.startswith('conv_gemm_wstat_kernel'):
        return len(a) == 7 and a[6] == 'true'
    return False


def _is_conv_gemm(k):
    """Kernel names of the fp32-MFMA conv / conv-transpose / data-gradient launches: conv_gemm_wstat_multi_kernel,
    conv_gemm_wstat_kernel<ACT = 0, ...> (an activation marks an nn.Linear launch), conv_gemm_direct_kernel<ACT,
    LINEAR = false, ...>, conv_gemm_kernel — without the bf16-pipe instantiations."""
    a = _template_args(k)
    if _is_bf16_pipe(k):
        return False
    if k.startswith('conv_gemm_wstat_multi_kernel') or k.startswith('conv_gemm_kernel'):
        return True
    if k.startswith('conv_gemm_wstat_kernel'):
        return len(a) > 0 and a[0] == '0'
    if k.startswith('conv_gemm_direct_kernel'):
        return len(a) > 1 and a[1] == 'false'
    return False


def _is_linear_gemm(k):
    a = _template_args(k)
    if k.startswith('conv_gemm_direct_kernel'):
        return len(a) > 1 and a[1] == 'true'
    if k.startswith('conv_gemm_wstat_kernel'):
        # (an nn.Linear launch without activation on the bf16 pipe shares its kernel name with the first pass of an
        # image-major 3x3 conv there; the feature models have no such conv, the Omniglot roofline groups by call)
        return len(a) > 0 and a[0] != '0'
    return False


def roofline_conv_gemm(step_fn):
    """Dominant kernels of the Omniglot step: the gather-GEMM contractions of the conv stacks.  Every launch of one
    extra, untimed training step is bracketed by HIP events on the launch stream; per group, achieved = sum of
    algorithmic FLOPs / sum of launch durations.  Two groups, by the matrix pipe the products run on:
      * the decoder's big layers (forward of the four sub-pixel phases, data gradient, weight gradient) run fp32
        products as six bf16 MFMA products of exact three-piece splits: priced against the bf16 dense peak / 6, with
        the fraction of the fp32-MFMA peak next to it (it can exceed 1: that pipe is not the one in use);
      * every other conv launch runs on the fp32 MFMA: priced against its peak.
    Returns (dominant group, other group)."""
    import torch
    from cmvae_amd import conv
    rec = {'bf16': [], 'f32': []}
    names = ('conv_gemm', 'conv_gemm_multi', 'conv_gemm_bnbwd', 'conv_wgrad_multi')
    origs = {n: getattr(conv, n) for n in names}

    pipe = lambda kind, gs: 'bf16' if conv.on_bf16_pipe(kind, gs) else 'f32'      # noqa: E731

    def timed(name, flops_of, group_of):
        orig = origs[name]

        def f(*args, **kw):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*args, **kw)
            e1.record()
            grp = group_of(*args, **kw)
            if grp:
                # kernel launches behind the call: more than six taps on the bf16 pipe run as accumulating passes
                # (rocprof counts those)
                nk = 1
                if grp == 'bf16' and name in ('conv_gemm_bnbwd', 'conv_gemm'):
                    nk = conv.bf16_pipe_passes('bnbwd' if name == 'conv_gemm_bnbwd' else 'gemm', args[0])
                rec[grp].append((e0, e1, flops_of(*args, **kw), nk))
            return r
        return f

    conv.conv_gemm = timed('conv_gemm', lambda g, *a, **kw: useful_flops(g, kw.get('n_col_blocks', 1)),
                           lambda g, *a, **kw: pipe('gemm', [g]) if (kw.get('n_col_blocks', 1) == 1 and
                                                                     kw.get('act', 0) == 0) else 'f32')
    conv.conv_gemm_multi = timed('conv_gemm_multi', lambda gs, *a, **kw: sum(useful_flops(g, 1) for g in gs),
                                 lambda gs, *a, **kw: pipe('multi', gs))
    conv.conv_gemm_bnbwd = timed('conv_gemm_bnbwd', lambda g, *a, **kw: useful_flops(g, 1),
                                 lambda g, *a, **kw: pipe('bnbwd', [g]))
    # weight gradients: the bf16-pipe launches belong to the dominant group; the fp32 streaming launches are left out
    # of the fp32 group (they were never part of it: it prices the forward / data-gradient kernel family)
    conv.conv_wgrad_multi = timed('conv_wgrad_multi', lambda gs, *a, **kw: sum(useful_flops(g, 1) for g in gs),
                                  lambda gs, *a, **kw: 'bf16' if conv.on_bf16_pipe('wgrad_multi', gs) else None)
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        for n, f in origs.items():
            setattr(conv, n, f)

    def group(key, kernel, select, peak, extra):
        r = rec[key]
        if not r:
            return None
        total_ms = sum(a.elapsed_time(b) for a, b, _, _ in r)
        total_flops = sum(f for _, _, f, _ in r)
        nk = sum(n for _, _, _, n in r)
        achieved = total_flops / (total_ms * 1e-3) / 1e12
        traffic, src = pmc_traffic(select)
        d = {'kernel': kernel, 'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': round(peak, 1),
             'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': traffic,
             'traffic_unit': 'HBM bytes per launch (PMC)', 'traffic_source': src, 'launches_per_step': nk,
             'avg_launch_us': round(total_ms * 1e3 / nk, 1),
             'algorithmic_gflop_per_launch': round(total_flops / nk / 1e9, 2), 'share_of_step_ms': round(total_ms, 2)}
        d.update(extra(achieved))
        return d

    bf = group('bf16', 'conv_gemm_wstat(_multi)_kernel<.., BF3> + conv_wgrad_bf3_multi_kernel (decoder layers 2-3: forward, '
               'data gradient, weight gradient; encoder 3x3 conv at 14x14: forward, data gradient; fp32 products as six '
               'bf16 MFMA products)', _is_bf16_pipe,
               BF16_PIPE_F32_EQUIV_PEAK_TFLOPS,
               lambda a: {'achieved_unit_note': 'algorithmic (fp32) FLOP/s', 'peak_note': 'bf16 dense MFMA peak 2500 / 6 '
                          'products per fp32 product', 'frac_of_fp32_mfma_peak': round(a / MFMA_F32_PEAK_TFLOPS, 4)})
    f32 = group('f32', 'conv_gemm_wstat(_multi)_kernel + conv_gemm_direct_kernel on the fp32 MFMA (the other conv / '
                'conv-transpose / data-gradient launches)', _is_conv_gemm, MFMA_F32_PEAK_TFLOPS, lambda a: {})
    if bf is None:
        return f32, None
    return bf, f32


def roofline_linear(step_fn):
    """Dominant kernel family of the feature-model step: the nn.Linear forward launches (cmvae_linear_fwd ->
    conv_gemm_direct_kernel<.., LINEAR = true>): proj, the SAB projections, the (mu, logvar) head, causal fc1.
    Algorithmic FLOPs per launch = 2 * rows * in_features * out_features."""
    import torch
    from cmvae_amd import linear as lin
    rec = []
    orig = lin.linear_fwd

    def timed(x2, weight, bias, act, out=None):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(x2, weight, bias, act, out)
        e1.record()
        rec.append((e0, e1, 2.0 * x2.shape[0] * x2.shape[1] * weight.shape[0]))
        return r

    lin.linear_fwd = timed
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        lin.linear_fwd = orig
    total_ms = sum(a.elapsed_time(b) for a, b, _ in rec)
    total_flops = sum(f for _, _, f in rec)
    achieved = total_flops / (total_ms * 1e-3) / 1e12
    traffic, src = pmc_traffic(_is_linear_gemm)
    return {'kernel': 'nn.Linear forward launches: conv_gemm_direct_kernel<LINEAR> on the fp32 MFMA; from 16384 rows on '
                      '(in_features <= 384) conv_gemm_wstat_kernel<.., BF3> on the bf16 pipe — priced against the fp32-MFMA '
                      'peak throughout', 'bound': 'mfma',
            'achieved': round(achieved, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': traffic,
            'traffic_unit': 'HBM bytes per launch (PMC)', 'traffic_source': src, 'launches_per_step': len(rec),
            'avg_launch_us': round(total_ms * 1e3 / len(rec), 1),
            'algorithmic_gflop_per_launch': round(total_flops / len(rec) / 1e9, 3),
            'share_of_step_ms': round(total_ms, 2)}


# ----------------------------------------------------------------------------- CPU baseline (oracle, bounded sample)
def _oracle(cfg, trainable):
    import numpy as np
    import torch
    from cmvae_amd.model import OmniglotGMVAE, FeatureGMVAE
    from oracle.gmvae_oracle import OracleGMVAE, make_trainable
    torch.manual_seed(10086)
    np.random.seed(10086)
    kw = model_kwargs(cfg)
    if cfg['variant'] == 'omniglot':
        m = OmniglotGMVAE(**kw)
        okw = dict(input_shape=(1, 28, 28))
    else:
        m = FeatureGMVAE(input_shape=cfg['feat'], linear=cfg['linear'], **kw)
        okw = dict(input_shape=cfg['feat'], linear=cfg['linear'])
    P = {k: v.detach().clone() for k, v in m.state_dict().items()}
    if trainable:
        P = make_trainable(P)
    del kw['latent_size']
    return OracleGMVAE(P, cfg['variant'], **okw, **kw), P


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline_train(cfg, threads, timed_steps=3):
    """The CPU oracle (a restatement of the reference's PyTorch op sequence) on the same training workload:
    1 warm-up step + `timed_steps` timed steps of one meta-batch."""
    import torch
    from oracle.gmvae_oracle import train_step
    torch.set_num_threads(threads)
    oracle, P = _oracle(cfg, True)
    opt = torch.optim.Adam([p for p in P.values() if p.requires_grad], lr=cfg['lr'], weight_decay=5e-5)
    X = synth_batch(cfg, cfg['batch'], cfg['sample'], 'cpu', 7)
    train_step(oracle, opt, X, *cfg['consts'])                       # warm-up: thread pool, allocator, oneDNN plans
    t0 = time.perf_counter()
    for _ in range(timed_steps):
        train_step(oracle, opt, X, *cfg['consts'])
    dt = (time.perf_counter() - t0) / timed_steps
    return {'value': round(cfg['batch'] / dt, 4), 'unit': 'train episodes/s', 'cores': threads, 'kind': 'port',
            'sample': '1 warm-up + %d timed training steps (%d episodes each) of the same workload, '
                      'oracle/gmvae_oracle.py' % (timed_steps, cfg['batch']),
            'seconds_per_step': round(dt, 2)}


def cpu_baseline_eval(cfg, threads, shots=(1, 5), calls=3, batch=None):
    """GMVAE.prediction on the CPU oracle: `calls` calls per shot of `batch` episodes."""
    import torch
    torch.set_num_threads(threads)
    oracle, _ = _oracle(cfg, False)
    oracle.training = False
    B = batch or cfg['batch']
    way = cfg['way']
    out = {'unit': 'eval episodes/s', 'cores': threads, 'kind': 'port',
           'sample': '%d prediction calls per shot, %d episodes per call, oracle/gmvae_oracle.py' % (calls, B)}
    for shot in shots:
        tr, te = way * shot, way * QUERY
        X = synth_batch(cfg, B, tr + te, 'cpu', 200 + shot)
        ytr = torch.arange(way).repeat_interleave(shot)[None].repeat(B, 1)
        oracle.prediction(X[:, :tr], ytr, X[:, tr:])                  # warm-up
        t0 = time.perf_counter()
        for _ in range(calls):
            oracle.prediction(X[:, :tr], ytr, X[:, tr:])
        dt = (time.perf_counter() - t0) / calls
        out['%dshot' % shot] = round(B / dt, 4)
        out['%dshot_seconds_per_call' % shot] = round(dt, 2)
    return out


# CMVAE_BENCH_REHEARSE=1: every rank on GPU 0, collectives over gloo (RCCL refuses two ranks on one device) — walks
# the N > 1 code path of this file on a one-GPU box; the numbers of such a run mean nothing and the line says so
REHEARSE = os.environ.get('CMVAE_BENCH_REHEARSE', '') == '1'


def dump_thread_states(path):
    """Where every native thread of this process sleeps (name, state, kernel wait channel, pending system call) —
    what faulthandler cannot see: the gloo / RCCL worker threads.  Reads /proc only."""
    rows = []
    base = '/proc/self/task'
    for tid in sorted(os.listdir(base), key=int):
        def rd(name):
            try:
                with open('%s/%s/%s' % (base, tid, name)) as f:
                    return f.read().strip()
            except OSError as e:
                return '<%s>' % e.__class__.__name__
        st = rd('stat')
        state = st[st.rfind(')') + 2:].split(' ', 1)[0] if ')' in st else '?'
        rows.append('tid %s comm=%s state=%s wchan=%s syscall=%s' % (tid, rd('comm'), state, rd('wchan'),
                                                                    ' '.join(rd('syscall').split()[:4])))
    with open(path, 'a') as f:
        f.write('\n'.join(rows) + '\n')
        f.flush()
        try:                                   # last: a HIP call from here may block as well
            import torch
            f.write('default stream idle: %s\n' % torch.cuda.default_stream().query())
        except Exception as e:                 # noqa: BLE001
            f.write('stream query failed: %r\n' % (e,))


# ----------------------------------------------------------------------------- self-launch
def self_launch(args):
    """--gpus N > 1 without a torchrun environment: start N ranks in a fresh child before this process touches the
    GPU (never re-exec a process that has initialised it) and pass its output through."""
    import torch
    n_dev = torch.cuda.device_count()                       # does not initialise the GPU
    if n_dev < args.gpus and not REHEARSE:
        print('bench.py: --gpus %d but only %d GPU(s) visible' % (args.gpus, n_dev), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1) // 2)))
    if REHEARSE and os.environ.get('CMVAE_REHEARSE_HW_QUEUES'):
        env['GPU_MAX_HW_QUEUES'] = os.environ['CMVAE_REHEARSE_HW_QUEUES']
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='o20')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph-step', choices=('auto', 'on', 'off'), default='auto',
                    help='training step replayed from a HIP graph (auto: feature configs on one GPU, as Trainer does)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak: the config\'s episodes per GPU (4 for Omniglot, 16 for the feature models); strong: '
                         '--global-batch episodes per step split over the GPUs (SURVEY.md §8d item 3)')
    ap.add_argument('--global-batch', type=int, default=32, help='episodes per step of the whole job under --scaling strong')
    ap.add_argument('--long-steps', type=int, default=500,
                    help='a second, longer timed region after the K-step one (steps_long_ms in the line); 0 = off')
    ap.add_argument('--eval-calls', type=int, default=40)
    ap.add_argument('--eval-streams', type=int, default=8,
                    help='HIP streams the independent prediction calls are dealt to, as Trainer.eval does (1 = sequential)')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    global PMC_CONFIG
    PMC_CONFIG = args.config

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))

    import numpy as np
    import torch
    t_start = time.perf_counter()
    from cmvae_amd import dist as cdist
    if REHEARSE:
        os.environ['LOCAL_RANK'] = '0'
        import faulthandler
        import threading
        limit = int(os.environ.get('CMVAE_BENCH_REHEARSE_LIMIT', '600'))
        faulthandler.dump_traceback_later(limit, exit=True)             # a rank that hangs says where
        tdir = os.environ.get('CMVAE_DIST_TRACE', '')
        if tdir:
            os.makedirs(tdir, exist_ok=True)
            wd = threading.Timer(max(1, limit - 5), dump_thread_states,
                                 (os.path.join(tdir, 'threads_rank%s.txt' % os.environ.get('RANK', '0')),))
            wd.daemon = True
            wd.start()
    rank, world = cdist.init_from_env('gloo' if REHEARSE else None)
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but the job has %d ranks' % (args.gpus, world))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    ranks_seen = 1
    multi = cdist.exchanging()                               # several ranks (or CMVAE_DIST_FORCE=1 on one)
    if multi:
        t = torch.ones(1, device=device)
        cdist.all_reduce(t, 'ranks_seen')                     # the ranks RCCL actually connects
        ranks_seen = int(t.item())

    way, S, M, Bl = cfg['way'], cfg['sample'], cfg['mc'], cfg['batch']
    if args.scaling == 'strong':
        if args.global_batch % world:
            raise SystemExit('bench.py: --global-batch %d is not divisible by %d ranks' % (args.global_batch, world))
        Bl = args.global_batch // world
    model = build_model(device, cfg)
    n_tail = model.tail_param_count() if os.environ.get('CMVAE_OVERLAP_TAIL', '1') != '0' else 0
    flat = cdist.FlatGrads(model.parameters()).overlap_tail(n_tail)
    from cmvae_amd.linear import FlatAdam
    opt = FlatAdam(flat, lr=cfg['lr'], weight_decay=5e-5)          # Adam as in omniglot/trainer.py:48-52
    np.random.seed(10086)
    torch.manual_seed(10086)                                       # every rank alike, as main.py seeds them

    def barrier():
        if multi:
            torch.distributed.barrier()

    def note(msg):
        if rank == 0:
            print('[bench %7.1fs] %s' % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    def max_over_ranks(dt):
        if multi:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            cdist.all_reduce(t, 'max_over_ranks', op=torch.distributed.ReduceOp.MAX)
            return float(t)
        return dt

    h_lambda, h_c, a_lambda = cfg['consts']
    N = S * M
    out = {'metric': 'train episodes/sec (whole node) + eval EM episodes/sec, Omniglot 20-way', 'n_gpus': world,
           'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': True, 'scaling': args.scaling,
           'vs_baseline': None,
           'dtype': 'f32', 'data': 'synthetic' + (' (REHEARSAL: all ranks on one GPU over gloo)' if REHEARSE else ''),
           'rccl_ranks': ranks_seen,
           'config': {'workload': cfg['workload'], 'name': args.config, 'global_batch_episodes': Bl * world,
                      'episodes_per_gpu': Bl, 'parallelism': 'dp%d' % world}}
    step = eager_step = None
    if not cfg.get('eval_only'):
        # synthetic meta-batches, resident in HBM before the timed region; each rank owns its shard
        n_pool = 4
        pool = [synth_batch(cfg, Bl * world, S, 'cpu', 100 + i)[cdist.shard_slice(Bl * world, rank, world)].to(device)
                for i in range(n_pool)]
        # (a deep ring: the host is never held back by the upload of a step the GPU has not reached yet, so that
        # host_ms_per_step below is the host's own issue time)
        seeds = cdist.SeedUploader((Bl, way), device, depth=64)

        # what Trainer does (--graph-step auto): the feature models' step, bound by the host's launch rate when issued
        # eagerly, is replayed from a HIP graph (with several ranks: up to the gradient pack); the Omniglot step
        # stays eager
        graphed = None
        if (args.graph_step == 'on' and (not multi or cfg['variant'] != 'omniglot')) or \
                (args.graph_step == 'auto' and cfg['variant'] != 'omniglot'):
            from cmvae_amd.graph_step import GraphedStep
            graphed = GraphedStep(model, flat, opt)      # several ranks: the exchange and Adam follow the replay
        out['config']['graph_step'] = graphed is not None

        def step(i):
            if graphed is not None:
                idx = seeds.upload(cdist.draw_em_init(N, way, Bl * world, rank, world, prefetch=True))
                return graphed(pool[i % n_pool], idx, h_lambda, h_c, a_lambda)[:3]
            return eager_step(i)

        def eager_step(i):
            model.train()
            flat.zero()
            X = pool[i % n_pool]
            idx = seeds.upload(cdist.draw_em_init(N, way, Bl * world, rank, world, prefetch=True))
            eps = cdist.draw_eps(Bl * world, S, M, 64, device, rank, world) if multi else None
            _, rec, kl, h_aug, _ = model(X, h_lambda, h_c, a_lambda, eps=eps, init_idx=idx)
            (rec + kl + h_aug).backward()
            flat.allreduce()
            opt.step()
            return rec, kl, h_aug

        def min_over_ranks(dt):
            if multi:
                t = torch.tensor([dt], device=device, dtype=torch.float64)
                cdist.all_reduce(t, 'min_over_ranks', op=torch.distributed.ReduceOp.MIN)
                return float(t)
            return dt

        def timed(n_steps):
            """n_steps steps between two barrier + synchronize pairs: (seconds max over ranks, min over ranks, host
            seconds this rank spent issuing the steps, last losses)."""
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            marks = []
            for i in range(n_steps):
                last = step(i)
                marks.append(time.perf_counter())
            # the host's own issue time per step: the first steps after the synchronize, while the launch queue is still
            # shallow (later the runtime holds the host back to the GPU's pace and the loop time says nothing)
            k = min(3, n_steps)
            t_issued = (marks[k - 1] - t0) / k * n_steps
            torch.cuda.synchronize()
            t_rank = time.perf_counter() - t0          # this rank's own step time, before it waits for the others
            barrier()
            torch.cuda.synchronize()
            dt = max_over_ranks(time.perf_counter() - t0)
            return dt, max_over_ranks(t_rank), min_over_ranks(t_rank), max_over_ranks(t_issued), last

        for i in range(args.warmup):
            step(i)
            torch.cuda.synchronize()
            note('warmup step %d done' % i)
        dt, rank_max, rank_min, t_host, last = timed(args.steps)
        losses = [float(x.detach()) for x in last]
        note('timed region done: %.2f ms/step' % (dt / args.steps * 1e3))
        out.update(value=round(Bl * world * args.steps / dt, 3), unit='train episodes/s',
                   ms_per_step=round(dt / args.steps * 1e3, 3),
                   # host_ms_per_step: what the host needs to issue one step (launches are asynchronous; a value near
                   # ms_per_step means the job is bound by the host)
                   host_ms_per_step=round(t_host / args.steps * 1e3, 3),
                   step_ms_rank_max=round(rank_max / args.steps * 1e3, 3),
                   step_ms_rank_min=round(rank_min / args.steps * 1e3, 3),
                   final_losses={'rec': losses[0], 'kl': losses[1], 'h_aug': losses[2]})
        if graphed is not None:
            # the same step issued launch by launch (what the graph replaces): bound by the host's launch rate
            g_keep, graphed = graphed, None
            for i in range(3):
                step(i)
            out['eager_ms_per_step'] = round(timed(args.steps)[0] / args.steps * 1e3, 3)
            graphed = g_keep
            note('eager timed region done: %.2f ms/step' % out['eager_ms_per_step'])
        if args.long_steps > args.steps and not REHEARSE:
            dt_long = timed(args.long_steps)[0]
            out['steps_long'] = args.long_steps
            out['steps_long_ms'] = round(dt_long / args.long_steps * 1e3, 3)
            note('long timed region done: %.2f ms/step over %d steps' % (out['steps_long_ms'], args.long_steps))

    # eval: GMVAE.prediction, query 15; the reference's episodes per call and 32 per call (what Trainer.eval uses
    # with --eval-batch-size 32); the sweep config runs 256 episodes per call over 4096 episodes
    model.eval()
    eval_rates = {}
    if cfg.get('eval_only'):
        plans = [(cfg['sweep_call'], 'eval_episodes_per_s', max(1, cfg['sweep_episodes'] // cfg['sweep_call']))]
    else:
        plans = [(Bl, 'eval_episodes_per_s', args.eval_calls), (32, 'eval_episodes_per_s_batch32', args.eval_calls)]
    from cmvae_amd.pipeline import PredictionPipeline
    for B_eval, key, calls in plans:
        for n_streams, suffix in ((args.eval_streams, ''), (1, '_single_stream')):
            if n_streams == 1 and args.eval_streams == 1 and suffix:
                continue
            rates = {}
            for shot in (1, 5):
                if calls <= 0:
                    continue
                tr, te = way * shot, way * QUERY
                X = synth_batch(cfg, B_eval, tr + te, device, 200 + shot)
                Xtr, Xte = X[:, :tr].contiguous(), X[:, tr:].contiguous()
                ytr = torch.arange(way, device=device).repeat_interleave(shot)[None].repeat(B_eval, 1)
                pipe = PredictionPipeline(model, n_streams)       # as Trainer.eval issues its calls
                ncall = calls * (2 if n_streams > 1 else 1)
                for _ in range(2 * n_streams):
                    pipe.submit(Xtr, ytr, Xte)
                pipe.results()
                barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(ncall):
                    pipe.submit(Xtr, ytr, Xte)
                pipe.results()
                barrier()
                torch.cuda.synchronize()
                edt = max_over_ranks(time.perf_counter() - t0)
                rates['%dshot' % shot] = round(B_eval * world * ncall / edt, 2)
                note('eval B=%d %d-shot, %d stream(s) done' % (B_eval, shot, n_streams))
                del X, Xtr, Xte
            eval_rates[key + suffix] = rates
    out.update(eval_rates)
    if cfg.get('eval_only'):
        r = eval_rates['eval_episodes_per_s']
        calls = plans[0][2]
        out.update(value=r.get('5shot'), unit='eval episodes/s (5-shot)', steps=calls,
                   ms_per_step=round(cfg['sweep_call'] * world / r['5shot'] * 1e3, 3) if r.get('5shot') else None)

    # roofline of the dominant kernel: one extra, untimed, instrumented training step (all ranks take part: the step
    # contains the BatchNorm-statistics and gradient all-reduces); the sweep config reports the EM pass
    if step is not None:
        roof_more = None
        if cfg['variant'] == 'omniglot':
            roof, roof_more = roofline_conv_gemm(lambda: eager_step(0))
        else:
            roof = roofline_linear(lambda: eager_step(0))      # launch by launch: the eager step, also where the step is replayed
    else:
        roof, roof_more = None, None
    if rank == 0:
        em_shapes = [('train %s' % args.config, (Bl, N, way))]
        if args.config in ('o20', 'ce_sweep', 'mi'):
            em_shapes.append(('eval sweep 5-way', (256, 75 * 256, 5)))
        em_roof = roofline_em_pass(device, em_shapes)
        if roof is None:
            roof = em_roof[-1]
            em_roof = em_roof[:-1]
        out['roofline'] = roof
        out['roofline_other'] = ([roof_more] if roof_more else []) + em_roof
        note('roofline measurements done; starting cpu baseline')
        if world == 1 and not args.no_cpu_baseline:
            # the host's cores: every one of them unless the environment fixes the thread count (the GPU boxes of this
            # project export OMP_NUM_THREADS = the CPU share of one GPU's slot; os.cpu_count() shows the whole machine)
            threads = int(os.environ.get('OMP_NUM_THREADS') or 0) or (os.cpu_count() or 1)
            host = {'cpu_model': cpu_model(), 'cpu_count': os.cpu_count(), 'threads': threads}
            if cfg.get('eval_only'):
                cb = cpu_baseline_eval(cfg, threads, calls=2, batch=4)
                out['cpu_baseline'] = dict(cb, value=cb['5shot'], **host)
                out['speedup_vs_cpu'] = round(out['value'] / cb['5shot'], 1)
            else:
                cb = cpu_baseline_train(cfg, threads)
                out['cpu_baseline'] = dict(cb, **host)
                out['speedup_vs_cpu'] = round(out['value'] / cb['value'], 1)
                note('cpu train baseline done; starting cpu eval baseline')
                out['cpu_baseline_eval'] = cpu_baseline_eval(cfg, threads)
        print(json.dumps(out), flush=True)
    barrier()
    if REHEARSE:
        import faulthandler
        faulthandler.cancel_dump_traceback_later()
    if multi:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
