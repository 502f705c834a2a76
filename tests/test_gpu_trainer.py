"""End-to-end Trainer runs on tiny synthetic pools: loop structure, checkpoints, final test (omniglot/trainer.py:66-240)."""
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(data_dir, save_dir, **kw):
    a = dict(data_dir=data_dir, save_dir=save_dir, unsupervised_em_iters=3, semisupervised_em_iters=3, fix_pi=False,
             hidden_size=64, latent_size=64, train_mc_sample_size=4, test_mc_sample_size=4, batch_size=2,
             sample_size=20, train_iters=4, freq_iters=2, eval_episodes=4, way=5, query=3, lr=1e-3, weight_decay=5e-5,
             debug=False, gpu_id=0, device='cuda:0', log_interval=1)
    a.update(kw)
    return SimpleNamespace(**a)


def _make(tmp_path, variant, *extra):
    out = str(tmp_path / variant)
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'make_synthetic_data.py'), variant, out,
                           '--train-classes', '8', '--val-classes', '8', '--test-classes', '8', *extra])
    return out


def test_omniglot_trainer_runs_and_checkpoints(tmp_path):
    from cmvae_amd.trainer import Trainer
    d = _make(tmp_path, 'omniglot')
    np.random.seed(10086); torch.manual_seed(10086)
    tr = Trainer(_args(d, str(tmp_path / 'run')), 'omniglot')
    w0 = tr.model.proj[0].weight.detach().clone()
    out = tr.train()
    assert set(out) == {1, 5} and all(0.0 <= m <= 1.0 for m, _ in out.values())
    assert not torch.equal(w0, tr.model.proj[0].weight) or tr.best_files       # parameters moved
    assert tr.best_files and all(os.path.exists(p) for p in tr.best_files.values())
    state = torch.load(next(iter(tr.best_files.values())), weights_only=True)
    assert set(state) >= {'state_dict', 'accuracy', 'epoch'}
    assert 'decoder.decoder.9.weight' in state['state_dict'] and 'causal.fc1.weight' in state['state_dict']
    # a fresh trainer reproduces the checkpointed model's predictions
    tr2 = Trainer(_args(d, str(tmp_path / 'run2')), 'omniglot')
    tr2.load_checkpoint(next(iter(tr.best_files.values())), resume_optimizer=True)
    for k, v in tr2.model.state_dict().items():
        assert torch.equal(v, state['state_dict'][k].to(v.device)), k
    lines = open(tr.log_path).read().strip().split('\n')
    assert len(lines) >= 4


@pytest.mark.parametrize('variant', ['mimgnet', 'celeba'])
def test_feature_trainers_run(tmp_path, variant):
    from cmvae_amd.trainer import Trainer
    d = _make(tmp_path, variant, '--per-class', '60', '--dim', '256')
    np.random.seed(10086); torch.manual_seed(10086)
    a = _args(d, str(tmp_path / 'run'), sample_size=5, batch_size=4, instances_per_class=60, train_iters=2, freq_iters=2)
    tr = Trainer(a, variant)
    assert tr.model.linear == (variant == 'mimgnet')
    out = tr.train()
    assert all(np.isfinite(m) for m, _ in out.values())


def test_eval_batch_size_is_independent_of_train_batch_size(tmp_path):
    """--eval-batch-size: larger prediction calls, episode count rounded up and trimmed (7 episodes in calls of 3)."""
    from cmvae_amd.trainer import Trainer
    d = _make(tmp_path, 'omniglot')
    np.random.seed(10086); torch.manual_seed(10086)
    tr = Trainer(_args(d, str(tmp_path / 'run'), eval_batch_size=3, eval_episodes=7), 'omniglot')
    mean, conf = tr.eval(1, test=True)
    assert 0.0 <= mean <= 1.0 and np.isfinite(conf)


def test_cli_entry_point_in_a_fresh_process(tmp_path):
    """`python omniglot/main.py ...` (the reference's command line): imports the package before torch, which must
    still bind the HIP library to PyTorch's runtime (cmvae_amd/_lib.py)."""
    d = _make(tmp_path, 'omniglot')
    cmd = [sys.executable, os.path.join(ROOT, 'omniglot', 'main.py'), '--data-dir', d, '--save-dir', str(tmp_path / 'cli'),
           '--train-iters', '4', '--freq-iters', '2', '--eval-episodes', '4', '--way', '5', '--sample-size', '20',
           '--batch-size', '2', '--query', '3', '--train-mc-sample-size', '4', '--test-mc-sample-size', '4',
           '--unsupervised-em-iters', '3', '--semisupervised-em-iters', '3']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    assert '5shot Final Test Accuracy' in r.stdout
    assert os.path.exists(str(tmp_path / 'cli' / 'train_log.jsonl'))


def test_prediction_pipeline_equals_sequential_calls(device):
    """Independent few-shot calls dealt to several HIP streams (cmvae_amd/pipeline.py, what Trainer.eval does), launched
    eagerly or replayed from HIP graphs: the labels of the sequential loop bit for bit, in submission order, for any
    number of streams (the noise of every call is drawn on the host thread in call order, outside the graphs)."""
    import torch
    from cmvae_amd.pipeline import PredictionPipeline
    from tests.helpers import build_pair, omniglot_batch
    K, shot, query, M, B = 5, 1, 3, 8, 2
    model, _ = build_pair('omniglot', K=K, M=M, iters=4, linear=False)
    model.to(device).eval()
    tr = K * shot
    episodes = [omniglot_batch(B, tr + K * query, seed=40 + i).to(device) for i in range(9)]
    y_tr = torch.arange(K, device=device).repeat_interleave(shot)[None].repeat(B, 1)

    def run(n_streams, graphs):
        torch.manual_seed(3)
        pipe = PredictionPipeline(model, n_streams, use_graphs=graphs)
        for X in episodes:
            pipe.submit(X[:, :tr].contiguous(), y_tr, X[:, tr:].contiguous())
        got = pipe.results()
        assert pipe.results() == [] and len(got) == len(episodes)
        return got

    torch.manual_seed(3)
    ref = [model.prediction(X[:, :tr].contiguous(), y_tr, X[:, tr:].contiguous()) for X in episodes]
    for n_streams, graphs in ((1, False), (3, False), (1, True), (3, True), (4, None)):
        for a, b in zip(run(n_streams, graphs), ref):
            assert torch.equal(a, b), (n_streams, graphs)
    # a second shape on the same pipeline captures a second graph
    pipe = PredictionPipeline(model, 2)
    X5 = omniglot_batch(B, 2 * K + K * query, seed=77).to(device)
    y5 = torch.arange(K, device=device).repeat_interleave(2)[None].repeat(B, 1)
    for _ in range(3):
        pipe.submit(episodes[0][:, :tr].contiguous(), y_tr, episodes[0][:, tr:].contiguous())
        pipe.submit(X5[:, :2 * K].contiguous(), y5, X5[:, 2 * K:].contiguous())
    out = pipe.results()
    assert tuple(out[0].shape) == (B, K * query) and tuple(out[1].shape) == (B, K * query)
    model.train()
    import pytest
    with pytest.raises(RuntimeError):
        PredictionPipeline(model, 2)


def test_graphed_step_equals_eager_step(device):
    """cmvae_amd/graph_step.py: the feature-model training step replayed from a HIP graph walks the eager step's
    trajectory — same noise realisation (drawn by the same generator call outside the graph), same kernels, and Adam's
    bias corrections from the device-side step count: parameters after 8 steps agree to fp32 rounding of the two Adam
    entry points, losses step by step."""
    import copy
    import numpy as np
    import torch
    from cmvae_amd import dist as cdist
    from cmvae_amd.graph_step import GraphedStep
    from cmvae_amd.linear import FlatAdam
    from cmvae_amd.model import FeatureGMVAE
    torch.manual_seed(4)
    B, S, M, K, Fd = 4, 5, 16, 5, 256
    m1 = FeatureGMVAE(input_shape=Fd, component_size=K, train_mc_sample_size=M, test_mc_sample_size=M,
                      unsupervised_em_iters=3, semisupervised_em_iters=3).to(device)
    m2 = copy.deepcopy(m1)
    g = torch.Generator().manual_seed(9)
    batches = [torch.relu(torch.randn(B, S, Fd, generator=g)).to(device) for _ in range(8)]
    rng = np.random.RandomState(3)
    seeds = [torch.from_numpy(np.stack([rng.choice(S * M, K, replace=False) for _ in range(B)])).to(device)
             for _ in range(8)]
    consts = (0.1, 0.1, 1e-3)

    f1 = cdist.FlatGrads(m1.parameters()); o1 = FlatAdam(f1, lr=1e-3, weight_decay=5e-5)
    torch.manual_seed(11)
    eager = []
    for X, idx in zip(batches, seeds):
        m1.train(); f1.zero()
        _, rec, kl, ha, h = m1(X, *consts, init_idx=idx)
        (rec + kl + ha).backward(); f1.allreduce(); o1.step()
        eager.append([float(t.detach()) for t in (rec, kl, ha, h)])

    f2 = cdist.FlatGrads(m2.parameters()); o2 = FlatAdam(f2, lr=1e-3, weight_decay=5e-5)
    stepper = GraphedStep(m2, f2, o2)
    torch.manual_seed(11)
    replayed = []
    for X, idx in zip(batches, seeds):
        replayed.append([float(t) for t in stepper(X, idx, *consts)])
    assert stepper.calls == 8 and len(stepper.graphs) == 1
    for a, b in zip(eager, replayed):
        for x, y in zip(a, b):
            assert abs(x - y) <= 2e-5 * max(abs(x), 1e-3), (a, b)
    for p1, p2 in zip(m1.parameters(), m2.parameters()):
        assert float((p1 - p2).detach().abs().max()) <= 2e-5 * float(p1.detach().abs().max()) + 1e-7
    assert o2.state_dict()['step'] == 8 and o1.state_dict()['step'] == 8
    # other loss constants (--dual-ascent moves them once per epoch): the graph of that shape is replaced, not kept
    # — one graph and one pool of activations per shape however often the constants change — and the new values act
    from cmvae_amd.trainer import dual_ascent_update
    h_lambda, h_c, h_old = 0.1, 0.1, float('inf')
    for _ in range(3):
        out = stepper(batches[0], seeds[0], h_lambda, h_c, 1e-3)
        h_ = float(out[3])
        assert len(stepper.graphs) == 1
        h_lambda, h_c, h_old = dual_ascent_update(h_, h_lambda, h_c, h_old)
    assert h_lambda > 0.1                                    # the schedule did move the constants
    # the last replay used the constants it was called with (A as it was before the step's Adam update)
    A = m2.causal.get_w().detach().clone()
    out = stepper(batches[0], seeds[0], h_lambda, h_c, 1e-3)
    h_now = float(out[3])
    assert abs(float(out[2]) - (h_lambda * h_now + 0.5 * h_c * h_now ** 2 + 1e-3 * float(A.abs().sum()))) \
        <= 1e-4 * abs(float(out[2]))
    assert len(stepper.graphs) == 1
