"""Episode-sharded data parallelism on the real kernels: two ranks share the one GPU of the test box and talk over
gloo (RCCL refuses two ranks on one device; the collectives' call sites are the same).  A step on the sharded
meta-batch must equal the single-process step on the whole meta-batch: BatchNorm statistics span all episodes
(collective C2 inside the HIP encoder / decoder pipelines; whole-meta-batch BatchNorm as in omniglot/module.py:15-31,
51-59), gradients are averaged (C1), EM seeds come from the shared numpy stream.

Two shapes: a tiny one (48 decoder images per rank: the LDS-staged weight gradient) and the shape the 8-GPU job runs
per rank — 4 episodes x 300 items x 32 MC samples, 20-way, 10 EM iterations — against one process on all 8 episodes:
streaming weight gradient, split sixteen-tap data gradient with the BatchNorm-backward sums in its epilogue feeding
`dist.global_sums`, and the two-launch BatchNorm finalize that replaces `cmvae_bn_finalize_part` when world > 1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
SHAPES = {'tiny': dict(B=2, S=12, K=3, M=4, iters=3), 'o20_per_rank': dict(B=8, S=300, K=20, M=32, iters=10)}


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(device, sh):
    from cmvae_amd.model import OmniglotGMVAE
    torch.manual_seed(4)
    return OmniglotGMVAE(component_size=sh['K'], train_mc_sample_size=sh['M'], test_mc_sample_size=sh['M'],
                         unsupervised_em_iters=sh['iters'], semisupervised_em_iters=sh['iters']).to(device).train()


def _inputs(sh):
    g = torch.Generator().manual_seed(5)
    X = (torch.rand(sh['B'], sh['S'], 1, 28, 28, generator=g) < 0.15).float()
    eps = torch.randn(sh['B'], sh['S'], sh['M'], 64, generator=g)
    return X, eps


def _bn_stats(model):
    return torch.cat([b.detach().reshape(-1).float().cpu() for n, b in model.named_buffers()
                      if n.endswith('running_mean') or n.endswith('running_var')])


def _step(model, flat, opt, X, eps, idx, device):
    flat.zero()
    _, rec, kl, h_aug, _ = model(X.to(device), 0.1, 0.1, 1e-4, eps=eps.to(device), init_idx=torch.from_numpy(idx).to(device))
    (rec + kl + h_aug).backward()
    flat.allreduce()
    grads = flat.flat.clone()
    opt.step()
    return torch.stack([rec.detach(), kl.detach(), h_aug.detach()]).cpu(), grads.cpu()


def _worker(rank, world, port, out, shape, backend='gloo'):
    # default hardware queues: the exchanges of GPU tensors over gloo are staged on the calling thread
    # (cmvae_amd/dist.py: _all_reduce; DESIGN.md §5 has the story of the stall this replaced)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0')
    if world == 1:
        os.environ['CMVAE_DIST_FORCE'] = '1'           # one rank walks the several-ranks code (read at import)
    sh = SHAPES[shape]
    from cmvae_amd import dist as cdist
    from cmvae_amd.linear import FlatAdam
    cdist.init_from_env(backend)
    assert cdist.exchanging()
    device = torch.device('cuda:0')
    model = _build(device, sh)
    flat = cdist.FlatGrads(model.parameters()).overlap_tail(model.tail_param_count())    # decoder + proj bucket
    opt = FlatAdam(flat, lr=1e-3, weight_decay=5e-5)
    X, eps = _inputs(sh)
    sl = cdist.shard_slice(sh['B'])
    np.random.seed(21)
    idx = cdist.draw_em_init(sh['S'] * sh['M'], sh['K'], sh['B'])
    losses, grads = _step(model, flat, opt, X[sl], eps[sl], idx, device)
    assert flat._tail_off < flat.flat.numel()          # the tail bucket exists; _step's allreduce consumed it
    out[rank] = dict(losses=losses, grads=grads, w=model.proj[0].weight.detach().cpu().clone(), bn=_bn_stats(model))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('shape', ['tiny', 'o20_per_rank'])
def test_two_ranks_on_hip_kernels_equal_one_process(device, shape):
    sh = SHAPES[shape]
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, shape), nprocs=world, join=True)
    from cmvae_amd import dist as cdist
    from cmvae_amd.linear import FlatAdam
    model = _build(device, sh)
    flat = cdist.FlatGrads(model.parameters())
    opt = FlatAdam(flat, lr=1e-3, weight_decay=5e-5)
    X, eps = _inputs(sh)
    np.random.seed(21)
    idx = cdist._draw_em_init_numpy(sh['S'] * sh['M'], sh['K'], sh['B'])     # the reference's own numpy calls
    losses, grads = _step(model, flat, opt, X, eps, idx, device)
    # every loss term is a mean over equal-sized shards (h_augment is replicated)
    mean_losses = (out[0]['losses'] + out[1]['losses']) / 2
    assert torch.allclose(mean_losses, losses, rtol=2e-5, atol=1e-6), (mean_losses, losses)
    scale = float(grads.abs().max())
    # Adam's first step is lr * g / (|g| + eps): where |g| is near eps = 1e-8 the update is rounding noise of the
    # gradient, so the weights are compared where the gradient is well above that
    gw = flat.views[[id(p) for p in flat.params].index(id(model.proj[0].weight))].detach().cpu()
    solid = gw.abs() > 1e-5 * scale
    assert float(solid.float().mean()) > 0.5
    bn = _bn_stats(model)
    for r in range(world):
        err = float((out[r]['grads'] - grads).abs().max())
        print('  %s rank %d: max |grad - single| %.2e of scale %.2e; BatchNorm running statistics %.2e'
              % (shape, r, err, scale, float((out[r]['bn'] - bn).abs().max())))
        assert err < 2e-4 * scale
        dw = (out[r]['w'] - model.proj[0].weight.detach().cpu()).abs()
        assert float(dw[solid].max()) < 1e-5 and float(dw.max()) < 2.1e-3      # never more than two steps of lr
        assert torch.allclose(out[r]['bn'], bn, atol=1e-6, rtol=1e-6)           # all 14 running_mean / running_var
    assert torch.equal(out[0]['grads'], out[1]['grads'])
    assert torch.equal(out[0]['w'], out[1]['w'])                            # the ranks stay in lock step
    assert torch.equal(out[0]['bn'], out[1]['bn'])


# ---------------------------------------------------------------------------------------------------------------------
# the feature models' step replayed from a HIP graph with several ranks (cmvae_amd/graph_step.py): graph up to the
# gradient pack, exchange and Adam after the replay
FB, FS, FM, FK, FD, FSTEPS = 4, 5, 16, 5, 256, 6


def _feat_model(device):
    from cmvae_amd.model import FeatureGMVAE
    torch.manual_seed(8)
    return FeatureGMVAE(input_shape=FD, component_size=FK, train_mc_sample_size=FM, test_mc_sample_size=FM,
                        unsupervised_em_iters=3, semisupervised_em_iters=3).to(device)


def _feat_batches():
    g = torch.Generator().manual_seed(12)
    return [torch.relu(torch.randn(FB, FS, FD, generator=g)) for _ in range(FSTEPS)]


def _graph_worker(rank, world, port, out, backend='gloo'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0')
    if world == 1:
        os.environ['CMVAE_DIST_FORCE'] = '1'
    from cmvae_amd import dist as cdist
    from cmvae_amd.graph_step import GraphedStep
    from cmvae_amd.linear import FlatAdam
    cdist.init_from_env(backend)
    assert cdist.exchanging()
    device = torch.device('cuda:0')
    model = _feat_model(device)
    flat = cdist.FlatGrads(model.parameters()).overlap_tail(model.tail_param_count())
    opt = FlatAdam(flat, lr=1e-3, weight_decay=5e-5)
    stepper = GraphedStep(model, flat, opt)
    sl = cdist.shard_slice(FB)
    np.random.seed(31)
    torch.manual_seed(32)
    losses = []
    for X in _feat_batches():
        idx = torch.from_numpy(cdist.draw_em_init(FS * FM, FK, FB)).to(device)
        losses.append([float(t) for t in stepper(X[sl].to(device).contiguous(), idx, 0.1, 0.1, 1e-3)])
    assert len(stepper.graphs) == 1 and stepper.calls == FSTEPS
    out[rank] = dict(losses=torch.tensor(losses), w=model.proj[0].weight.detach().cpu().clone())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_graphed_feature_step_equals_one_process(device):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_graph_worker, args=(world, port, out), nprocs=world, join=True)
    from cmvae_amd import dist as cdist
    from cmvae_amd.linear import FlatAdam
    model = _feat_model(device)
    w0 = model.proj[0].weight.detach().cpu().clone()
    flat = cdist.FlatGrads(model.parameters())
    opt = FlatAdam(flat, lr=1e-3, weight_decay=5e-5)
    np.random.seed(31)
    torch.manual_seed(32)
    losses = []
    for X in _feat_batches():
        idx = torch.from_numpy(cdist._draw_em_init_numpy(FS * FM, FK, FB)).to(device)
        model.train()
        flat.zero()
        _, rec, kl, ha, h = model(X.to(device), 0.1, 0.1, 1e-3, init_idx=idx)
        (rec + kl + ha).backward()
        flat.allreduce()
        opt.step()
        losses.append([float(t.detach()) for t in (rec, kl, ha, h)])
    losses = torch.tensor(losses)
    mean_losses = (out[0]['losses'] + out[1]['losses']) / 2        # means over equal shards; h terms replicated
    print('  graphed 2-rank vs eager 1-process losses, max rel %.2e'
          % float(((mean_losses - losses).abs() / losses.abs().clamp_min(1e-6)).max()))
    assert torch.allclose(mean_losses, losses, rtol=1e-4, atol=1e-6), (mean_losses, losses)
    assert torch.equal(out[0]['w'], out[1]['w'])                    # the ranks stay in lock step
    w = model.proj[0].weight.detach().cpu()
    moved = (w - w0).abs() > 2e-3                                    # weights that took clear steps in one direction
    assert float(moved.float().mean()) > 0.2
    # Adam divides by sqrt(v): an element whose gradient is rounding noise in some step may differ by a fraction of
    # that step's lr; nearly all elements agree far below it, none differs by a whole step
    d = (out[0]['w'] - w)[moved].abs()
    print('  weights after %d steps: median |d| %.1e, 99th percentile %.1e, max %.1e'
          % (FSTEPS, float(d.median()), float(d.quantile(0.99)), float(d.max())))
    assert float(d.quantile(0.99)) < 1e-4 and float(d.max()) < 1e-3


# ---------------------------------------------------------------------------------------------------------------------
# The production backend on the one GPU of the test box: RCCL ("nccl") refuses two ranks on a device, but a process group
# of ONE rank with CMVAE_DIST_FORCE=1 walks the whole several-ranks code over it — BatchNorm-statistics all-reduces from
# the forward and from the autograd thread, the tail bucket all-reduced asynchronously from a gradient hook, the head
# exchange behind it, the sharded noise draw, and HIP-graph capture next to a live RCCL communicator and its watchdog.
# An all-reduce over one rank is the identity, so the results must be the plain single-process ones.
def _single_process_step(device, sh):
    from cmvae_amd import dist as cdist
    from cmvae_amd.linear import FlatAdam
    model = _build(device, sh)
    flat = cdist.FlatGrads(model.parameters())
    opt = FlatAdam(flat, lr=1e-3, weight_decay=5e-5)
    X, eps = _inputs(sh)
    np.random.seed(21)
    idx = cdist._draw_em_init_numpy(sh['S'] * sh['M'], sh['K'], sh['B'])
    losses, grads = _step(model, flat, opt, X, eps, idx, device)
    return losses, grads, model.proj[0].weight.detach().cpu(), _bn_stats(model)


def test_one_rank_over_rccl_walks_the_multi_rank_path(device):
    sh = SHAPES['tiny']
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(1, _free_port(), out, 'tiny', 'nccl'), nprocs=1, join=True)
    losses, grads, w, bn = _single_process_step(device, sh)
    assert torch.allclose(out[0]['losses'], losses, rtol=2e-5, atol=1e-6), (out[0]['losses'], losses)
    scale = float(grads.abs().max())
    assert float((out[0]['grads'] - grads).abs().max()) < 2e-4 * scale
    assert torch.allclose(out[0]['bn'], bn, atol=1e-6, rtol=1e-6)


def test_one_rank_over_rccl_graphed_feature_step(device):
    """The feature models' HIP-graph step with a live RCCL process group: capture, replay, eager all-reduce + Adam."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_graph_worker, args=(1, _free_port(), out, 'nccl'), nprocs=1, join=True)
    assert out[0]['losses'].shape == (FSTEPS, 4) and bool(torch.isfinite(out[0]['losses']).all())
