"""Episode-sharded data parallelism on the real kernels: two ranks share the one GPU of the test box and talk over
gloo (RCCL refuses two ranks on one device; the collectives' call sites are the same).  A step on the sharded
meta-batch must equal the single-process step on the whole meta-batch: BatchNorm statistics span all episodes
(collective C2 inside the HIP encoder / decoder pipelines), gradients are averaged (C1), EM seeds come from the shared
numpy stream."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
B, S, K, M = 2, 12, 3, 4


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(device):
    from cmvae_amd.model import OmniglotGMVAE
    torch.manual_seed(4)
    return OmniglotGMVAE(component_size=K, train_mc_sample_size=M, test_mc_sample_size=M, unsupervised_em_iters=3,
                         semisupervised_em_iters=3).to(device).train()


def _inputs():
    g = torch.Generator().manual_seed(5)
    X = (torch.rand(B, S, 1, 28, 28, generator=g) < 0.15).float()
    eps = torch.randn(B, S, M, 64, generator=g)
    return X, eps


def _step(model, flat, opt, X, eps, idx, device):
    flat.zero()
    _, rec, kl, h_aug, _ = model(X.to(device), 0.1, 0.1, 1e-4, eps=eps.to(device), init_idx=torch.from_numpy(idx).to(device))
    (rec + kl + h_aug).backward()
    flat.allreduce()
    grads = flat.flat.clone()
    opt.step()
    return torch.stack([rec.detach(), kl.detach(), h_aug.detach()]).cpu(), grads.cpu()


def _worker(rank, world, port, out):
    # (one hardware queue per process: two processes with several queues each on ONE GPU can dead-lock in a collective
    # that waits for the other rank's kernels — seen with bench.py's full-size rehearsal; set before the
    # child's first HIP call)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', GPU_MAX_HW_QUEUES='1')
    from cmvae_amd import dist as cdist
    from cmvae_amd.linear import FlatAdam
    cdist.init_from_env('gloo')
    device = torch.device('cuda:0')
    model = _build(device)
    flat = cdist.FlatGrads(model.parameters()).overlap_tail(model.tail_param_count())    # decoder + proj bucket
    opt = FlatAdam(flat, lr=1e-3, weight_decay=5e-5)
    X, eps = _inputs()
    sl = cdist.shard_slice(B)
    np.random.seed(21)
    idx = cdist.draw_em_init(S * M, K, B)
    losses, grads = _step(model, flat, opt, X[sl], eps[sl], idx, device)
    assert flat._tail_off < flat.flat.numel()          # the tail bucket exists; _step's allreduce consumed it
    out[rank] = dict(losses=losses, grads=grads, w=model.proj[0].weight.detach().cpu().clone(),
                     rm=model.decoder.decoder[1].running_mean.detach().cpu().clone())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_on_hip_kernels_equal_one_process(device):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    from cmvae_amd import dist as cdist
    from cmvae_amd.linear import FlatAdam
    model = _build(device)
    flat = cdist.FlatGrads(model.parameters())
    opt = FlatAdam(flat, lr=1e-3, weight_decay=5e-5)
    X, eps = _inputs()
    np.random.seed(21)
    idx = np.stack([np.random.choice(S * M, K, replace=False) for _ in range(B)], 0).astype(np.int64)
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
    for r in range(world):
        assert float((out[r]['grads'] - grads).abs().max()) < 2e-4 * scale
        dw = (out[r]['w'] - model.proj[0].weight.detach().cpu()).abs()
        assert float(dw[solid].max()) < 1e-5 and float(dw.max()) < 2.1e-3      # never more than two steps of lr
        assert torch.allclose(out[r]['rm'], model.decoder.decoder[1].running_mean.detach().cpu(), atol=1e-6)
    assert torch.equal(out[0]['grads'], out[1]['grads'])
    assert torch.equal(out[0]['w'], out[1]['w'])                            # the ranks stay in lock step
