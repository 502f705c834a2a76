"""CPU, world_size 2 over gloo: the episode-sharded data-parallel host logic (cmvae_amd/dist.py) gives the
same numbers as one process on the whole meta-batch — BatchNorm statistics all-reduce (C2), flat gradient
all-reduce (C1), EM-seed slicing from the shared numpy stream."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class EpisodeBatchNorm2d(nn.BatchNorm2d):
    def forward(self, x):
        from tests.sync_bn_reference import sync_batch_norm
        return sync_batch_norm(x, self) if self.training else super().forward(x)


def _net():
    torch.manual_seed(3)
    return nn.Sequential(nn.Conv2d(1, 4, 3, 1, 1), EpisodeBatchNorm2d(4), nn.ReLU(), nn.Conv2d(4, 2, 3, 1, 1),
                         EpisodeBatchNorm2d(2))


def _data():
    g = torch.Generator().manual_seed(9)
    return torch.randn(8, 1, 6, 6, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cmvae_amd import dist as cdist
    r, w = cdist.init_from_env('gloo')
    assert (r, w) == (rank, world)
    net = _net().train()
    flat = cdist.FlatGrads(net.parameters()).overlap_tail(4)      # second conv + second BatchNorm: the tail bucket
    x = _data()[cdist.shard_slice(8)]
    flat.zero()
    y = net(x)
    loss = (y ** 2).mean()                       # a mean over an equal-sized shard
    loss.backward()
    assert flat._tail_done and flat._tail_off == 4 * 9 + 4 + 4 + 4      # the tail went out during backward
    flat.allreduce()
    # a second step whose tail gets no gradient for one parameter falls back to one exchange of everything
    flat.zero()
    (net[0](x) ** 2).mean().backward()
    assert not flat._tail_done
    flat.allreduce()
    g_partial = flat.flat.clone()
    flat.zero()
    loss = (net(x) ** 2).mean()
    loss.backward()
    flat.allreduce()
    grad_full, rm_full, rv_full = flat.flat.clone(), net[1].running_mean.clone(), net[4].running_var.clone()
    # a second backward between two zero() calls would add into the tail bucket that already went out: refused
    flat.zero()
    (net(x) ** 2).mean().backward()
    try:
        (net(x) ** 2).mean().backward()
        second_backward_refused = False
    except RuntimeError as e:
        second_backward_refused = 'backward ran twice' in str(e)
    flat.allreduce()                              # consumes the outstanding tail exchange on both ranks
    assert second_backward_refused
    np.random.seed(77)
    idx = cdist.draw_em_init(50, 3, 8)
    # reparametrisation noise: same generator state on both ranks -> disjoint slices of the single-process draw
    torch.manual_seed(5)
    eps = cdist.draw_eps(8, 3, 2, 4, 'cpu')
    # evaluation: 3 calls of a global batch of 4 episodes, 2 per rank; value = global episode number
    mine = torch.tensor([4.0 * t + 2 * rank + j for t in range(3) for j in range(2)])
    gathered = cdist.gather_episode_values(mine, 2)
    assert gathered.tolist() == [float(i) for i in range(12)], gathered
    out[rank] = dict(y=y.detach(), grad=grad_full, rm=rm_full,
                     rv=rv_full, idx=idx, after=np.random.rand(), eps=eps.clone(),
                     g_partial=g_partial)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_step_equals_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    # single process on the whole meta-batch
    from cmvae_amd import dist as cdist
    net = _net().train()
    flat = cdist.FlatGrads(net.parameters())
    flat.zero()
    (net[0](_data()) ** 2).mean().backward()
    flat.pack()
    g_partial = flat.flat.clone()
    for _ in range(2):                                 # the workers' first two full training-mode forwards
        flat.zero()
        y = net(_data())
        (y ** 2).mean().backward()
        flat.pack()
    got_y = torch.cat([out[0]['y'], out[1]['y']], 0)
    assert torch.allclose(got_y, y.detach(), atol=1e-5)
    for r in range(world):
        assert torch.allclose(out[r]['grad'], flat.flat, atol=1e-5), (out[r]['grad'] - flat.flat).abs().max()
        assert torch.allclose(out[r]['rm'], net[1].running_mean, atol=1e-6)
        assert torch.allclose(out[r]['rv'], net[4].running_var, atol=1e-6)
    for r in range(world):
        assert torch.allclose(out[r]['g_partial'], g_partial, atol=1e-5)
    torch.manual_seed(5)
    assert torch.equal(torch.cat([out[0]['eps'], out[1]['eps']], 0), torch.randn(8, 3, 2, 4))
    # EM seeds: every rank drew the global stream and kept its own slice
    np.random.seed(77)
    ref = np.stack([np.random.choice(50, 3, replace=False) for _ in range(8)], 0)
    after = np.random.rand()
    assert np.array_equal(np.concatenate([out[0]['idx'], out[1]['idx']], 0), ref)
    assert out[0]['after'] == after and out[1]['after'] == after


def test_shard_slice_rejects_uneven_batches():
    from cmvae_amd import dist as cdist
    import pytest
    with pytest.raises(ValueError):
        cdist.shard_slice(7, 0, 2)
    assert cdist.shard_slice(8, 1, 4) == slice(2, 4)
