"""Episode-sharded data parallelism: one process per GPU, RCCL through
``torch.distributed`` (backend "nccl" is RCCL on ROCm; "gloo" on CPU for the tests).

The reference has no distributed code; this module is new work.  Two exchange steps exist:
  C1  one all-reduce of a flat fp32 gradient buffer per step (FlatGrads.allreduce);
  C2  per-channel BatchNorm statistics (sync_batch_norm), Omniglot training only, so that
      normalisation spans the whole meta-batch exactly as in the single-process reference.
Everything else on the path is per-episode and needs no collective.
"""
import os

import numpy as np
import torch
import torch.distributed as tdist


def is_dist():
    return tdist.is_available() and tdist.is_initialized()


def world_size():
    return tdist.get_world_size() if is_dist() else 1


def rank():
    return tdist.get_rank() if is_dist() else 0


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws <= 1 or is_dist():
        return rank(), world_size()
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    tdist.init_process_group(backend=backend)
    return rank(), world_size()


def shard_slice(n_global, r=None, ws=None):
    """Contiguous slice of the global episode axis owned by rank r."""
    r = rank() if r is None else r
    ws = world_size() if ws is None else ws
    if n_global % ws != 0:
        raise ValueError('global meta-batch %d is not divisible by world size %d' % (n_global, ws))
    per = n_global // ws
    return slice(r * per, (r + 1) * per)


def draw_em_init(n_samples, k, b_global, r=None, ws=None):
    """EM seeds for the global meta-batch, drawn exactly as the reference does — one
    ``np.random.choice(N, K, replace=False)`` per episode in episode order from numpy's global
    stream (omniglot/model.py:351-356) — on every rank, which then keeps its own slice.  The
    numpy stream therefore stays identical to a single-process run."""
    idx = np.stack([np.random.choice(n_samples, k, replace=False) for _ in range(b_global)], 0)
    return idx[shard_slice(b_global, r, ws)].astype(np.int64)


class SeedUploader:
    """EM seeds host -> device without stalling the host: a ring of pinned buffers and asynchronous copies.
    (A plain `.to(device)` from pageable memory blocks the host until everything queued before it has run, which
    drains the launch queue once per step.)"""

    def __init__(self, shape, device, depth=4):
        self.device = torch.device(device)
        self.pinned = [torch.empty(shape, dtype=torch.int64).pin_memory() for _ in range(depth)]
        self.events = [None] * depth
        self.i = 0

    def upload(self, idx_numpy):
        k = self.i
        self.i = (k + 1) % len(self.pinned)
        if self.events[k] is not None:
            self.events[k].synchronize()                 # the copy that last used this buffer has run
        self.pinned[k].copy_(torch.from_numpy(idx_numpy))
        out = self.pinned[k].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return out


def gather_episode_values(values, local_batch):
    """Per-episode values [n_calls * local_batch] of every rank -> one vector in global episode order
    (call t holds the episodes t*B_g ... (t+1)*B_g - 1, rank r owns the r-th contiguous slice of them).
    The only collective of the evaluation loop."""
    if world_size() == 1:
        return values
    parts = [torch.empty_like(values) for _ in range(world_size())]
    tdist.all_gather(parts, values.contiguous())
    return torch.stack([p.view(-1, local_batch) for p in parts], 1).reshape(-1)


class FlatGrads:
    """All parameter gradients in one flat fp32 buffer (4.56 MB for Omniglot): the step's gradient exchange is a
    single all-reduce and the optimizer a single kernel over it.

    Between `zero()` and `pack()` the parameters carry no `.grad`, so autograd hands every gradient over as it
    is produced (no per-parameter accumulation kernel); `pack()` — called by `allreduce()` — gathers them into
    the flat buffer with one multi-tensor copy and points every `.grad` at its slice."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros((n + 3) // 4 * 4, dtype=ref.dtype, device=ref.device)     # padded for float4 kernels
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
            p.grad = self.views[-1]

    def zero(self):
        for p in self.params:
            p.grad = None

    def pack(self):
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)

    def allreduce(self):
        """Gather the gradients, then mean over ranks (every loss term is a mean over equal-sized shards)."""
        self.pack()
        ws = world_size()
        if ws > 1:
            tdist.all_reduce(self.flat, op=tdist.ReduceOp.SUM)
            self.flat.div_(ws)


class _SyncBN(torch.autograd.Function):
    """The C2 exchange step written out on torch tensors: what is summed, what is all-reduced and how the
    backward uses the global sums.  The world_size-2 gloo tests run it on the CPU; on the GPU the same
    exchange is `cmvae_amd.decoder._global_sums` between the HIP kernels (GEMM-epilogue partial sums ->
    all-reduce -> cmvae_bn_finalize / cmvae_bn_bwd_finalize)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        C = x.shape[1]
        red = [0, 2, 3]
        stats = torch.empty(2 * C + 1, dtype=x.dtype, device=x.device)
        stats[:C] = x.sum(red)
        stats[C:2 * C] = (x * x).sum(red)
        stats[2 * C] = x.numel() // C
        if world_size() > 1:
            tdist.all_reduce(stats, op=tdist.ReduceOp.SUM)
        n = stats[2 * C]
        mean = stats[:C] / n
        var = stats[C:2 * C] / n - mean * mean
        invstd = torch.rsqrt(var + eps)
        xhat = (x - mean.view(1, C, 1, 1)) * invstd.view(1, C, 1, 1)
        ctx.save_for_backward(xhat, weight, invstd, n)
        ctx.mark_non_differentiable(mean, var, n)
        return xhat * weight.view(1, C, 1, 1) + bias.view(1, C, 1, 1), mean, var, n

    @staticmethod
    def backward(ctx, dy, _gm, _gv, _gn):
        xhat, weight, invstd, n = ctx.saved_tensors
        C = dy.shape[1]
        red = [0, 2, 3]
        local = torch.empty(2 * C, dtype=dy.dtype, device=dy.device)
        local[:C] = dy.sum(red)
        local[C:] = (dy * xhat).sum(red)
        dbias, dweight = local[:C].clone(), local[C:].clone()
        if world_size() > 1:
            tdist.all_reduce(local, op=tdist.ReduceOp.SUM)
        mdy = (local[:C] / n).view(1, C, 1, 1)
        mdyx = (local[C:] / n).view(1, C, 1, 1)
        dx = (weight * invstd).view(1, C, 1, 1) * (dy - mdy - xhat * mdyx)
        return dx, dweight, dbias, None


def sync_batch_norm(x, bn):
    """Training-mode BatchNorm2d with statistics all-reduced over ranks; running statistics are
    updated like nn.BatchNorm2d (momentum, unbiased variance over the global count)."""
    y, mean, var, n = _SyncBN.apply(x, bn.weight, bn.bias, bn.eps)
    if bn.track_running_stats:
        with torch.no_grad():
            m = bn.momentum
            bn.num_batches_tracked += 1
            if m is None:
                m = 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1 - m).add_(var * (n / (n - 1)), alpha=m)
    return y
