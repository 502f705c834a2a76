"""Episode-sharded data parallelism: one process per GPU, RCCL through
``torch.distributed`` (backend "nccl" is RCCL on ROCm; "gloo" on CPU for the tests).

The reference has no distributed code; this module is new work.  Two exchange steps exist:
  C1  one all-reduce of a flat fp32 gradient buffer per step (FlatGrads.allreduce);
  C2  per-channel BatchNorm statistics (global_sums, called from the HIP encoder / decoder pipelines between
      the GEMM-epilogue partial sums and cmvae_bn_finalize / cmvae_bn_bwd_finalize), Omniglot training only,
      so that normalisation spans the whole meta-batch exactly as in the single-process reference.
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


def draw_eps(b_global, s, m, d, device, r=None, ws=None, dtype=torch.float32, generator=None):
    """Reparametrisation noise of the global meta-batch, drawn exactly as a single process would — one
    ``randn`` of shape [B_global, S, M, D] from the device generator (omniglot/model.py:121) — on every rank,
    which keeps its own slice.  All ranks are seeded alike (main.py seeds 10086 everywhere), so the shards get
    independent noise and the union equals the single-process draw.  Costs one extra 78 MB write per step at 8
    ranks of the Omniglot 20-way shape."""
    eps = torch.randn(b_global, s, m, d, device=device, dtype=dtype, generator=generator)
    return eps[shard_slice(b_global, r, ws)]


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


def global_sums(sums):
    """Collective C2: per-channel sums of a BatchNorm layer ({sum x, sum x^2} in the forward pass, {sum dy,
    sum dy*xhat} in the backward pass; 128 floats) summed over the ranks.  The one implementation of the exchange:
    the HIP pipelines (cmvae_amd/encoder.py, decoder.py) and the CPU restatement the gloo tests run
    (tests/sync_bn_reference.py) both call it."""
    if world_size() > 1:
        g = sums.clone()
        tdist.all_reduce(g, op=tdist.ReduceOp.SUM)
        return g
    return sums


class FlatGrads:
    """All parameter gradients in one flat fp32 buffer (4.56 MB for Omniglot): the step's gradient exchange is one
    all-reduce per bucket and the optimizer a single kernel over it.

    Between `zero()` and `pack()` the parameters carry no `.grad`, so autograd hands every gradient over as it
    is produced (no per-parameter accumulation kernel); `pack()` — called by `allreduce()` — gathers them into
    the flat buffer with one multi-tensor copy and points every `.grad` at its slice.

    `overlap_tail(n)`: the last n parameters (module order: ..., proj, decoder — the part of the model whose
    backward runs first) form a second bucket whose all-reduce is issued from a gradient hook as soon as its last
    gradient exists, so it travels over xGMI while the rest of the backward pass (causal layer, set attention,
    encoder) still runs; `allreduce()` then exchanges the head bucket and waits for the tail."""

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
        self._tail_start = len(self.params)      # index of the first parameter of the tail bucket
        self._tail_off = self.flat.numel()
        self._tail_pending = 0
        self._tail_work = None
        self._tail_done = False
        self._hooks = []

    def overlap_tail(self, n_tail_params):
        """Make the last `n_tail_params` parameters a bucket that is all-reduced during the backward pass."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        n_tail_params = max(0, min(int(n_tail_params), len(self.params)))
        self._tail_start = len(self.params) - n_tail_params
        self._tail_off = sum(p.numel() for p in self.params[:self._tail_start]) if n_tail_params else self.flat.numel()
        if n_tail_params and world_size() > 1:
            for p in self.params[self._tail_start:]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._tail_hook))
        return self

    def _tail_hook(self, _param):
        self._tail_pending -= 1
        if self._tail_pending == 0 and not self._tail_done:
            self._pack(self._tail_start, len(self.params))
            self._tail_work = tdist.all_reduce(self.flat[self._tail_off:], op=tdist.ReduceOp.SUM, async_op=True)
            self._tail_done = True

    def zero(self):
        for p in self.params:
            p.grad = None
        self._tail_pending = len(self.params) - self._tail_start
        self._tail_work, self._tail_done = None, False

    def _pack(self, lo, hi):
        src, dst = [], []
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)

    def pack(self):
        self._pack(0, len(self.params))

    def allreduce(self):
        """Gather the gradients, then mean over ranks (every loss term is a mean over equal-sized shards)."""
        ws = world_size()
        if ws == 1:
            self.pack()
            return
        if self._tail_done:
            self._pack(0, self._tail_start)
            tdist.all_reduce(self.flat[:self._tail_off], op=tdist.ReduceOp.SUM)
            self._tail_work.wait()               # stream-ordered on RCCL: the host does not block
            self._tail_work = None
        else:
            # no hook fired (overlap off, or a parameter of the tail got no gradient): one exchange of everything
            self.pack()
            tdist.all_reduce(self.flat, op=tdist.ReduceOp.SUM)
        self.flat.div_(ws)
