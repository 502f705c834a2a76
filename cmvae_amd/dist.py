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
import threading
import time

import numpy as np
import torch
import torch.distributed as tdist

# CMVAE_DIST_TRACE=<dir>: every collective of this module is logged on issue and on completion (rank, host thread,
# running number, what, elements) to <dir>/rank<r>.log — how a stalled multi-rank run says which exchange each rank
# is in and which ones are still outstanding
_TRACE_DIR = os.environ.get('CMVAE_DIST_TRACE', '')
_trace_lock = threading.Lock()
_trace_seq = [0]
_trace_file = [None]


def trace(msg):
    if not _TRACE_DIR:
        return
    with _trace_lock:
        if _trace_file[0] is None:
            os.makedirs(_TRACE_DIR, exist_ok=True)
            _trace_file[0] = open(os.path.join(_TRACE_DIR, 'rank%d.log' % rank()), 'a')
        _trace_file[0].write('%.6f [%s] %s\n' % (time.time(), threading.current_thread().name, msg))
        _trace_file[0].flush()


class _Done:
    """Work handle of an exchange that already completed on the calling thread."""

    def wait(self):
        return True


def _all_reduce(t, op, async_op):
    if t.is_cuda and tdist.get_backend() == 'gloo':
        # gloo has no GPU transport: c10d would stage the tensor through pinned memory on a pool stream and block one
        # of its worker threads in hipStreamSynchronize.  Staging on the calling thread instead keeps it to ONE host
        # thread per process that ever blocks on the GPU (tests and the one-GPU rehearsal of bench.py).
        # The production backend is RCCL ("nccl"), which takes the tensor as it is.
        h = t.detach().cpu()
        tdist.all_reduce(h, op=op)
        t.copy_(h)
        return _Done() if async_op else None
    return tdist.all_reduce(t, op=op, async_op=async_op)


def all_reduce(t, what, op=None, async_op=False):
    """The one call site of torch.distributed.all_reduce on the path (traceable)."""
    op = tdist.ReduceOp.SUM if op is None else op
    if not _TRACE_DIR:
        return _all_reduce(t, op, async_op)
    with _trace_lock:
        _trace_seq[0] += 1
        n = _trace_seq[0]
    trace('issue #%d %s numel=%d async=%s' % (n, what, t.numel(), async_op))
    w = _all_reduce(t, op, async_op)
    trace('%s #%d %s' % ('enqueued' if async_op else 'done', n, what))
    return w


def is_dist():
    return tdist.is_available() and tdist.is_initialized()


def world_size():
    return tdist.get_world_size() if is_dist() else 1


def rank():
    return tdist.get_rank() if is_dist() else 0


# CMVAE_DIST_FORCE=1: a ONE-rank job walks the several-ranks code (process group, BatchNorm-statistics and gradient
# exchanges, the hook-issued tail bucket, the sharded noise draw) — how the RCCL path is exercised on a one-GPU box:
# an all-reduce over one rank is the identity, but the process group, its streams and its interplay with hooks and HIP
# graph capture are the real ones (tests/test_gpu_dist.py, `CMVAE_DIST_FORCE=1 python bench.py`)
_FORCE = os.environ.get('CMVAE_DIST_FORCE', '') == '1'


def exchanging():
    """True when the step exchanges BatchNorm statistics and gradients: several ranks (or CMVAE_DIST_FORCE)."""
    return world_size() > 1 or (_FORCE and is_dist())


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if (ws <= 1 and not _FORCE) or is_dist():
        return rank(), world_size()
    if ws <= 1:                                      # CMVAE_DIST_FORCE: a process group of one
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('MASTER_PORT', '29533')
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


def _draw_em_init_numpy(n_samples, k, b_global):
    """The reference's own call sequence (omniglot/model.py:351-356) through the numpy interpreter path: what
    SeedStream reproduces; kept as the checker of tests/test_seed_stream.py."""
    return np.stack([np.random.choice(n_samples, k, replace=False) for _ in range(b_global)], 0).astype(np.int64)


class SeedStream:
    """EM seeds of the global meta-batch — one ``np.random.choice(N, K, replace=False)`` per episode in episode
    order from numpy's GLOBAL generator, exactly as the reference draws them (omniglot/model.py:351-356) — taken
    off the step's critical path.

    Every rank draws the seeds of all `b_global` episodes and keeps its slice, so the numpy stream stays identical
    to a single-process run; that is O(world size) serial host work per step (numpy: 0.23 ms per draw at N = 9600,
    7.2 ms at 32 episodes, against a 12.5 ms GPU step plus the host's own launch work).  Two measures:
      * the draw runs in the library (`cmvae_em_seed_draw_host`: numpy's legacy algorithm restated, bit-identical
        indices and generator state, several times faster, no interpreter lock);
      * `next(..., prefetch=True)` returns the seeds of this step and starts the draw of the NEXT step on a helper
        thread.  The prefetched result is only used if numpy's global state is still the snapshot it was drawn
        from when it is collected; any other consumer of the stream in between (evaluation episodes at an epoch
        boundary, the data shuffle) simply voids it and the seeds are drawn again, in order, from the current
        state.  The stream is therefore the sequential one in every interleaving."""

    def __init__(self):
        self._pool = None
        self._pending = None

    @staticmethod
    def _state():
        name, key, pos, has_gauss, cached = np.random.get_state(legacy=True)
        if name != 'MT19937':
            raise RuntimeError('cmvae_amd: numpy global generator is %s, expected MT19937' % name)
        return np.ascontiguousarray(key, dtype=np.uint32).copy(), int(pos), has_gauss, cached

    @staticmethod
    def _run(key, pos, n_samples, k, b_global):
        import ctypes
        from ._lib import lib, check
        out = np.empty((b_global, k), dtype=np.int64)
        scratch = np.empty(n_samples, dtype=np.int64)
        cpos = ctypes.c_int(pos)
        check(lib.cmvae_em_seed_draw_host(key.ctypes.data, ctypes.byref(cpos), int(n_samples), int(k), int(b_global),
                                          out.ctypes.data, scratch.ctypes.data), 'cmvae_em_seed_draw_host')
        return out, key, cpos.value

    def draw(self, n_samples, k, b_global):
        """Synchronous draw: [b_global, k] int64, numpy's global state advanced as the reference's loop would."""
        key, pos, has_gauss, cached = self._state()
        out, key, pos = self._run(key, pos, n_samples, k, b_global)
        np.random.set_state(('MT19937', key, pos, has_gauss, cached))
        return out

    def _prefetch(self, n_samples, k, b_global):
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='cmvae_seeds')
        key, pos, has_gauss, cached = self._state()
        fut = self._pool.submit(self._run, key.copy(), pos, n_samples, k, b_global)
        self._pending = ((n_samples, k, b_global), key, pos, has_gauss, cached, fut)

    def next(self, n_samples, k, b_global, prefetch=False):
        """Seeds of this step [b_global, k]; with `prefetch` the next step's draw (same shape) starts right away."""
        out = None
        if self._pending is not None:
            shape, key0, pos0, has_gauss, cached, fut = self._pending
            self._pending = None
            res, key1, pos1 = fut.result()
            key_now, pos_now, hg_now, _ = self._state()
            if shape == (n_samples, k, b_global) and pos_now == pos0 and hg_now == has_gauss and \
                    np.array_equal(key_now, key0):
                np.random.set_state(('MT19937', key1, pos1, has_gauss, cached))
                out = res
        if out is None:
            out = self.draw(n_samples, k, b_global)
        if prefetch:
            self._prefetch(n_samples, k, b_global)
        return out


_seed_stream = SeedStream()


def draw_em_init(n_samples, k, b_global, r=None, ws=None, prefetch=False):
    """EM seeds for the global meta-batch, drawn exactly as the reference does — one
    ``np.random.choice(N, K, replace=False)`` per episode in episode order from numpy's global
    stream (omniglot/model.py:351-356) — on every rank, which then keeps its own slice.  The
    numpy stream therefore stays identical to a single-process run.  `prefetch`: start the next step's draw on a
    helper thread (SeedStream)."""
    idx = _seed_stream.next(n_samples, k, b_global, prefetch=prefetch)
    return np.ascontiguousarray(idx[shard_slice(b_global, r, ws)])


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
    values = values.contiguous()
    if values.is_cuda and tdist.get_backend() == 'gloo':          # staged on the calling thread (see _all_reduce)
        host = values.cpu()
        parts = [torch.empty_like(host) for _ in range(world_size())]
        tdist.all_gather(parts, host)
        parts = [p.to(values.device) for p in parts]
    else:
        parts = [torch.empty_like(values) for _ in range(world_size())]
        tdist.all_gather(parts, values)
    return torch.stack([p.view(-1, local_batch) for p in parts], 1).reshape(-1)


def global_sums(sums):
    """Collective C2: per-channel sums of a BatchNorm layer ({sum x, sum x^2} in the forward pass, {sum dy,
    sum dy*xhat} in the backward pass; 128 floats) summed over the ranks.  The one implementation of the exchange:
    the HIP pipelines (cmvae_amd/encoder.py, decoder.py) and the CPU restatement the gloo tests run
    (tests/sync_bn_reference.py) both call it."""
    if exchanging():
        g = sums.clone()
        all_reduce(g, 'bn_sums')
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
        if n_tail_params and exchanging():
            for p in self.params[self._tail_start:]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._tail_hook))
        return self

    def _tail_hook(self, _param):
        self._tail_pending -= 1
        if self._tail_pending < 0:
            # a second backward pass since zero(): the tail bucket already went out (and holds the ranks' sum), so
            # this gradient would be accumulated into exchanged values and never be reduced itself
            raise RuntimeError('cmvae_amd.FlatGrads: backward ran twice between zero() calls while the tail bucket is '
                               'exchanged during the backward pass; call zero() before every backward, or '
                               'overlap_tail(0) to accumulate gradients over several passes')
        if self._tail_pending == 0 and not self._tail_done:
            self._pack(self._tail_start, len(self.params))
            self._tail_work = all_reduce(self.flat[self._tail_off:], 'grad_tail', async_op=True)
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

    def exchange(self):
        """Mean over ranks of the packed buffer, one all-reduce of everything (for callers that pack elsewhere — the
        HIP-graph step packs inside the graph and exchanges after the replay)."""
        ws = world_size()
        if exchanging():
            all_reduce(self.flat, 'grad_all')
            self.flat.div_(ws)

    def allreduce(self):
        """Gather the gradients, then mean over ranks (every loss term is a mean over equal-sized shards)."""
        ws = world_size()
        if not exchanging():
            self.pack()
            return
        if self._tail_done:
            self._pack(0, self._tail_start)
            all_reduce(self.flat[:self._tail_off], 'grad_head')
            trace('wait grad_tail')
            self._tail_work.wait()               # stream-ordered on RCCL: the host does not block
            trace('waited grad_tail')
            self._tail_work = None
            self.flat.div_(ws)
        else:
            # no hook fired (overlap off, or a parameter of the tail got no gradient): one exchange of everything
            self.pack()
            self.exchange()
