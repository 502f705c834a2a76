"""The training step replayed from a HIP graph.

The feature-model step (mimgnet / celeba: `mimgnet/trainer.py:84-106`) is ~230 small launches for ~1.6 ms of GPU work:
issued eagerly it runs at the host's launch rate (2.3-2.7 ms/step depending on the box's CPU); captured once —
gradient reset, forward, backward, gradient pack, Adam — and replayed it takes 1.8 ms.  The Omniglot step is bound by
its kernels (14.1 ms eager, 14.3 ms replayed: tools/probe_graph_train.py), so `Trainer` only switches this on for the
feature variants (`--graph-step auto`).

What makes the step replayable: inputs, EM seeds and the reparametrisation noise live in static buffers that are
refilled before every replay (the noise by the same `torch.randn` call, in the same generator order, as the eager
step draws it: same realisation); the loss constants are frozen into the graph, so a change of (h_lambda, h_c,
a_lambda) — `--dual-ascent` moves them once per epoch — drops the graph of that shape and captures a new one (one
graph per shape at any time: a graph's private pool holds a whole step of activations);
Adam reads its step count from device memory (`FlatAdam.device_step`, `cmvae_adam_step_dev`).  With several ranks the
graph ends after the gradient pack: the all-reduce (one bucket; the early tail bucket is switched off, a collective
cannot be issued from a hook inside a capture) and the Adam launch follow every replay eagerly, and the noise buffer
takes this rank's slice of the global draw (`dist.draw_eps`), as the eager sharded step does.
"""
import torch

from . import dist as cdist


class GraphedStep:
    WARMUP = 3          # eager steps before the capture (lazy initialisations, workspace caches)

    def __init__(self, model, flat, optimizer):
        self.model, self.flat, self.optimizer = model, flat, optimizer
        self.device = flat.flat.device
        if self.device.type != 'cuda':
            raise RuntimeError('cmvae_amd: GraphedStep needs the model on the GPU (there is no CPU path)')
        optimizer.device_step()
        self.world, self.rank = cdist.world_size(), cdist.rank()
        self.exchanging = cdist.exchanging()
        if self.exchanging:
            flat.overlap_tail(0)
        self.calls = 0
        self.graphs = {}            # shapes -> (consts, graph, X, idx, eps, outputs)

    def _body(self, X, idx, eps, consts):
        self.model.train()
        self.flat.zero()
        _, rec, kl, h_aug, h = self.model(X, *consts, eps=eps, init_idx=idx)
        (rec + kl + h_aug).backward()
        self.flat.pack()
        if not self.exchanging:
            self.optimizer.step()
        return rec.detach(), kl.detach(), h_aug.detach(), h.detach()

    def _after(self):
        """What follows the graph when the job has several ranks: gradient exchange and update."""
        if self.exchanging:
            self.flat.exchange()
            self.optimizer.step()

    def _draw(self, shape, out=None):
        """The step's reparametrisation noise, by the same generator calls as the eager step."""
        if not self.exchanging:
            if out is None:
                return torch.randn(shape, device=self.device, dtype=torch.float32)
            return torch.randn(shape, out=out)
        eps = cdist.draw_eps(shape[0] * self.world, shape[1], shape[2], shape[3], self.device, self.rank, self.world)
        return eps if out is None else out.copy_(eps)

    def _eps_shape(self, X):
        m = self.model
        return (X.shape[0], X.shape[1], m.train_mc_sample_size, m.latent_size)

    def __call__(self, X, idx, h_lambda, h_c, a_lambda):
        """One step on the meta-batch X [B,S,...] with EM seeds idx [B,K] (device tensors); returns the four loss
        tensors of `GMVAE.forward` (rec, kl, h_augment, h) — static buffers, valid until the next call."""
        consts = (float(h_lambda), float(h_c), float(a_lambda))
        self.calls += 1
        if self.calls <= self.WARMUP:
            out = self._body(X, idx, self._draw(self._eps_shape(X)), consts)
            self._after()
            return out
        key = (tuple(X.shape), tuple(idx.shape), X.dtype)
        entry = self.graphs.get(key)
        if entry is not None and entry[0] != consts:
            # new loss constants: the old graph (and the step of activations in its pool) goes before the new capture
            del self.graphs[key]
            entry[1].reset()
            entry = None
        if entry is None:
            sX, sidx = X.clone(), idx.clone()
            seps = torch.zeros(self._eps_shape(X), device=self.device, dtype=torch.float32)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):           # records only: no weight is touched until the replay below
                out = self._body(sX, sidx, seps, consts)
            entry = self.graphs[key] = (consts, graph, sX, sidx, seps, out)
        _, graph, sX, sidx, seps, out = entry
        sX.copy_(X, non_blocking=True)
        sidx.copy_(idx, non_blocking=True)
        self._draw(tuple(seps.shape), out=seps)
        graph.replay()
        self._after()
        return out
