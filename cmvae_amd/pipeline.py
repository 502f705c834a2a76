"""Few-shot evaluation calls in flight on several HIP streams, each replayed from a HIP graph.

`Trainer.eval` (omniglot/trainer.py:218-240) issues `eval_episodes / batch_size` independent `prediction` calls in a
row.  At the reference's 4 episodes per call one call is a chain of ~130 small dependent kernels (encoder, two SABs,
ten semi-supervised EM iterations of three launches each) that occupies a fraction of the 256 CUs:
  * on ONE stream the call is bound by that chain's latency on the GPU — a hipGraph replay of it takes as long as the
    eager call (1.15 vs 1.17 ms, tools/probe_graph.py);
  * dealt round-robin to a few streams the chains overlap on the GPU, and then the host's ~0.75 ms of launch work per
    call is the limit (3400 -> 4260 episodes/s with four streams);
  * so every stream replays its call from a captured graph (host cost of a replay: ~15 us): the chains overlap and
    the host keeps up.
The calls do not depend on each other; results come back in submission order.  Inputs are copied into per-stream
static buffers, outputs are cloned out of them.  The reparametrisation noise is NOT drawn inside the graphs: every
graph of a generator shares one device-side Philox offset, which the replays of different streams would overwrite
under each other.  Each call's noise is drawn eagerly, on the host thread in submission order, into the slot's
static buffer (the same generator calls as the sequential loop, so the same realisation and bit-identical labels)
and handed to `prediction(..., eps=)`.  On the streams alone: 4700 episodes/s, with the graphs 5200-5700 (one box,
B = 4).
"""
import torch


class _Slot:
    """One stream and the graphs captured on it, keyed by the shapes of a call."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.graphs = {}


class PredictionPipeline:
    def __init__(self, model, n_streams=8, use_graphs=None):
        self.model = model
        dev = next(model.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('cmvae_amd: PredictionPipeline needs the model on the GPU (there is no CPU path)')
        if model.training:
            raise RuntimeError('cmvae_amd: PredictionPipeline is for model.eval() (BatchNorm running statistics)')
        self.device = dev
        self.n = max(1, int(n_streams))
        self.use_graphs = (self.n > 1) if use_graphs is None else bool(use_graphs)
        self.slots = [_Slot(dev) for _ in range(self.n)] if (self.n > 1 or self.use_graphs) else []
        self.i = 0
        self.out = []

    def _eps_shape(self, X_tr, X_te):
        m = self.model
        return (X_tr.shape[0], X_tr.shape[1] + X_te.shape[1], m.train_mc_sample_size, m.latent_size)

    def _capture(self, slot, key, X_tr, y_tr, X_te):
        """Static input buffers + one captured `prediction` call on the slot's stream (called on that stream)."""
        sx_tr, sy_tr, sx_te = X_tr.clone(), y_tr.clone(), X_te.clone()
        seps = torch.zeros(self._eps_shape(X_tr, X_te), device=self.device, dtype=torch.float32)
        for _ in range(2):                               # lazy initialisations happen outside the capture
            self.model.prediction(sx_tr, sy_tr, sx_te, eps=seps)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=slot.stream):
            y = self.model.prediction(sx_tr, sy_tr, sx_te, eps=seps)
        slot.graphs[key] = (graph, sx_tr, sy_tr, sx_te, seps, y)
        return slot.graphs[key]

    def submit(self, X_tr, y_tr, X_te):
        """Enqueue one `model.prediction` call; returns its index in `results()`."""
        if not self.slots:
            self.out.append(self.model.prediction(X_tr, y_tr, X_te))
            return len(self.out) - 1
        main = torch.cuda.current_stream(self.device)
        slot = self.slots[self.i % len(self.slots)]
        self.i += 1
        s = slot.stream
        s.wait_stream(main)                              # the inputs were produced on the caller's stream
        with torch.cuda.stream(s):
            if self.use_graphs:
                key = (tuple(X_tr.shape), tuple(y_tr.shape), tuple(X_te.shape), X_tr.dtype)
                entry = slot.graphs.get(key) or self._capture(slot, key, X_tr, y_tr, X_te)
                graph, sx_tr, sy_tr, sx_te, seps, y_static = entry
                torch.randn(seps.shape, out=seps)        # the eager loop's draw, in submission order
                sx_tr.copy_(X_tr, non_blocking=True)
                sy_tr.copy_(y_tr, non_blocking=True)
                sx_te.copy_(X_te, non_blocking=True)
                graph.replay()
                y = y_static.clone()
            else:
                y = self.model.prediction(X_tr, y_tr, X_te)
        for t in (X_tr, y_tr, X_te):
            t.record_stream(s)
        self.out.append(y)
        return len(self.out) - 1

    def results(self):
        """All predictions so far, in submission order, usable on the caller's stream; resets the pipeline."""
        main = torch.cuda.current_stream(self.device)
        for slot in self.slots:
            main.wait_stream(slot.stream)
        for y in self.out:
            y.record_stream(main)
        out, self.out = self.out, []
        return out
