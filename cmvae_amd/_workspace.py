"""Scratch buffers of the C-ABI calls that take a caller-owned workspace (weight-gradient partials, transposed
weights): one per (device, purpose, stream) — launches on different streams may be in flight together — that only
ever grows.  A buffer that was outgrown stays alive: a captured HIP graph (cmvae_amd/graph_step.py, pipeline.py) has
the address it was recorded with baked in, and a replay must never write into memory the allocator has handed to
someone else."""
import torch

_cache = {}


def get(nbytes, device, purpose):
    key = (torch.device(device), purpose, torch.cuda.current_stream(device).cuda_stream)
    bufs = _cache.setdefault(key, [])
    if not bufs or bufs[-1].numel() < nbytes:
        bufs.append(torch.empty(int(nbytes), dtype=torch.uint8, device=device))
    return bufs[-1]
