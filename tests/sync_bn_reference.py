"""CPU restatement of the BatchNorm-statistics exchange (collective C2) on torch tensors, for the world_size-2
gloo tests: what is summed, what goes through `cmvae_amd.dist.global_sums` (the SAME function the HIP encoder /
decoder pipelines call between their kernels) and how the backward pass uses the global sums.  Test infrastructure
only."""
import torch

from cmvae_amd.dist import global_sums


class _SyncBN(torch.autograd.Function):
    """Training-mode BatchNorm2d whose statistics span all ranks."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        C = x.shape[1]
        red = [0, 2, 3]
        stats = torch.empty(2 * C + 1, dtype=x.dtype, device=x.device)
        stats[:C] = x.sum(red)
        stats[C:2 * C] = (x * x).sum(red)
        stats[2 * C] = x.numel() // C
        stats = global_sums(stats)
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
        local = global_sums(local)
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
