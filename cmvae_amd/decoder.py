"""OmniglotDecoder + Bernoulli likelihood as one autograd node on the HIP kernels
(omniglot/module.py:43-71, omniglot/model.py:458-466).

Forward:  H_rec [n,256] (channels-last: column (i*2+j)*64 + c) -> three ConvTranspose2d layers, the four
sub-pixel phases of a layer as ONE gather-GEMM launch (BatchNorm statistics from the GEMM epilogue,
BatchNorm+ReLU folded into the next layer's operand load) -> fused ConvTranspose2d(64,1)+Sigmoid+BCE.
Backward: last layer in two passes (sums by linearity in dlogit, then the apply pass that writes dy3), then
per layer: data-gradient GEMM with the BN-backward sums of the layer below from its epilogue, BN-backward
apply of the layer below, weight gradient of all four phases in one launch.
Pre-BatchNorm activations y1..y3 are the only tensors kept for the backward pass.
"""
import torch

from . import conv
from . import dist as _dist
from ._lib import lib, check

_p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
_LAYERS = ((4, 2), (3, 4), (4, 7))          # (kernel, input size) of the three 64->64 transposed convs
# rows of the decoder GEMMs enumerate (y, x, image): whole tiles skip the taps that fall outside the image (14 % of
# the multiply-adds of the 7x7 -> 14x14 layer, 31 % / 44 % of the two smaller ones would multiply zero padding).
# Measured on the layer-3 launches: forward phase 575 -> 512 us, weight gradient 635 -> 600 us, data gradient
# 2.34 -> 2.32 ms (its 16 gathered row sets per tile no longer overlap inside a tile: what the skipped taps save goes
# into memory traffic).
PIXEL_MAJOR = True
TAPS = None                                 # tests set this to a dict: receives the activations and folds of a forward


def _stream():
    return torch.cuda.current_stream().cuda_stream


def colsum(part):
    out = torch.empty(part.shape[1], device=part.device)
    check(lib.cmvae_colsum_reduce(_p(part), part.shape[0], part.shape[1], _p(out), _stream()), 'cmvae_colsum_reduce')
    return out


def bn_finalize(sums, count, bn, training_update=True):
    dev = sums.device
    scale, shift, mean, invstd = (torch.empty(64, device=dev) for _ in range(4))
    upd = training_update and bn.track_running_stats
    if bn.momentum is None:
        # nn.BatchNorm2d: cumulative moving average, 1 / num_batches_tracked counted after the increment (one
        # host read of the counter; the reference's layers all use the default momentum 0.1)
        mom = 1.0 / float(int(bn.num_batches_tracked) + 1) if upd else 0.0
    else:
        mom = bn.momentum
    check(lib.cmvae_bn_finalize(_p(sums), float(count), _p(bn.weight.detach()), _p(bn.bias.detach()), bn.eps, mom,
                                _p(bn.running_mean) if upd else None, _p(bn.running_var) if upd else None,
                                _p(scale), _p(shift), _p(mean), _p(invstd), _stream()), 'cmvae_bn_finalize')
    if upd:
        bn.num_batches_tracked += 1
    return scale, shift, mean, invstd


def bn_stats_finalize(part, count, bn):
    """Training-mode fold (scale, shift, mean, invstd) from the conv GEMM's partial statistics [nblk,128]; count =
    values per channel over ALL ranks.  One process: one launch (sums, finalize, running statistics and the batch
    counter); several: column sums, all-reduce (collective C2), finalize."""
    if _dist.exchanging() or bn.momentum is None:
        return bn_finalize(_global_sums(colsum(part)), count, bn)
    dev = part.device
    scale, shift, mean, invstd = (torch.empty(64, device=dev) for _ in range(4))
    upd = bn.track_running_stats
    check(lib.cmvae_bn_finalize_part(_p(part), part.shape[0], float(count), _p(bn.weight.detach()), _p(bn.bias.detach()),
                                     bn.eps, bn.momentum, _p(bn.running_mean) if upd else None,
                                     _p(bn.running_var) if upd else None, _p(bn.num_batches_tracked) if upd else None,
                                     _p(scale), _p(shift), _p(mean), _p(invstd), _stream()), 'cmvae_bn_finalize_part')
    return scale, shift, mean, invstd


def bn_bwd_sums_finalize(part, count):
    """(c1, c2, dgamma, dbeta) from partial BatchNorm-backward sums [nblk,128]; count = this rank's values per
    channel."""
    if _dist.exchanging():
        return bn_bwd_finalize(colsum(part), count)
    dev = part.device
    c1, c2, dgamma, dbeta = (torch.empty(64, device=dev) for _ in range(4))
    check(lib.cmvae_bn_bwd_finalize_part(_p(part), part.shape[0], float(count), _p(c1), _p(c2), _p(dgamma), _p(dbeta),
                                         _stream()), 'cmvae_bn_bwd_finalize_part')
    return c1, c2, dgamma, dbeta


def bn_eval_fold(bn):
    scale, shift, invstd = (torch.empty(64, device=bn.weight.device) for _ in range(3))
    check(lib.cmvae_bn_eval_fold(_p(bn.weight.detach()), _p(bn.bias.detach()), _p(bn.running_mean), _p(bn.running_var),
                                 bn.eps, _p(scale), _p(shift), _p(invstd), _stream()), 'cmvae_bn_eval_fold')
    return scale, shift, bn.running_mean, invstd


_global_sums = _dist.global_sums          # collective C2 (cmvae_amd/dist.py)


def bn_bwd_finalize(sums_local, count):
    dev = sums_local.device
    c1, c2, dgamma, dbeta = (torch.empty(64, device=dev) for _ in range(4))
    sums_global = _global_sums(sums_local)
    check(lib.cmvae_bn_bwd_finalize(_p(sums_local), _p(sums_global), float(count * _dist.world_size()), _p(c1), _p(c2),
                                    _p(dgamma), _p(dbeta), _stream()), 'cmvae_bn_bwd_finalize')
    return c1, c2, dgamma, dbeta


def bn_relu_bwd_sums(d_a, y, fold, masked):
    rows = y.shape[0]
    part = torch.empty(lib.cmvae_rowpass_grid(rows), 128, device=y.device)
    check(lib.cmvae_bn_relu_bwd_reduce(_p(d_a), _p(y), _p(fold[0]), _p(fold[1]), _p(fold[2]), _p(fold[3]), rows,
                                       int(masked), _p(part), _stream()), 'cmvae_bn_relu_bwd_reduce')
    return colsum(part)


def bn_relu_bwd_apply(d_a, y, fold, c1, c2, masked, out):
    check(lib.cmvae_bn_relu_bwd_apply(_p(d_a), _p(y), _p(fold[0]), _p(fold[1]), _p(fold[2]), _p(fold[3]), _p(c1),
                                      _p(c2), y.shape[0], int(masked), _p(out), _stream()), 'cmvae_bn_relu_bwd_apply')
    return out


def convT_forward(x, n, Hi, k, weight, bias, fold_in, want_stats):
    """x [n*Hi*Hi,64] (pre-fold) -> y [n*Ho*Ho,64]; returns (y, partial column sums {sum,sumsq} [nblk,128] or None)."""
    Ho = conv.convT_out(Hi, k)
    y = torch.empty(n * Ho * Ho, 64, device=x.device)
    wprep = weight.detach().permute(2, 3, 1, 0).contiguous()                 # [ky][kx][co][ci]: all k*k taps
    phases = [conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=PIXEL_MAJOR) for py in (0, 1) for px in (0, 1)]
    sc, sh = (fold_in[0], fold_in[1]) if fold_in is not None else (None, None)
    # all four phases in one launch (conv_gemm_wstat_multi_kernel)
    part = conv.conv_gemm_multi([g for g, _ in phases], x, wprep, y, bias=bias.detach(), a_scale=sc, a_shift=sh,
                                stats=want_stats)
    return y, (part if want_stats else None)


class _DecoderBCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h_rec, target, M, training, bns, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, g3, be3, w4, b4):
        n = h_rec.shape[0]
        ws, bs = (w1, w2, w3), (b1, b2, b3)
        acts, folds = [h_rec.detach().reshape(n * 4, 64)], [None]
        world = _dist.world_size()
        for li, (k, Hi) in enumerate(_LAYERS):
            y, part = convT_forward(acts[-1], n, Hi, k, ws[li], bs[li], folds[-1], want_stats=training)
            if training:
                count = y.shape[0] * world
                fold = bn_stats_finalize(part, count, bns[li])
            else:
                fold = bn_eval_fold(bns[li])
            acts.append(y)
            folds.append(fold)
        p = torch.empty(n, 784, device=h_rec.device)
        loss_part = torch.empty(n, device=h_rec.device)
        tgt = target.detach().reshape(-1, 784)
        if tgt.shape[0] * M != n:
            raise RuntimeError('cmvae_amd: decoder got %d images for %d targets x M=%d' % (n, tgt.shape[0], M))
        w4f = w4.detach().reshape(64, 16).contiguous()
        check(lib.cmvae_dec_out_fwd(_p(acts[3]), _p(folds[3][0]), _p(folds[3][1]), _p(w4f), _p(b4.detach()), _p(tgt), M,
                                    n, _p(p), _p(loss_part), _stream()), 'cmvae_dec_out_fwd')
        loss = loss_part.sum()
        ctx.training = training
        ctx.M = M
        ctx.n = n
        ctx.acts, ctx.folds = acts, folds
        if TAPS is not None:
            TAPS['dec'] = (acts, folds)
        ctx.save_for_backward(p, tgt, w1, w2, w3, w4f)
        ctx.mark_non_differentiable(p)
        ctx.set_materialize_grads(False)
        return p, loss

    @staticmethod
    def backward(ctx, _gp, g_loss):
        if not ctx.training:
            raise RuntimeError('cmvae_amd: decoder backward is only defined in training mode (batch statistics)')
        p, tgt, w1, w2, w3, w4f = ctx.saved_tensors
        ws = (w1, w2, w3)
        n, M, acts, folds = ctx.n, ctx.M, ctx.acts, ctx.folds
        dev = p.device
        d = torch.empty(n * 196, 64, device=dev)
        g_loss = g_loss.contiguous().float()
        # pass 1: dW4, db4 and the BatchNorm-backward sums of layer 3 (the masked gradient itself is not stored)
        # by linearity in dlogit: two contractions over the pixels, no gradient tile (csrc/decoder_ops.hip)
        L = lib.cmvae_dec_out_bwd_sums_partial_len()
        part = torch.empty(lib.cmvae_dec_out_bwd_sums_grid(n), L, device=dev)
        tot_lin = torch.empty(L, device=dev)
        tot = torch.empty(lib.cmvae_dec_out_bwd_partial_len(), device=dev)
        check(lib.cmvae_dec_out_bwd_sums(_p(acts[3]), *(_p(t) for t in folds[3]), _p(w4f), _p(p), _p(tgt), M, n,
                                         _p(g_loss), 1.0, _p(part), _p(tot_lin), _p(tot), _stream()),
              'cmvae_dec_out_bwd_sums')
        dw4 = tot[:1024].reshape(64, 1, 4, 4)
        db4 = tot[1152:1153]
        sums = tot[1024:1152]
        grads_w, grads_b, grads_g, grads_be = [None] * 3, [None] * 3, [None] * 3, [None] * 3
        # Per layer: data gradient (with the BatchNorm-backward sums of the layer below from its epilogue), finalize and
        # apply of that BatchNorm backward, weight gradient of all four phases in one launch.  (Measured and dropped:
        # the BatchNorm passes on a side stream next to the weight-gradient GEMMs — 14.1 -> 14.5 ms/step.)
        c1, c2, dgamma, dbeta = bn_bwd_finalize(sums, acts[3].shape[0])
        # pass 2 of the last layer's backward writes dy3 directly
        check(lib.cmvae_dec_out_bwd_apply(_p(acts[3]), *(_p(t) for t in folds[3]), _p(w4f), _p(p), _p(tgt), M, n,
                                          _p(g_loss), 1.0, _p(c1), _p(c2), _p(d), _stream()), 'cmvae_dec_out_bwd_apply')
        grads_g[2], grads_be[2] = dgamma, dbeta
        dy = d
        for li in (2, 1, 0):
            k, Hi = _LAYERS[li]
            x_in, fold_in = acts[li], folds[li]
            gd, order = conv.geom_convT_dgrad(n, Hi, Hi, k, pixel_major=PIXEL_MAJOR)
            wtd = ws[li].detach().permute(2, 3, 0, 1).contiguous()   # [ky][kx][ci][co]
            d_in = torch.empty(n * Hi * Hi, 64, device=dev)
            dy_below = None
            if li > 0:
                # the BatchNorm-backward sums of the layer below come out of the data gradient's epilogue
                fused_sums = conv.conv_gemm_bnbwd(gd, dy, wtd, d_in, x_in, fold_in)      # partials [nblk,128]
                c1b, c2b, grads_g[li - 1], grads_be[li - 1] = bn_bwd_sums_finalize(fused_sums, x_in.shape[0])
                dy_below = bn_relu_bwd_apply(d_in, x_in, fold_in, c1b, c2b, False, out=d_in)     # in place
            else:
                conv.conv_gemm(gd, dy, wtd, d_in)
            dW = torch.empty_like(ws[li])
            db = torch.empty(64, device=dev)
            sc, sh = (fold_in[0], fold_in[1]) if fold_in is not None else (None, None)
            phases = [conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=PIXEL_MAJOR)
                      for py in (0, 1) for px in (0, 1)]
            conv.conv_wgrad_multi([g for g, _ in phases], x_in, dy, dW, [t for _, t in phases], s_ci=64 * k * k,
                                  s_co=k * k, db=db, a_scale=sc, a_shift=sh)
            grads_w[li], grads_b[li] = dW, db
            d, dy = d_in, dy_below
        d_h = d.reshape(n, 256)
        return (d_h, None, None, None, None,
                grads_w[0], grads_b[0], grads_g[0], grads_be[0],
                grads_w[1], grads_b[1], grads_g[1], grads_be[1],
                grads_w[2], grads_b[2], grads_g[2], grads_be[2], dw4, db4)


# proj output column (pix*64 + c)  <-  reference column (c*4 + pix): channels-last view of [64,2,2]
_PERMS = {}


def nhwc_perm(device, hidden=64):
    """(perm, inverse) index tensors, built once per device: channels-last column j = pix*hidden + c takes the
    reference's column c*4 + pix."""
    key = (str(device), hidden)
    if key not in _PERMS:
        j = torch.arange(4 * hidden, device=device)
        perm = (j % hidden) * 4 + j // hidden
        _PERMS[key] = (perm, torch.argsort(perm))
    return _PERMS[key]


class _Permute(torch.autograd.Function):
    """x.index_select(dim, perm) for a permutation: the backward is the gather by the inverse permutation (one
    kernel each way, instead of advanced indexing's sort-based index_put)."""

    @staticmethod
    def forward(ctx, x, dim, perm, inv):
        ctx.dim, ctx.inv = dim, inv
        return x.index_select(dim, perm)

    @staticmethod
    def backward(ctx, g):
        return g.index_select(ctx.dim, ctx.inv), None, None, None


def permute(x, dim, perm, inv):
    return _Permute.apply(x, dim, perm, inv)


def decoder_bce(decoder, h_rec_nhwc, target, M):
    """Returns (X_rec probabilities [n,784], BCE sum).  h_rec_nhwc: [n,256] in channels-last order."""
    seq = decoder.decoder
    convs = (seq[0], seq[3], seq[6], seq[9])
    bns = (seq[1], seq[4], seq[7])
    args = []
    for i in range(3):
        args += [convs[i].weight, convs[i].bias, bns[i].weight, bns[i].bias]
    args += [convs[3].weight, convs[3].bias]
    return _DecoderBCE.apply(h_rec_nhwc.contiguous(), target, M, decoder.training, bns, *args)
