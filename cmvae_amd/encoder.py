"""OmniglotEncoder as one autograd node on the HIP kernels (omniglot/module.py:7-40).

x [n,784] -> Conv(1,64) -> BN -> ReLU -> MaxPool(ceil) -> 3 x [Conv(64,64) -> BN -> ReLU -> MaxPool] ->
H [n,256] in channels-last order (column (i*2+j)*64 + c; the reference's column is c*4 + i*2+j — the
consumer permutes its weight columns instead of the data).  BatchNorm statistics come from the conv
epilogues, BatchNorm+ReLU is folded into the pooling kernel, and the backward pass recomputes the
pooling argmax from the saved pre-BatchNorm activations.  Layer 1 in training keeps no activation at all: its
pre-pooling output is recomputed from the images (nine multiply-adds per value) in the pooling pass and in a
one-pass backward that gets the conv1 weight gradient by linearity.
"""
import torch

from . import conv
from . import dist as _dist
from ._lib import lib, check
from .decoder import (colsum, bn_stats_finalize, bn_eval_fold, bn_bwd_sums_finalize, bn_relu_bwd_apply, _p, _stream)

_SIZES = (28, 14, 7, 4)          # conv input sizes; pooled outputs are 14, 7, 4, 2
TAPS = None                      # tests set this to a dict: receives the pre-BatchNorm outputs and folds of a forward


class _Encoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, training, bns, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, g3, be3, w4, b4, g4, be4):
        x = x.detach().reshape(-1, 784).contiguous()
        n = x.shape[0]
        dev = x.device
        ws, bs = (w1, w2, w3, w4), (b1, b2, b3, b4)
        world = _dist.world_size()
        ys, folds, pooled = [], [], []
        a = None
        for li, H in enumerate(_SIZES):
            rows = n * H * H
            Ho = (H + 1) // 2
            if li == 0 and not training:
                # evaluation: conv1 + BatchNorm(running statistics) + ReLU + pooling in one pass, y1 never exists
                fold = bn_eval_fold(bns[0])
                a = torch.empty(n * Ho * Ho, 64, device=dev)
                check(lib.cmvae_enc_conv1_pool_fwd(_p(x), _p(w1.detach().reshape(64, 9)), _p(b1.detach()), _p(fold[0]),
                                                   _p(fold[1]), n, H, H, _p(a), _stream()), 'cmvae_enc_conv1_pool_fwd')
                ys.append(None); folds.append(fold); pooled.append(a)
                continue
            if li == 0:
                # y1 is never stored: batch statistics from a pass that only sums, then conv1 + BatchNorm + ReLU +
                # pooling in one pass (the evaluation kernel with the batch fold); the backward recomputes y1 too
                part = torch.empty(lib.cmvae_rowpass_grid(rows), 128, device=dev)
                w9, b1d = w1.detach().reshape(64, 9), b1.detach()
                check(lib.cmvae_enc_conv1_fwd(_p(x), _p(w9), _p(b1d), n, H, H, None, _p(part), _stream()),
                      'cmvae_enc_conv1_fwd')
                fold = bn_stats_finalize(part, rows * world, bns[0])
                a = torch.empty(n * Ho * Ho, 64, device=dev)
                check(lib.cmvae_enc_conv1_pool_fwd(_p(x), _p(w9), _p(b1d), _p(fold[0]), _p(fold[1]), n, H, H, _p(a),
                                                   _stream()), 'cmvae_enc_conv1_pool_fwd')
                y = None
                if TAPS is not None:                       # tests look at the pre-BatchNorm outputs
                    y = torch.empty(rows, 64, device=dev)
                    check(lib.cmvae_enc_conv1_fwd(_p(x), _p(w9), _p(b1d), n, H, H, _p(y), None, _stream()),
                          'cmvae_enc_conv1_fwd')
                ys.append(y); folds.append(fold); pooled.append(a)
                continue
            y = torch.empty(rows, 64, device=dev)
            wt = ws[li].detach().permute(2, 3, 0, 1).reshape(9, 64, 64).contiguous()      # [tap][co][ci]
            part = conv.conv_gemm(conv.geom_conv3x3(n, H, H), a, wt, y, bias=bs[li].detach(), stats=training)
            if training:
                fold = bn_stats_finalize(part, rows * world, bns[li])
            else:
                fold = bn_eval_fold(bns[li])
            a = torch.empty(n * Ho * Ho, 64, device=dev)
            check(lib.cmvae_pool_fwd(_p(y), _p(fold[0]), _p(fold[1]), n, H, H, _p(a), _stream()), 'cmvae_pool_fwd')
            ys.append(y); folds.append(fold); pooled.append(a)
        ctx.training, ctx.n = training, n
        ctx.ys, ctx.folds, ctx.pooled = ys, folds, pooled
        if TAPS is not None:
            TAPS['enc'] = (ys, folds)
        ctx.save_for_backward(x, w2, w3, w4, w1, b1)
        return a.view(n, 256)

    @staticmethod
    def backward(ctx, d_h):
        if not ctx.training:
            raise RuntimeError('cmvae_amd: encoder backward is only defined in training mode (batch statistics)')
        x, w2, w3, w4, w1, b1 = ctx.saved_tensors
        ws = (None, w2, w3, w4)
        n, ys, folds, pooled = ctx.n, ctx.ys, ctx.folds, ctx.pooled
        dev = x.device
        d = d_h.contiguous().view(n * 4, 64)
        gw, gb, gg, gbe = [None] * 4, [None] * 4, [None] * 4, [None] * 4
        for li in (3, 2, 1, 0):
            H = _SIZES[li]
            Ho = (H + 1) // 2
            rows = n * H * H
            y, fold = ys[li], folds[li]
            if li == 0:
                # the weight gradient is linear in dybn and xhat: one pass that recomputes y1 from the images
                # accumulates everything, dybn is never written (cmvae_enc_conv1_bwd_lin_rc); c1, c2 enter in the
                # finalize
                part_m = torch.empty(256, 64, device=dev)
                check(lib.cmvae_enc_conv1_tap_sums(_p(x), n, H, H, _p(part_m), _stream()), 'cmvae_enc_conv1_tap_sums')
                part = torch.empty(lib.cmvae_enc_conv1_bwd_lin_rc_grid(n), lib.cmvae_enc_conv1_bwd_lin_cols(),
                                   device=dev)
                check(lib.cmvae_enc_conv1_bwd_lin_rc(_p(d), _p(x), _p(w1.detach().reshape(64, 9)), _p(b1.detach()),
                                                     *(_p(t) for t in fold), _p(part_m), n, H, H, _p(part),
                                                     _stream()), 'cmvae_enc_conv1_bwd_lin_rc')
                tot = colsum(part)
                sums_glob = _dist.global_sums(tot[:128])
                dW, gb[0], gg[0], gbe[0] = (torch.empty(k, device=dev) for k in (576, 64, 64, 64))
                check(lib.cmvae_enc_conv1_lin_finalize(_p(tot), _p(sums_glob), float(rows * _dist.world_size()), n, H, H,
                                                       _p(part_m), _p(w1.detach().reshape(64, 9)), _p(b1.detach()),
                                                       _p(fold[0]), _p(fold[2]), _p(fold[3]), _p(dW), _p(gb[0]),
                                                       _p(gg[0]), _p(gbe[0]), _stream()),
                      'cmvae_enc_conv1_lin_finalize')
                gw[0] = dW.reshape(64, 1, 3, 3)
                continue
            dybn = torch.empty(rows, 64, device=dev)
            part = torch.empty(lib.cmvae_rowpass_grid(n * Ho * Ho), 128, device=dev)
            check(lib.cmvae_pool_bwd(_p(d), _p(y), *(_p(t) for t in fold), n, H, H, _p(dybn), _p(part), _stream()),
                  'cmvae_pool_bwd')
            c1, c2, gg[li], gbe[li] = bn_bwd_sums_finalize(part, rows)
            dy = bn_relu_bwd_apply(dybn, y, fold, c1, c2, True, out=dybn)
            a_in = pooled[li - 1]
            dW = torch.empty_like(ws[li])
            db = torch.empty(64, device=dev)
            conv.conv_wgrad(conv.geom_conv3x3(n, H, H), a_in, dy, dW, list(range(9)), s_ci=9, s_co=576, db=db)
            gw[li], gb[li] = dW, db
            wtd = ws[li].detach().permute(2, 3, 1, 0).reshape(9, 64, 64).contiguous()     # [tap][ci][co]
            d = torch.empty(rows, 64, device=dev)
            conv.conv_gemm(conv.geom_conv3x3(n, H, H, transpose=True), dy, wtd, d)
        out = [None, None, None]
        for li in range(4):
            out += [gw[li], gb[li], gg[li], gbe[li]]
        return tuple(out)


def encode_images(encoder, x):
    """x [n,1,28,28] (or [n,784]) -> H [n,256], channels-last column order."""
    seq = encoder.encoder
    args = []
    bns = []
    for i in range(4):
        cv, bn = seq[4 * i], seq[4 * i + 1]
        bns.append(bn)
        args += [cv.weight, cv.bias, bn.weight, bn.bias]
    return _Encoder.apply(x, encoder.training, tuple(bns), *args)
