"""fp32-MFMA gather-GEMM kernels (conv 3x3, transposed-conv phases, data/weight gradients, Linear)
against PyTorch CPU fp32 references of the same ops.  Tolerance 1e-5 relative (exact-fp32 fma chains
only differ from the reference in summation order)."""
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5


def nhwc(x):       # [n,C,H,W] -> [n*H*W, C]
    return x.permute(0, 2, 3, 1).contiguous().view(-1, x.shape[1])


def from_nhwc(m, n, H, W):
    return m.view(n, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize('n,H,W', [(3, 7, 7), (5, 14, 14), (2, 4, 4), (130, 2, 2), (16, 7, 7), (48, 5, 3), (160, 2, 2),
                                   # >= 196 608 rows: forward and data gradient as two passes (5 + 4 taps) on the bf16 pipe
                                   (1040, 14, 14)])
def test_conv3x3_fwd_dgrad_wgrad(device, n, H, W):
    from cmvae_amd import conv
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 64, H, W, generator=g, requires_grad=True)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(64, generator=g).requires_grad_(True)
    y = F.conv2d(x, w, b, 1, 1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = nhwc(x.detach()).to(device)
    wt = w.detach().permute(2, 3, 0, 1).contiguous().to(device)            # [9][co][ci]
    out = torch.empty(n * H * W, 64, device=device)
    geom = conv.geom_conv3x3(n, H, W)
    part = conv.conv_gemm(geom, xd, wt, out, bias=b.detach().to(device), stats=True)
    assert rel_err(from_nhwc(out.cpu(), n, H, W), y) < TOL
    st = part.sum(0).cpu()
    assert rel_err(st[:64], y.detach().sum((0, 2, 3))) < 1e-4
    assert rel_err(st[64:], (y.detach() ** 2).sum((0, 2, 3))) < 1e-4
    # data gradient
    gyd = nhwc(gy).to(device)
    wtd = w.detach().permute(2, 3, 1, 0).contiguous().to(device)           # [9][ci][co]
    dx = torch.empty(n * H * W, 64, device=device)
    conv.conv_gemm(conv.geom_conv3x3(n, H, W, transpose=True), gyd, wtd, dx)
    assert rel_err(from_nhwc(dx.cpu(), n, H, W), x.grad) < TOL
    # weight gradient: dW[co][ci][ky][kx]
    dW = torch.zeros(64, 64, 3, 3, device=device)
    db = torch.zeros(64, device=device)
    conv.conv_wgrad(geom, xd, gyd, dW, [t for t in range(9)], s_ci=9, s_co=64 * 9, db=db)
    assert rel_err(dW, w.grad) < TOL
    assert rel_err(db, b.grad) < TOL


@pytest.mark.parametrize('n,Hi,k,pm', [(6, 2, 4, False), (5, 4, 3, False), (3, 7, 4, False), (70, 7, 4, False),
                                          (6, 2, 4, True), (5, 4, 3, True), (70, 7, 4, True), (256, 7, 4, True),
                                          (300, 4, 3, True), (641, 2, 4, True), (200, 7, 4, True),
                                          # image counts that are multiples of 16: the streaming weight-gradient kernel
                                          # (1, 2 and 4 taps per phase; fewer chunks than streams; either row order)
                                          (16, 2, 4, False), (32, 4, 3, False), (48, 7, 4, False), (304, 4, 3, True),
                                          (1024, 7, 4, True), (2000, 2, 4, True),
                                          # >= 196 608 rows with 1 / 2 / 2 / 4 taps per phase (k = 3): the bf16-pipe forward next to the
                                          # fp32 streaming weight gradient (the bf16-pipe one takes four-tap phases: k = 4)
                                          (4128, 4, 3, True)])
def test_convT_phases_with_bn_fold(device, n, Hi, k, pm):
    """Transposed conv over relu(bn(x)) folded into the operand load; dgrad; wgrad.  pm: pixel-major row order
    (rows enumerate (y, x, image); tiles that sit at one grid position skip the taps outside the image) — image
    counts that are multiples of the 128-row tile, not multiples of it (tiles spanning two grid positions) and
    smaller than it."""
    from cmvae_amd import conv
    g = torch.Generator().manual_seed(k + Hi)
    x = torch.randn(n, 64, Hi, Hi, generator=g)
    sc = torch.rand(64, generator=g) + 0.5
    sh = torch.randn(64, generator=g) * 0.3
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    w = (torch.randn(64, 64, k, k, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(64, generator=g).requires_grad_(True)
    y = F.conv_transpose2d(a, w, b, 2, 1)
    Ho = y.shape[-1]
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = nhwc(x).to(device)
    scd, shd, bd = sc.to(device), sh.to(device), b.detach().to(device)
    wprep = w.detach().permute(2, 3, 1, 0).contiguous().view(k * k, 64, 64).to(device)   # [tap][co][ci]
    out = torch.full((n * Ho * Ho, 64), float('nan'), device=device)
    gyd = nhwc(gy).to(device)
    dW = torch.zeros(64, 64, k, k, device=device)
    db = torch.zeros(64, device=device)
    ssum = torch.zeros(128)
    first = True
    for py in (0, 1):
        for px in (0, 1):
            geom, tap_idx = conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=pm)
            wp = wprep
            part = conv.conv_gemm(geom, xd, wp, out, bias=bd, a_scale=scd, a_shift=shd, stats=True)
            ssum += part.sum(0).cpu()
            conv.conv_wgrad(geom, xd, gyd, dW, tap_idx, s_ci=64 * k * k, s_co=k * k, db=db, accumulate_db=not first,
                            a_scale=scd, a_shift=shd)
            first = False
    assert rel_err(from_nhwc(out.cpu(), n, Ho, Ho), y) < TOL
    assert rel_err(ssum[:64], y.detach().sum((0, 2, 3))) < 1e-4
    assert rel_err(dW, w.grad) < TOL
    assert rel_err(db, b.grad) < TOL
    # all four phases in one launch: the same rows by the same code, bit for bit; statistics over all phases
    for fold in (True, False):
        geoms = [conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=pm)[0] for py in (0, 1) for px in (0, 1)]
        o1 = torch.full((n * Ho * Ho, 64), float('nan'), device=device)
        for geom in geoms:
            conv.conv_gemm(geom, xd, wprep, o1, bias=bd, a_scale=scd if fold else None, a_shift=shd if fold else None)
        om = torch.full((n * Ho * Ho, 64), float('nan'), device=device)
        pmulti = conv.conv_gemm_multi(geoms, xd, wprep, om, bias=bd, a_scale=scd if fold else None,
                                      a_shift=shd if fold else None, stats=True)
        if conv.on_bf16_pipe('multi', geoms):
            # large pixel-major launches run their products on the bf16 pipe (three-piece split, six products per
            # fp32 product: csrc/conv_gemm.hip mfma_split6): the same sums to fp32 rounding, not bit for bit
            assert rel_err(om, o1) < 1e-6 and float((om - o1).abs().max()) < 2e-6 * float(o1.abs().max())
        else:
            assert torch.equal(om, o1)
        assert pmulti.shape == (conv.gemm_multi_grid(), 128)
        o64 = o1.double().cpu()
        o64 = om.double().cpu()
        assert rel_err(pmulti.sum(0).cpu(), torch.cat([o64.sum(0), (o64 ** 2).sum(0)]).float()) < 1e-5
    # ... and their weight gradients in one launch + one reduction
    for fold in (True, False):
        phases = [conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=pm) for py in (0, 1) for px in (0, 1)]
        dWm = torch.full((64, 64, k, k), float('nan'), device=device)
        dbm = torch.full((64,), float('nan'), device=device)
        conv.conv_wgrad_multi([ge for ge, _ in phases], xd, gyd, dWm, [t for _, t in phases], s_ci=64 * k * k,
                              s_co=k * k, db=dbm, a_scale=scd if fold else None, a_shift=shd if fold else None)
        if fold:
            assert rel_err(dWm, w.grad) < TOL
        else:
            ref = torch.autograd.grad(F.conv_transpose2d(x.requires_grad_(True), w, b, 2, 1), w, gy)[0]
            assert rel_err(dWm, ref) < TOL
        assert rel_err(dbm, b.grad) < TOL
    # data gradient wrt a (the post-activation input)
    gd, order = conv.geom_convT_dgrad(n, Hi, Hi, k, pixel_major=pm)
    wtd = w.detach().permute(2, 3, 0, 1).contiguous().to(device)      # [ky][kx][ci][co]
    da = torch.empty(n * Hi * Hi, 64, device=device)
    conv.conv_gemm(gd, gyd, wtd, da)
    assert rel_err(from_nhwc(da.cpu(), n, Hi, Hi), a.grad) < TOL
    # the data gradient with the BatchNorm-backward sums of the layer below from its epilogue: the same gradient bit
    # for bit, the sums of bn_relu_bwd_sums (a pass of its own over the gradient and x) to rounding
    from cmvae_amd import decoder
    mean = torch.randn(64, generator=g) * 0.2
    invstd = torch.rand(64, generator=g) + 0.5
    fold = tuple(t.to(device) for t in (sc, sh, mean, invstd))
    da3 = torch.full_like(da, float('nan'))
    part = conv.conv_gemm_bnbwd(gd, gyd, wtd, da3, xd, fold)
    assert torch.equal(da3, da)
    want = decoder.bn_relu_bwd_sums(da, xd, fold, masked=False)
    got = decoder.colsum(part)
    assert rel_err(got[:64], want[:64]) < 1e-5 and rel_err(got[64:], want[64:]) < 1e-5
    m64 = (xd.double().cpu() * sc.double() + sh.double()) > 0
    g64 = da.double().cpu() * m64
    ref = torch.cat([g64.sum(0), (g64 * (xd.double().cpu() - mean.double()) * invstd.double()).sum(0)])
    assert rel_err(got.cpu(), ref.float()) < 1e-5
    if pm:
        # the same per-row sums in the same tap order: forward and data gradient agree bit for bit with the
        # image-major launches (also without a folded BatchNorm, as in the decoder's first layer)
        for fold in (True, False):
            outs = []
            for flag in (False, True):
                o = torch.full((n * Ho * Ho, 64), float('nan'), device=device)
                for py in (0, 1):
                    for px in (0, 1):
                        geom, _ = conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=flag)
                        conv.conv_gemm(geom, xd, wprep, o, bias=bd, a_scale=scd if fold else None,
                                       a_shift=shd if fold else None)
                outs.append(o)
            assert torch.equal(outs[0], outs[1])
        da2 = torch.empty_like(da)
        conv.conv_gemm(conv.geom_convT_dgrad(n, Hi, Hi, k)[0], gyd, wtd, da2)
        assert torch.equal(da, da2)


def test_bf16_pipe_products_are_fp32_products(device):
    """The big decoder launches run their fp32 products on the bf16 matrix pipe: each operand is the exact sum of three
    bf16 pieces and six of the nine piece products are accumulated in fp32 (csrc/conv_gemm.hip: mfma_split6).  What
    is dropped is below 2^-24 of |a b|, fp32's own rounding, so the results must be as close to float64 as an fp32
    computation is.  Layer-3-shaped transposed conv (64 -> 64, k 4, stride 2) over 1024 images of 14 x 14 — forward
    (802 816 rows), data gradient (200 704 rows, four accumulating passes of one tap parity class each) and weight
    gradient (32-image chunks) all
    take that path — against float64 on the CPU, next to PyTorch's own fp32 results on the same GPU."""
    from cmvae_amd import conv
    n, Hi, k = 1024, 14, 4
    Ho = conv.convT_out(Hi, k)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(n, 64, Hi, Hi, generator=g)
    sc = torch.rand(64, generator=g) + 0.5
    sh = torch.randn(64, generator=g) * 0.3
    a = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    w = torch.randn(64, 64, k, k, generator=g) * 0.1
    gy = torch.randn(n, 64, Ho, Ho, generator=g)
    xd, gyd = nhwc(x).to(device), nhwc(gy).to(device)
    scd, shd = sc.to(device), sh.to(device)
    # float64 references: forward and data gradient on a few images, the weight gradient over all of them
    sel = torch.arange(0, n, 171)
    y64 = F.conv_transpose2d(a[sel].double(), w.double(), None, 2, 1)
    da64 = F.conv2d(gy[sel].double(), w.double(), None, 2, 1)
    a64 = a.double().requires_grad_(False)
    w64 = w.double().requires_grad_(True)
    F.conv_transpose2d(a64, w64, None, 2, 1).backward(gy.double())
    dw64 = w64.grad
    # PyTorch fp32 on the GPU
    ag, wg = a.to(device), w.to(device).requires_grad_(True)
    yt = F.conv_transpose2d(ag, wg, None, 2, 1)
    yt.backward(gy.to(device))
    dat = F.conv2d(gy.to(device)[sel.to(device)], w.to(device), None, 2, 1)
    # this library
    wprep = w.permute(2, 3, 1, 0).contiguous().view(k * k, 64, 64).to(device)
    phases = [conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=True) for py in (0, 1) for px in (0, 1)]
    out = torch.full((n * Ho * Ho, 64), float('nan'), device=device)
    conv.conv_gemm_multi([ge for ge, _ in phases], xd, wprep, out, a_scale=scd, a_shift=shd)
    dW = torch.full((64, 64, k, k), float('nan'), device=device)
    conv.conv_wgrad_multi([ge for ge, _ in phases], xd, gyd, dW, [t for _, t in phases], s_ci=64 * k * k, s_co=k * k,
                          a_scale=scd, a_shift=shd)
    gd, _ = conv.geom_convT_dgrad(n, Hi, Hi, k, pixel_major=True)
    wtd = w.permute(2, 3, 0, 1).contiguous().to(device)
    da = torch.full((n * Hi * Hi, 64), float('nan'), device=device)
    fold = (scd, shd, torch.zeros(64, device=device), torch.ones(64, device=device))
    conv.conv_gemm_bnbwd(gd, gyd, wtd, da, xd, fold)

    def err(v, ref):
        return float((v.double().cpu() - ref).abs().max() / ref.abs().max())
    mine = {'forward': err(from_nhwc(out.cpu(), n, Ho, Ho)[sel], y64), 'dgrad': err(from_nhwc(da.cpu(), n, Hi, Hi)[sel], da64),
            'wgrad': err(dW, dw64)}
    torch32 = {'forward': err(yt.detach()[sel.to(device)], y64), 'dgrad': err(dat, da64), 'wgrad': err(wg.grad, dw64)}
    print('  max |error| / max |float64 value|: this library %s; torch fp32 on the GPU %s' % (mine, torch32))
    for key in mine:
        # at least as close to float64 as an fp32 computation: 1e-6 of scale is a few fp32 roundings of a 1024-term
        # sum (a bf16-only product would sit near 4e-3, a two-piece split near 1e-5)
        assert mine[key] < 1e-6, (key, mine[key])
        assert mine[key] < 2.0 * torch32[key] + 2e-7, (key, mine[key], torch32[key])


def test_dgrad_bn_sums_epilogue_weight_stationary_kernel(device):
    """Layer-2 shape of the training step (9-tap stride-2 data gradient, >= 196608 rows: two accumulating passes of
    6 + 3 taps on the bf16 pipe, cut at the taps' parity classes): gradient equal to the plain launch, sums equal to the
    separate reduction pass."""
    from cmvae_amd import conv, decoder
    n, Hi, k = 12304, 4, 3
    Ho = conv.convT_out(Hi, k)
    g = torch.Generator().manual_seed(5)
    gy = torch.randn(n * Ho * Ho, 64, generator=g).to(device)
    x = torch.randn(n * Hi * Hi, 64, generator=g).to(device)
    wtd = (torch.randn(k, k, 64, 64, generator=g) * 0.1).to(device)
    fold = tuple(t.to(device) for t in (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3,
                                        torch.randn(64, generator=g) * 0.2, torch.rand(64, generator=g) + 0.5))
    gd, _ = conv.geom_convT_dgrad(n, Hi, Hi, k, pixel_major=True)
    want = torch.empty(n * Hi * Hi, 64, device=device)
    conv.conv_gemm(gd, gy, wtd, want)
    got = torch.full_like(want, float('nan'))
    part = conv.conv_gemm_bnbwd(gd, gy, wtd, got, x, fold)
    # (both run their products on the bf16 pipe in two accumulating passes — the plain launch 5 + 4 taps in tap order,
    # the one with the sums 6 + 3 by parity class: the same sums to fp32 rounding)
    assert rel_err(got, want) < 1e-6 and float((got - want).abs().max()) < 1e-5 * float(want.abs().max())
    s_want = decoder.bn_relu_bwd_sums(got, x, fold, masked=False)
    s_got = decoder.colsum(part)
    assert rel_err(s_got[:64], s_want[:64]) < 1e-5 and rel_err(s_got[64:], s_want[64:]) < 1e-5


@pytest.mark.parametrize('nblk', [1, 7, 64, 512, 768])
def test_bn_finalize_from_partials_equals_two_step_path(device, nblk):
    """One launch (column sums + finalize + running statistics + batch counter) against colsum -> finalize; same
    for the backward finalize.  The sums are taken in another order: equal to rounding."""
    from cmvae_amd import decoder
    g = torch.Generator().manual_seed(nblk)
    count = 1000.0 * nblk
    x = torch.randn(nblk, 64, generator=g).abs() * 1000 / nblk
    part = torch.cat([x * 0.3, x * 0.3 * 0.3 + x], 1).to(device)          # {sum, sum of squares}: variance > 0
    bns = []
    for _ in range(2):
        bn = torch.nn.BatchNorm2d(64).to(device)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(64, generator=torch.Generator().manual_seed(1)) + 0.5)
            bn.bias.copy_(torch.randn(64, generator=torch.Generator().manual_seed(2)))
        bns.append(bn)
    want = decoder.bn_finalize(decoder.colsum(part), count, bns[0])
    got = decoder.bn_stats_finalize(part, count, bns[1])
    for a, b in zip(got, want):
        assert rel_err(a, b) < 1e-5
    assert rel_err(bns[1].running_mean, bns[0].running_mean) < 1e-5
    assert rel_err(bns[1].running_var, bns[0].running_var) < 1e-5
    assert int(bns[1].num_batches_tracked) == int(bns[0].num_batches_tracked) == 1
    for a, b in zip(decoder.bn_bwd_sums_finalize(part, count), decoder.bn_bwd_finalize(decoder.colsum(part), count)):
        assert rel_err(a, b) < 1e-5


def test_dgrad_bn_sums_sixteen_taps_in_accumulating_passes(device):
    """Layer-3 shape of the training step (16-tap stride-2 data gradient over >= 196608 rows, image count not a multiple
    of the 256-row tile): four weight-stationary launches on the bf16 pipe, one per parity class of the taps, the later
    ones adding to the output.  The partial sums are added in another order than the single accumulator chain of the
    direct kernel: equal to rounding; the sums equal the separate reduction."""
    from cmvae_amd import conv, decoder
    n, Hi, k = 4112, 7, 4
    Ho = conv.convT_out(Hi, k)
    g = torch.Generator().manual_seed(9)
    gy = torch.randn(n * Ho * Ho, 64, generator=g).to(device)
    x = torch.randn(n * Hi * Hi, 64, generator=g).to(device)
    wtd = (torch.randn(k, k, 64, 64, generator=g) * 0.1).to(device)
    fold = tuple(t.to(device) for t in (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3,
                                        torch.randn(64, generator=g) * 0.2, torch.rand(64, generator=g) + 0.5))
    gd, _ = conv.geom_convT_dgrad(n, Hi, Hi, k, pixel_major=True)
    want = torch.empty(n * Hi * Hi, 64, device=device)
    conv.conv_gemm(gd, gy, wtd, want)                       # direct kernel, one chain over the 16 taps
    got = torch.full_like(want, float('nan'))
    part = conv.conv_gemm_bnbwd(gd, gy, wtd, got, x, fold)
    assert rel_err(got, want) < 1e-6
    assert float((got - want).abs().max()) < 1e-5 * float(want.abs().max())
    s_want = decoder.bn_relu_bwd_sums(got, x, fold, masked=False)
    s_got = decoder.colsum(part)
    assert rel_err(s_got[:64], s_want[:64]) < 1e-5 and rel_err(s_got[64:], s_want[64:]) < 1e-5
    # fp64 reference on a few images
    sel = torch.arange(0, n, 517)
    gy4 = gy.view(n, Ho, Ho, 64)[sel].permute(0, 3, 1, 2).double().cpu()   # (pixel-major is the launch's row ORDER only)
    w = wtd.permute(2, 3, 0, 1).double().cpu()                                          # [ci][co][ky][kx]
    ref = F.conv2d(gy4, w, None, 2, 1)                                                  # dgrad of convT(k, 2, 1)
    mine = got.view(n, Hi, Hi, 64)[sel].permute(0, 3, 1, 2).double().cpu()
    assert rel_err(mine, ref) < 1e-5


@pytest.mark.parametrize('n,M', [(8, 4), (600, 3), (2048, 32)])
def test_dec_out_backward_sums_by_linearity(device, n, M):
    """cmvae_dec_out_bwd_sums (two contractions over the pixels, no gradient tile) against cmvae_dec_out_bwd (gradient
    tile, mask and sums per row): dW4, the BatchNorm-backward sums of layer 3 and db4 to rounding; BatchNorm shifts
    of either sign and a saturated sigmoid included."""
    from cmvae_amd import decoder
    from cmvae_amd._lib import lib, check
    _p = lambda t: t.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(n)
    y3 = (torch.randn(n * 196, 64, generator=g) * 1.5).to(device)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    mean, invstd = torch.randn(64, generator=g) * 0.3, torch.rand(64, generator=g) + 0.5
    scale = gamma * invstd
    fold = tuple(t.to(device) for t in (scale, beta - mean * scale, mean, invstd))
    w4 = (torch.randn(64, 16, generator=g) * 0.2).to(device)
    p = torch.rand(n, 784, generator=g)
    p[0, :5] = 0.0
    p[1, :5] = 1.0
    p = p.to(device)
    tgt = (torch.rand(n // M, 784, generator=g) < 0.2).float().to(device)
    gl = torch.tensor([0.37], device=device)
    part = torch.empty(lib.cmvae_dec_out_grid(n), lib.cmvae_dec_out_bwd_partial_len(), device=device)
    check(lib.cmvae_dec_out_bwd(_p(y3), *(_p(t) for t in fold), _p(w4), _p(p), _p(tgt), M, n, _p(gl), 1.0, None, _p(part),
                                st), 'cmvae_dec_out_bwd')
    want = decoder.colsum(part)
    L = lib.cmvae_dec_out_bwd_sums_partial_len()
    part2 = torch.empty(lib.cmvae_dec_out_bwd_sums_grid(n), L, device=device)
    tot, got = torch.empty(L, device=device), torch.empty(1153, device=device)
    check(lib.cmvae_dec_out_bwd_sums(_p(y3), *(_p(t) for t in fold), _p(w4), _p(p), _p(tgt), M, n, _p(gl), 1.0, _p(part2),
                                     _p(tot), _p(got), st), 'cmvae_dec_out_bwd_sums')
    for sl in (slice(0, 1024), slice(1024, 1088), slice(1088, 1152), slice(1152, 1153)):
        assert rel_err(got[sl], want[sl]) < 2e-5, sl


@pytest.mark.parametrize('n', [48, 1200])
def test_conv1_backward_in_one_pass_by_linearity(device, n):
    """Encoder layer 1 in training: no activation is stored (batch statistics from a summing pass, pooled output from
    the evaluation kernel with the batch fold) and the backward is ONE pass that recomputes y1 from the images and gets
    the conv1 weight gradient by linearity — sums of dybn, xhat and their products with the centred taps, c1 and c2 put
    in afterwards (cmvae_enc_conv1_bwd_lin_rc, cmvae_enc_conv1_lin_finalize).  Against float64 autograd of the whole
    encoder (omniglot/module.py:12-33): output, and every parameter gradient."""
    from cmvae_amd import encoder
    from cmvae_amd.module import OmniglotEncoder
    torch.manual_seed(5)
    enc = OmniglotEncoder(64).to(device).train()
    x = (torch.rand(n, 784, device=device) < 0.2).float()
    gh = torch.randn(n, 256, device=device)
    h_hip = encoder.encode_images(enc, x)
    h_hip.backward(gh)
    # float64 autograd of the whole encoder on the CPU
    h = x.double().cpu().view(n, 1, 28, 28)
    ref = {}
    for i in range(4):
        cv, bn = enc.encoder[4 * i], enc.encoder[4 * i + 1]
        for name, t in (('%d.weight' % (4 * i), cv.weight), ('%d.bias' % (4 * i), cv.bias),
                        ('%d.weight' % (4 * i + 1), bn.weight), ('%d.bias' % (4 * i + 1), bn.bias)):
            ref[name] = t.detach().double().cpu().requires_grad_(True)
        h = F.conv2d(h, ref['%d.weight' % (4 * i)], ref['%d.bias' % (4 * i)], padding=1)
        h = F.batch_norm(h, None, None, ref['%d.weight' % (4 * i + 1)], ref['%d.bias' % (4 * i + 1)], True, 0.1, bn.eps)
        h = F.max_pool2d(F.relu(h), 2, ceil_mode=True)
    h = h.permute(0, 2, 3, 1).reshape(n, 256)                                   # channels-last
    h.backward(gh.double().cpu())
    assert rel_err(h_hip, h) < 1e-5
    scale = max(float(t.grad.abs().max()) for t in ref.values())
    # (at 1200 images a few ReLU / max-pool decisions of the fp32 forward differ from the float64 one's: every gradient
    # upstream of such a unit moves by about 1 / n)
    tol = 2e-5 if n == 48 else 2e-3
    for name, p in enc.encoder.named_parameters():
        g_ref = ref[name].grad
        if name.endswith('bias') and int(name.split('.')[0]) % 4 == 0:
            # a conv bias in front of a BatchNorm has a zero gradient: both sides hold rounding noise around zero
            assert float(p.grad.abs().max()) < 3e-5 * scale, name
        else:
            assert rel_err(p.grad, g_ref) < tol, (name, rel_err(p.grad, g_ref))


def test_multi_launch_rejects_what_it_cannot_run(device):
    """More than four taps or geometries, geometries over different tensors: an error, not a launch."""
    from cmvae_amd import conv
    n = 16
    x = torch.zeros(n * 49, 64, device=device)
    w = torch.zeros(16, 64, 64, device=device)
    out = torch.zeros(n * 196, 64, device=device)
    phases = [conv.geom_convT_phase(n, 7, 7, 4, py, px)[0] for py in (0, 1) for px in (0, 1)]
    with pytest.raises(RuntimeError, match='geometries'):
        conv.conv_gemm_multi(phases + phases[:1], x, w, out)
    with pytest.raises(RuntimeError, match='taps'):
        conv.conv_gemm_multi([conv.geom_conv3x3(n, 7, 7)], x, w, x.clone())
    dW = torch.zeros(64, 64, 4, 4, device=device)
    with pytest.raises(RuntimeError, match='taps'):
        conv.conv_wgrad_multi([conv.geom_conv3x3(n, 7, 7)], x, x, dW, [list(range(9))], 9, 576)
    other = conv.geom_convT_phase(n, 4, 4, 3, 0, 0)[0]
    with pytest.raises(RuntimeError, match='same tensors'):
        conv.conv_wgrad_multi([phases[0], other], x, out, dW, [[0, 1, 2, 3], [0, 1, 2, 3]], 1024, 16)


@pytest.mark.parametrize('rows,K,N,act', [(300, 128, 256, 2), (1000, 256, 256, 1), (77, 256, 128, 0),
                                          (513, 64, 2048, 0), (1008, 192, 128, 1), (4800, 256, 64, 0), (16, 64, 64, 0)])
def test_linear_fwd_dgrad_wgrad(device, rows, K, N, act):
    from cmvae_amd import conv
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, K, generator=g, requires_grad=True)
    w = (torch.randn(N, K, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(N, generator=g).requires_grad_(True)
    pre = F.linear(x, w, b)
    y = [pre, torch.relu(pre), F.elu(pre)][act]
    gp = torch.randn(pre.shape, generator=g)
    pre.backward(gp)
    xd, wd, bd = x.detach().to(device), w.detach().to(device), b.detach().to(device)
    out = torch.empty(rows, N, device=device)
    conv.conv_gemm(conv.geom_linear(rows, K, N), xd, wd, out, bias=bd, act=act, n_col_blocks=N // 64)
    assert rel_err(out, y) < TOL
    # dx = gp @ W : a Linear with weight W^T [K,N]
    gpd = gp.to(device)
    dx = torch.empty(rows, K, device=device)
    if N // 64 <= 16:
        conv.conv_gemm(conv.geom_linear(rows, N, K), gpd, wd.t().contiguous(), dx, n_col_blocks=K // 64)
        assert rel_err(dx, x.grad) < TOL
    # dW[n][k], db[n]
    dW = torch.zeros(N, K, device=device)
    db = torch.zeros(N, device=device)
    geom = conv.geom_linear(rows, K, N)
    for nb in range(N // 64):
        conv.conv_wgrad(geom, xd, gpd, dW, [nb * 64 * K + 64 * t for t in range(K // 64)], s_ci=1, s_co=K,
                        db=db[nb * 64:], g_pitch=N, g_col=nb * 64)
    assert rel_err(dW, w.grad) < TOL
    assert rel_err(db, b.grad) < TOL


@pytest.mark.parametrize('n,H,W', [(5, 28, 28), (3, 7, 7), (2, 13, 9), (300, 28, 28)])
def test_enc_conv1_pool_eval_kernel(device, n, H, W):
    """Fused Conv2d(1,64,3,1,1) + BatchNorm(eval) + ReLU + MaxPool2d(2, ceil_mode=True) (omniglot/module.py:14-17)
    against the same four torch ops in fp32; odd sizes exercise the clipped pooling window."""
    from cmvae_amd._lib import lib, check
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(n, 1, H, W, generator=g) < 0.3).float()
    w = torch.randn(64, 1, 3, 3, generator=g) * 0.4
    b = torch.randn(64, generator=g) * 0.1
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    rm, rv = torch.randn(64, generator=g) * 0.3, torch.rand(64, generator=g) + 0.3
    ref = F.max_pool2d(F.relu(F.batch_norm(F.conv2d(x, w, b, padding=1), rm, rv, gamma, beta, False, 0.1, 1e-5)), 2,
                       ceil_mode=True)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, 64)                      # channels-last rows
    d = lambda t: t.to(device).contiguous()                            # noqa: E731
    scale, shift, invstd = (torch.empty(64, device=device) for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream
    gm, bt, rmd, rvd = d(gamma), d(beta), d(rm), d(rv)
    check(lib.cmvae_bn_eval_fold(gm.data_ptr(), bt.data_ptr(), rmd.data_ptr(), rvd.data_ptr(), 1e-5, scale.data_ptr(),
                                 shift.data_ptr(), invstd.data_ptr(), st), 'cmvae_bn_eval_fold')
    assert rel_err(invstd, torch.rsqrt(rv + 1e-5)) < 1e-6
    xd, wd, bd = d(x.reshape(n, H * W)), d(w.reshape(64, 9)), d(b)
    out = torch.empty(ref.shape[0], 64, device=device)
    check(lib.cmvae_enc_conv1_pool_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                       n, H, W, out.data_ptr(), st), 'cmvae_enc_conv1_pool_fwd')
    assert rel_err(out, ref) < 1e-5
