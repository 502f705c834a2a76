"""Host side of the fp32-MFMA gather-GEMM family (csrc/conv_gemm.hip): geometry builders for
conv 3x3, the sub-pixel phases of the stride-2 transposed convs, their data gradients and
nn.Linear, plus thin wrappers over cmvae_conv_gemm / cmvae_conv_wgrad.

All activations are channels-last matrices [pixels, C] (C = 64 in the conv stacks).
"""
import ctypes
import functools

import torch

from . import _workspace
from ._lib import lib, check, ConvGeom

_p = lambda t: None if t is None else t.data_ptr()  # noqa: E731


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _geom(rows, Hg, Wg, Hi, Wi, sy, sx, Ho, Wo, oys, oxs, oy0, ox0, taps, a_pitch, o_pitch, w_pitch, w_tap_stride,
          w_nblk_stride=0, a_cols=None, w_taps=None, pixel_major=False):
    """w_taps: index of each tap's 64x64 block in the weight tensor (default 0..ntaps-1).
    pixel_major: rows enumerate (y, x, image) instead of (image, y, x): a 128-row tile then sits at one grid
    position and the kernels skip the taps that fall outside the image for it (include/cmvae_abi.h)."""
    g = ConvGeom()
    g.pixel_major = int(bool(pixel_major))
    g.rows_total, g.Hg, g.Wg, g.Hi, g.Wi, g.sy, g.sx = rows, Hg, Wg, Hi, Wi, sy, sx
    g.Ho, g.Wo, g.oys, g.oxs, g.oy0, g.ox0 = Ho, Wo, oys, oxs, oy0, ox0
    g.ntaps = len(taps)
    g.a_pitch, g.o_pitch, g.w_pitch = a_pitch, o_pitch, w_pitch
    g.w_nblk_stride = w_nblk_stride
    for t, (dy, dx) in enumerate(taps):
        g.dy[t], g.dx[t] = dy, dx
        g.a_col[t] = 0 if a_cols is None else a_cols[t]
        g.w_off[t] = (t if w_taps is None else w_taps[t]) * w_tap_stride
    return g


@functools.lru_cache(maxsize=None)
def geom_conv3x3(n, H, W, transpose=False, C=64):
    """3x3 / stride 1 / pad 1 conv over [n,H,W,C]; transpose=True gives the data-gradient taps."""
    taps = [((1 - ky) if transpose else (ky - 1), (1 - kx) if transpose else (kx - 1))
            for ky in range(3) for kx in range(3)]
    return _geom(n * H * W, H, W, H, W, 1, 1, H, W, 1, 1, 0, 0, taps, C, C, C, C * C)


def convT_out(h, k):
    return (h - 1) * 2 - 2 + k


def convT_phase_taps(k, py, px):
    """Kernel taps (ky,kx) and source offsets (dy,dx) feeding output pixels of parity (py,px)."""
    out = []
    for ky in range(k):
        if (py + 1 - ky) % 2:
            continue
        for kx in range(k):
            if (px + 1 - kx) % 2:
                continue
            out.append((ky, kx, (py + 1 - ky) // 2, (px + 1 - kx) // 2))
    return out


@functools.lru_cache(maxsize=None)
def geom_convT_phase(n, Hi, Wi, k, py, px, C=64, pixel_major=False):
    """One sub-pixel phase of ConvTranspose2d(k, stride 2, pad 1): [n,Hi,Wi,C] -> [n,Ho,Wo,C].
    Returns (geometry, tap_idx): the geometry's weight offsets address the FULL kernel [k*k][co][ci], tap t of
    the phase being kernel tap tap_idx[t] = ky*k + kx."""
    Ho, Wo = convT_out(Hi, k), convT_out(Wi, k)
    Hg, Wg = (Ho - py + 1) // 2, (Wo - px + 1) // 2
    taps = convT_phase_taps(k, py, px)
    tap_idx = [t[0] * k + t[1] for t in taps]
    g = _geom(n * Hg * Wg, Hg, Wg, Hi, Wi, 1, 1, Ho, Wo, 2, 2, py, px, [(t[2], t[3]) for t in taps], C, C, C, C * C,
              w_taps=tap_idx, pixel_major=pixel_major)
    return g, tap_idx


@functools.lru_cache(maxsize=None)
def geom_convT_dgrad(n, Hi, Wi, k, C=64, pixel_major=False):
    """Data gradient of the transposed conv = stride-2 conv over d_out [n,Ho,Wo,C] -> [n,Hi,Wi,C].
    Returns (geometry, order): tap t of the geometry is kernel tap order[t] = ky*k + kx; the weight offsets
    address the full kernel [k*k][ci][co] in its natural order."""
    Ho, Wo = convT_out(Hi, k), convT_out(Wi, k)
    # taps kx and kx+2 of one kernel row read the same d_out pixels one output column apart: running them
    # back to back turns half of the operand re-reads into L2 hits
    kxs = [kx for par in (0, 1) for kx in range(par, k, 2)]
    order = [ky * k + kx for ky in range(k) for kx in kxs]
    taps = [(t // k - 1, t % k - 1) for t in order]
    return _geom(n * Hi * Wi, Hi, Wi, Ho, Wo, 2, 2, Hi, Wi, 1, 1, 0, 0, taps, C, C, C, C * C, w_taps=order,
                 pixel_major=pixel_major), order


@functools.lru_cache(maxsize=None)
def geom_linear(rows, K, N, a_pitch=None, o_pitch=None):
    """y = x @ W^T with W [N,K] (torch layout); K, N multiples of 64, K <= 2048."""
    assert K % 64 == 0 and N % 64 == 0 and K // 64 <= 32
    taps = [(0, 0)] * (K // 64)
    return _geom(rows, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, taps, a_pitch or K, o_pitch or N, K, 64, 64 * K,
                 a_cols=[64 * t for t in range(K // 64)])


def gemm_grid(rows, ncb=1):
    return lib.cmvae_conv_gemm_grid(rows, ncb)


def conv_gemm(g, A, Wt, out, bias=None, a_scale=None, a_shift=None, act=0, n_col_blocks=1, stats=False):
    """Launch the forward gather-GEMM.  Returns per-block partial statistics [grid,128] if asked."""
    part = None
    if isinstance(stats, torch.Tensor):
        part = stats                     # caller-provided [gemm_grid(rows,1),128] slice
    elif stats:
        part = torch.empty(gemm_grid(g.rows_total, 1), 128, device=A.device, dtype=torch.float32)
    check(lib.cmvae_conv_gemm(ctypes.byref(g), _p(A), _p(Wt), _p(bias), _p(a_scale), _p(a_shift), act, n_col_blocks,
                              _p(out), _p(part), _stream()), 'cmvae_conv_gemm')
    return part


def conv_gemm_bnbwd(g, A, Wt, out, x, fold):
    """out = conv_gemm(g, A, Wt) (a data gradient at relu(bn(x))) and, from the same launch's epilogue, the partial
    BatchNorm-backward sums [grid,128] = {sum g, sum g*xhat} of the layer below; fold = (scale, shift, mean, invstd)."""
    part = torch.empty(gemm_grid(max(g.rows_total, 1), 1), 128, device=A.device, dtype=torch.float32)
    check(lib.cmvae_conv_gemm_bnbwd(ctypes.byref(g), _p(A), _p(Wt), _p(out), _p(x), _p(fold[0]), _p(fold[1]),
                                    _p(fold[2]), _p(fold[3]), _p(part), _stream()), 'cmvae_conv_gemm_bnbwd')
    return part


@functools.lru_cache(maxsize=None)
def gemm_multi_grid():
    return lib.cmvae_conv_gemm_multi_grid()


def conv_gemm_multi(geoms, A, Wt, out, bias=None, a_scale=None, a_shift=None, stats=False):
    """Up to four geometries (the sub-pixel phases of a transposed conv) over the same tensors in one launch.
    Returns the partial statistics [gemm_multi_grid(),128] over all geometries' rows if asked."""
    arr = (ConvGeom * len(geoms))(*geoms)
    part = torch.empty(gemm_multi_grid(), 128, device=A.device, dtype=torch.float32) if stats else None
    check(lib.cmvae_conv_gemm_multi(arr, len(geoms), _p(A), _p(Wt), _p(bias), _p(a_scale), _p(a_shift), _p(out),
                                    _p(part), _stream()), 'cmvae_conv_gemm_multi')
    return part


@functools.lru_cache(maxsize=None)
def bf16_pipe_min_rows():
    return lib.cmvae_conv_bf16_pipe_min_rows()


def on_bf16_pipe(kind, geoms):
    """Whether a launch runs its fp32 products on the bf16 matrix pipe (three-piece split, six products) — the rules
    of csrc/conv_gemm.hip restated for measurement code (bench.py groups its roofline by pipe) and tests.
    kind: 'gemm' (conv_gemm with act = 0 and one column block), 'multi' (conv_gemm_multi), 'bnbwd' (conv_gemm_bnbwd),
    'wgrad_multi' (conv_wgrad_multi: four-tap phases)."""
    gs = list(geoms)
    g0 = gs[0]
    npix = g0.Hg * g0.Wg
    pm = bool(g0.pixel_major) and npix > 0 and g0.rows_total // npix >= 64
    rows = sum(g.rows_total for g in gs)
    if kind == 'gemm':          # conv_gemm (act 0, one column block): 7 .. 9 taps in two passes, either row order
        return 6 < g0.ntaps <= 9 and rows >= bf16_pipe_min_rows()
    if kind == 'multi':
        return pm and rows >= bf16_pipe_min_rows()
    if kind == 'bnbwd':
        return pm and rows >= bf16_pipe_min_rows() and g0.ntaps <= 18
    if kind == 'wgrad_multi':
        def rows4(g):       # four taps: two operand rows x {unshifted, shifted by one column}
            if g.ntaps != 4 or g.sx != 1 or g.Wg != g.Wi:
                return False
            for t in (0, 2):
                d = sorted((g.dx[t], g.dx[t + 1]), key=abs)
                if g.dy[t] != g.dy[t + 1] or g.a_col[t] != g.a_col[t + 1] or d[0] != 0 or abs(d[1]) != 1:
                    return False
            return True
        return npix > 0 and (g0.rows_total // npix) % 32 == 0 and rows >= bf16_pipe_min_rows() and all(rows4(g) for g in gs)
    raise ValueError(kind)


def bf16_pipe_passes(kind, g):
    """Kernel launches behind one bf16-pipe call of `kind` ('gemm' / 'bnbwd', see on_bf16_pipe): at most six taps of
    weight planes fit LDS, so more taps run as accumulating passes — cut at the boundaries of the stride-2 parity
    classes of the taps where a data gradient has them (csrc/conv_gemm.hip, cmvae_conv_gemm_bnbwd)."""
    if g.ntaps <= 6:
        return 1
    if kind == 'bnbwd' and g.sy == 2 and g.sx == 2:
        cnt = [0, 0, 0, 0]
        for t in range(g.ntaps):
            cnt[(g.dy[t] & 1) * 2 + (g.dx[t] & 1)] += 1
        if max(cnt) <= 6:
            passes, in_pass = 1, 0
            for c in (3, 2, 1, 0):
                if cnt[c] == 0:
                    continue
                if in_pass + cnt[c] > 6:
                    passes, in_pass = passes + 1, 0
                in_pass += cnt[c]
            return passes
    return -(-g.ntaps // 6)


def _wgrad_ws(rows, device):
    return _workspace.get(lib.cmvae_conv_wgrad_workspace_bytes(rows), device, 'conv_wgrad')


def conv_wgrad(g, A, G, dW, tap_off, s_ci, s_co, db=None, accumulate_db=False, a_scale=None, a_shift=None,
               g_pitch=64, g_col=0):
    ws = _wgrad_ws(g.rows_total, A.device)
    offs = (ctypes.c_int64 * len(tap_off))(*tap_off)
    check(lib.cmvae_conv_wgrad(ctypes.byref(g), _p(A), _p(a_scale), _p(a_shift), _p(G), g_pitch, g_col, offs,
                               s_ci, s_co, _p(dW), _p(db), int(accumulate_db), _p(ws), ws.numel(), _stream()),
          'cmvae_conv_wgrad')


def conv_wgrad_multi(geoms, A, G, dW, tap_offs, s_ci, s_co, db=None, a_scale=None, a_shift=None, g_pitch=64):
    """Weight (and bias) gradient of up to four geometries over the same tensors — the sub-pixel phases of a
    transposed conv — in one launch + one reduction.  tap_offs: one list of dW tap offsets per geometry."""
    ws = _wgrad_ws(geoms[0].rows_total, A.device)
    arr = (ConvGeom * len(geoms))(*geoms)
    flat = [t for offs in tap_offs for t in offs]
    offs = (ctypes.c_int64 * len(flat))(*flat)
    check(lib.cmvae_conv_wgrad_multi(arr, len(geoms), _p(A), _p(a_scale), _p(a_shift), _p(G), g_pitch, offs, s_ci,
                                     s_co, _p(dW), _p(db), _p(ws), ws.numel(), _stream()), 'cmvae_conv_wgrad_multi')
