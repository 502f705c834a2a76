"""When do the workgroups of the weight-stationary GEMM launches finish?  (needs the -DCMVAE_ABLATE=30 build:
CMVAE_HIP_LIB=tools/bin/libcmvae_abl30.so)  Layer-3 forward, the four sub-pixel phases in one launch, and the two halves
of the layer-3 data gradient: per launch the span, and the share of the span the workgroups spend waiting for the
slowest one."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.')
from cmvae_amd import conv
from cmvae_amd._lib import lib
dev = torch.device('cuda:0')
n, Hi, k = 38400, 7, 4
g = torch.Generator(device=dev).manual_seed(0)
x = torch.empty(n * 49, 64, device=dev).normal_(generator=g)
w = torch.empty(16, 64, 64, device=dev).normal_(generator=g) * 0.05
b = torch.zeros(64, device=dev)
sc = torch.rand(64, device=dev) + 0.5
sh = torch.randn(64, device=dev) * 0.3
out = torch.empty(n * 196, 64, device=dev)
geoms = [conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=True)[0] for py in (0, 1) for px in (0, 1)]
lib.cmvae_debug_wg_times.restype = ctypes.c_int
lib.cmvae_debug_wg_times.argtypes = [ctypes.c_void_p, ctypes.c_int]


def report(tag, nwg):
    torch.cuda.synchronize()
    buf = np.zeros((nwg, 3), dtype=np.uint64)
    assert lib.cmvae_debug_wg_times(buf.ctypes.data, nwg) == 0
    st, en, tiles = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64), buf[:, 2].astype(np.int64)
    t0 = st.min()
    span = (en.max() - t0) / 100.0
    dur = (en - st) / 100.0
    endr = (en - t0) / 100.0
    idle = (span - endr).sum() / (span * nwg)
    print('%s: %d workgroups, span %.1f us; start spread %.1f us; end: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us;'
          ' workgroup-time idle before the launch ends %.1f %%; tiles per workgroup %d..%d; us per tile p10 %.2f median %.2f p90 %.2f'
          % (tag, nwg, span, (st.max() - t0) / 100.0, endr.min(), np.percentile(endr, 10), np.median(endr),
             np.percentile(endr, 90), endr.max(), 100 * idle, tiles.min(), tiles.max(),
             np.percentile(dur / np.maximum(tiles, 1), 10), np.median(dur / np.maximum(tiles, 1)),
             np.percentile(dur / np.maximum(tiles, 1), 90)))
    q = np.zeros((nwg, 4), dtype=np.uint64)
    lib.cmvae_debug_wg_quarters.restype = ctypes.c_int
    lib.cmvae_debug_wg_quarters.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.cmvae_debug_wg_quarters(q.ctypes.data, nwg) == 0
    q = (q.astype(np.int64) - t0) / 100.0
    early = endr < np.median(endr)
    for name, sel in (('workgroups that finish early', early), ('workgroups that finish late', ~early)):
        print('     %s: 1/4 of their tiles at %.0f us, 1/2 at %.0f, 3/4 at %.0f, 7/8 at %.0f, all at %.0f (medians)'
              % (name, np.median(q[sel, 0]), np.median(q[sel, 1]), np.median(q[sel, 2]), np.median(q[sel, 3]),
                 np.median(endr[sel])))
    return st, en, tiles


nwg = conv.gemm_multi_grid()
for _ in range(3):
    conv.conv_gemm_multi(geoms, x, w, out, bias=b, a_scale=sc, a_shift=sh, stats=True)
st, en, tiles = report('layer-3 forward, 4 phases in one launch', nwg)
# per phase (begin[] partitions the grid; recover the partition from the tile counts is not possible: print quarters)
q = nwg // 8
for i in range(8):
    sl = slice(i * q, (i + 1) * q)
    print('   workgroups %4d..%4d: end median %.1f us, tiles %d..%d' % (i * q, (i + 1) * q - 1,
          np.median((en[sl] - st.min()) / 100.0), tiles[sl].min(), tiles[sl].max()))
for ph, geom in enumerate(geoms):
    for _ in range(2):
        conv.conv_gemm(geom, x, w, out, bias=b, a_scale=sc, a_shift=sh, stats=True)
    report('layer-3 forward phase %d alone' % ph, 2 * 256)
# layer-3 data gradient (two accumulating eight-tap launches of the eight-wave kernel: one workgroup per CU)
gd, order = conv.geom_convT_dgrad(n, Hi, Hi, k, pixel_major=True)
dy = torch.empty(n * 196, 64, device=dev).normal_(generator=g)
wtd = torch.empty(16, 64, 64, device=dev).normal_(generator=g) * 0.05
d_in = torch.empty(n * 49, 64, device=dev)
fold = (sc, sh, torch.zeros(64, device=dev), torch.ones(64, device=dev))
for _ in range(2):
    conv.conv_gemm_bnbwd(gd, dy, wtd, d_in, x, fold)
report('layer-3 data gradient, second (accumulating) half', 256)
