"""How long do the waves of the layer-3 weight gradient wait for the rows they requested one chunk earlier?
(needs the -DCMVAE_ABLATE=73 build: CMVAE_HIP_LIB=tools/bin/libcmvae_w73.so; the kernel stamps the 100 MHz wall clock
around an explicit wait at the top of every chunk)"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.')
from cmvae_amd import conv
from cmvae_amd._lib import lib
dev = torch.device('cuda:0')
n, Hi, k = 38400, 7, 4
g = torch.Generator(device=dev).manual_seed(0)
x = torch.empty(n * 49, 64, device=dev).normal_(generator=g)
gy = torch.empty(n * 196, 64, device=dev).normal_(generator=g)
sc = torch.rand(64, device=dev) + 0.5
sh = torch.randn(64, device=dev) * 0.3
phases = [conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=True) for py in (0, 1) for px in (0, 1)]
dW = torch.zeros(64, 64, k, k, device=dev)
db = torch.zeros(64, device=dev)
lib.cmvae_debug_wg_times.restype = ctypes.c_int
lib.cmvae_debug_wg_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
for it in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    conv.conv_wgrad_multi([ge for ge, _ in phases], x, gy, dW, [t for _, t in phases], s_ci=64 * k * k, s_co=k * k, db=db,
                          a_scale=sc, a_shift=sh)
    e1.record()
    torch.cuda.synchronize()
print('launch + reduce %.1f us' % (e0.elapsed_time(e1) * 1e3))
nw = 1024
buf = np.zeros((nw, 3), dtype=np.uint64)
assert lib.cmvae_debug_wg_times(buf.ctypes.data, nw) == 0
tot, cyc = buf[:, 0].astype(np.float64) / 100.0, buf[:, 1].astype(np.float64)       # us, shader cycles
loop = (buf[:, 2] >> np.uint64(20)).astype(np.float64) / 100.0
chunks = (buf[:, 2] & np.uint64((1 << 20) - 1)).astype(np.float64)
ok = chunks > 0
print('%d waves; chunks per wave %d..%d; loop time per wave: median %.1f us (min %.1f, max %.1f)'
      % (ok.sum(), chunks[ok].min(), chunks[ok].max(), np.median(loop[ok]), loop[ok].min(), loop[ok].max()))
print('waiting at the top of a chunk: per wave total median %.1f us = %.1f %% of its loop (p10 %.1f %%, p90 %.1f %%)'
      % (np.median(tot[ok]), 100 * np.median(tot[ok] / loop[ok]), 100 * np.percentile(tot[ok] / loop[ok], 10),
         100 * np.percentile(tot[ok] / loop[ok], 90)))
print('per chunk: loop %.2f us = %.0f shader cycles, of which waiting %.2f us; shader clock in the loop %.2f GHz (median over waves; '
      'min %.2f, max %.2f)' % ((loop[ok] / chunks[ok]).mean(), (cyc[ok] / chunks[ok]).mean(), (tot[ok] / chunks[ok]).mean(),
                               np.median(cyc[ok] / loop[ok]) / 1e3, (cyc[ok] / loop[ok]).min() / 1e3, (cyc[ok] / loop[ok]).max() / 1e3))
