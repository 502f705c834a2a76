"""Shader clock during the layer-3 streaming weight gradient (needs a -DCMVAE_ABLATE=20 build in CMVAE_HIP_LIB)."""
import ctypes, runpy, sys
sys.path.insert(0, '.')
sys.argv = ['tools/probe_wgrad.py']
runpy.run_path('tools/probe_wgrad.py')
from cmvae_amd._lib import lib
buf = (ctypes.c_longlong * 2)()
lib.cmvae_debug_clock_probe.restype = ctypes.c_int
assert lib.cmvae_debug_clock_probe(buf) == 0
print('clock64 ticks %d, wall ticks %d (100 MHz) -> %.3f GHz over %.1f us' % (buf[0], buf[1], buf[0] / buf[1] / 10.0, buf[1] / 100.0))
