"""Can a memory-bound kernel (BatchNorm-backward apply on a 481 MB tensor) hide behind an MFMA-bound one (the four
layer-3 weight-gradient phase launches) when the two run on different streams?  Sequential vs concurrent time."""
import sys, torch
sys.path.insert(0, '.')
from cmvae_amd import conv
from cmvae_amd.decoder import bn_relu_bwd_apply
dev = torch.device('cuda:0')
n, Hi, k = 38400, 7, 4
g = torch.Generator(device=dev).manual_seed(0)
x = torch.empty(n * 49, 64, device=dev).normal_(generator=g)
gy = torch.empty(n * 196, 64, device=dev).normal_(generator=g)
sc = torch.rand(64, device=dev) + 0.5
sh = torch.randn(64, device=dev) * 0.3
dW = torch.zeros(64, 64, k, k, device=dev)
geoms = [conv.geom_convT_phase(n, Hi, Hi, k, py, px, pixel_major=True) for py in (0, 1) for px in (0, 1)]
d_a = torch.randn(n * 49, 64, device=dev); y = torch.randn(n * 49, 64, device=dev); out = torch.empty_like(y)
fold = [torch.rand(64, device=dev) + 0.5 for _ in range(4)]
c1, c2 = torch.randn(64, device=dev), torch.randn(64, device=dev)
side = torch.cuda.Stream()

def wg():
    for geom, tap_idx in geoms:
        conv.conv_wgrad(geom, x, gy, dW, tap_idx, s_ci=64 * k * k, s_co=k * k, a_scale=sc, a_shift=sh)
def bn(reps=4):
    for _ in range(reps):
        bn_relu_bwd_apply(d_a, y, fold, c1, c2, True, out)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        bn()
    wg()
    main.wait_stream(side)
print('wgrad x4 alone   %.3f ms' % timed(wg))
print('bn apply x4 alone %.3f ms' % timed(bn))
print('concurrent        %.3f ms' % timed(both))
