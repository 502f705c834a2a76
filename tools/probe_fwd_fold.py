"""Layer-3 forward phases (38400 images, pixel-major): with the BatchNorm + ReLU fold in the operand load and without."""
import sys, torch
sys.path.insert(0, '.')
from cmvae_amd import conv
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
for fold in (True, False, True, False):
    def run():
        for geom in geoms:
            conv.conv_gemm(geom, x, w, out, bias=b, a_scale=sc if fold else None, a_shift=sh if fold else None, stats=True)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); e1.synchronize()
    print('fold=%s: %.1f us per phase launch' % (fold, e0.elapsed_time(e1) / 20 * 1e3))
