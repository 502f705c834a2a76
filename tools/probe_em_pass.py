"""Throughput of the fused E/M pass (cmvae_em_pass) at a given shape: GB/s of latent rows read once."""
import sys, torch
sys.path.insert(0, '.')
from cmvae_amd import ops
dev = torch.device('cuda:0')
shapes = [(256, 19200, 5), (4, 9600, 20), (64, 9600, 20), (256, 19200, 16), (32, 38400, 32)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1].split(','))]
for B, N, K in shapes:
    z = torch.randn(B, N, 64, device=dev)
    mu = torch.randn(B, K, 64, device=dev)
    logw = torch.randn(B, K, device=dev)
    ws, _ = ops.em_pass(z, mu, logw)
    for _ in range(3):
        ops.em_pass(z, mu, logw, ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        ops.em_pass(z, mu, logw, ws)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / it
    print('B %4d N %6d K %2d: %8.1f us  %7.1f GB/s' % (B, N, K, ms * 1e3, z.numel() * 4 / ms / 1e6))
