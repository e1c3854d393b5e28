"""LayerNorm forward / backward on the step's shapes (us per call, GB/s on the minimal traffic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
def timeit(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (R, C) in [(8192, 1280), (4096, 1280), (32768, 640), (16384, 640)]:
    xs = [torch.randn(R, C, device="cuda").bfloat16() for _ in range(6)]; dy = torch.randn(R, C, device="cuda").bfloat16()
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    y, st = ops.layernorm_fwd(xs[0], g, b)
    i = [0]
    def f():
        i[0] = (i[0] + 1) % 6
        ops.layernorm_fwd(xs[i[0]], g, b)
    def bw():
        i[0] = (i[0] + 1) % 6
        ops.layernorm_bwd(xs[i[0]], dy, g, st)
    tf, tb = timeit(f), timeit(bw)
    n = R * C * 2
    print(f"LN {R}x{C}: fwd {tf:6.1f} us ({2*n/tf/1e3:5.0f} GB/s) | bwd {tb:6.1f} us ({3*n/tb/1e3:5.0f} GB/s)", flush=True)
