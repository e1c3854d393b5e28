"""GroupNorm forward / backward on the step's shapes: one-kernel (slab in registers) form against the three-launch
general path (PEA_GN_UNFUSED=1 in a second process).  Prints us per call and GB/s on the minimal traffic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops

def timeit(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

print("PEA_GN_UNFUSED =", os.environ.get("PEA_GN_UNFUSED"))
for (B, HW, C) in [(8, 1024, 1280), (8, 1024, 2560), (8, 1024, 1920), (8, 4096, 640), (8, 4096, 1280), (8, 4096, 1920), (8, 4096, 960),
                   (8, 16384, 320), (8, 16384, 640), (4, 1024, 1280), (4, 1024, 2560), (4, 4096, 640), (4, 4096, 1280), (4, 16384, 320), (4, 16384, 640), (4, 4096, 1920)]:
    x = torch.randn(B, HW, C, device="cuda").bfloat16(); dy = torch.randn_like(x)
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    y, stats = ops.groupnorm_fwd(x, gamma, beta, 32, 1e-5, True)
    tf = timeit(lambda: ops.groupnorm_fwd(x, gamma, beta, 32, 1e-5, True))
    acc = torch.zeros_like(x)
    tb = timeit(lambda: ops.groupnorm_bwd(x, dy, gamma, beta, stats, 32, True))
    ta = timeit(lambda: ops.groupnorm_bwd(x, dy, gamma, beta, stats, 32, True, accum_into=acc))
    n = B * HW * C * 2
    print(f"B{B} HW{HW} C{C}: fwd {tf:7.1f} us ({2*n/tf/1e3:6.0f} GB/s) | bwd {tb:7.1f} us ({3*n/tb/1e3:6.0f} GB/s) | bwd+acc {ta:7.1f} us ({4*n/ta/1e3:6.0f} GB/s)", flush=True)
