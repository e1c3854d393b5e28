"""self-attention forward / backward timings on the step's two shapes (min of 3 rounds of 20); for A/B runs through scripts/lib_multi.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
BF = torch.bfloat16
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (B, H, Sq, Skv) in [(4, 10, 4096, 4096), (4, 20, 1024, 1024)]:
    C = H * 64
    q = torch.randn(B, Sq, C, device="cuda").to(BF); k = torch.randn(B, Skv, C, device="cuda").to(BF); v = torch.randn(B, Skv, C, device="cuda").to(BF)
    o, lse = ops.attention_fwd(q, k, v, H)
    do = torch.randn_like(o)
    tf = min(timeit(lambda: ops.attention_fwd(q, k, v, H)) for _ in range(3))
    tb = min(timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, H)) for _ in range(3))
    print(f"S{Sq}: fwd {tf:7.1f} us  bwd {tb:7.1f} us", flush=True)
