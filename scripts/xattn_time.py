"""cross-attention forward / backward on the step's two shapes (77 keys); backward: the round-3 one-pass kernel (v1) against the
specialised-wave kernel (v2), alternating in one process; min of 3 rounds of 30"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib
BF = torch.bfloat16
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (B, H, Sq, Skv) in [(4, 20, 1024, 77), (4, 10, 4096, 77), (8, 20, 1024, 77), (8, 10, 4096, 77)]:
    C = H * 64
    q = (torch.randn(B, Sq, C, device="cuda") * 0.18).to(BF); k = torch.randn(B, Skv, C, device="cuda").to(BF); v = torch.randn(B, Skv, C, device="cuda").to(BF)
    o, lse = ops.attention_fwd(q, k, v, H, q_prescaled=True)
    do = torch.randn_like(o)
    tf = min(timeit(lambda: ops.attention_fwd(q, k, v, H, q_prescaled=True)) for _ in range(3))
    tb = {0: 1e9, 2: 1e9, 3: 1e9}
    for _ in range(3):
        for ver in (0, 2, 3):
            lib().pea_debug_set_xattn_bwd_v2(ver)
            tb[ver] = min(tb[ver], timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, H, q_prescaled=True)))
    lib().pea_debug_set_xattn_bwd_v2(3)
    mb = 2.0 * B * Sq * C * 4 / 1e6
    print(f"xattn B{B} H{H} Sq{Sq}: fwd {tf:6.1f} us | bwd (+reduce) v1 {tb[0]:6.1f} us, v2 {tb[2]:6.1f} us, v3 {tb[3]:6.1f} us | Q,dO,O,dQ = {mb:.0f} MB -> {mb / 5e3 * 1e3 / 1e3:.1f} us at 5 TB/s", flush=True)
