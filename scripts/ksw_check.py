"""variant 41 (intra-workgroup K split on the 128 x 160 one-tile kernel) against variant 25 and fp32 torch: values and time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd._lib import lib, check, ptr, stream_ptr
L = lib(); BF = torch.bfloat16
def med(fn, iters=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return sorted(ts)[2]
for (M, N, K, res) in [(4096, 1280, 1280, False), (4096, 1280, 1280, True), (4096, 1280, 3840, True), (4096, 1280, 10240, True), (4000, 1280, 2560, True)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(M, K, device="cuda", generator=g).to(BF); w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(BF)
    r = torch.randn(M, N, device="cuda", generator=g).to(BF) if res else None
    bias = torch.randn(N, device="cuda", generator=g)
    ref = a.float() @ w.float().T + bias + (r.float() if res else 0)
    outs, ts = {}, {}
    sp = stream_ptr()
    for v in (25, 41):
        L.pea_debug_set_gemm_variant(v)
        c = torch.zeros(M, N, device="cuda", dtype=BF)
        f = lambda: check(L.pea_op_gemm(ptr(a), K, ptr(w), K, ptr(c), N, M, N, K, 1.0, ptr(bias), None, 0, 1, 0, None, 0, ptr(r), N if res else 0, 0, 0, sp))
        f(); torch.cuda.synchronize()
        outs[v] = c.float().clone()
        ts[v] = med(f)
    L.pea_debug_set_gemm_variant(-1)
    e25 = float((outs[25] - ref).abs().max()); e41 = float((outs[41] - ref).abs().max())
    d = float((outs[41] - outs[25]).abs().max())
    ulp = float(((outs[41] - ref).abs() / (ref.abs() * 2 ** -8 + 1e-3)).max())
    print(f"M{M} N{N} K{K} res={res}: v25 {ts[25]:6.1f} us  v41 {ts[41]:6.1f} us | max err vs fp32 v25 {e25:.3e} v41 {e41:.3e} (worst {ulp:.2f} bf16 ulp), |v41 - v25| {d:.3e}", flush=True)
