"""LayerNorm backward with its saved input x in the Infinity Cache (hot) against x from HBM (cold: rotated through > 256 MiB),
dy always fresh -- would a prefetch of x pay?  Output buffers are preallocated (no allocator in the timed loop)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd._lib import lib, check, ptr, stream_ptr
L = lib()
def med(fn, n, iters=40):
    for i in range(4): fn(i % n)
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(iters): fn((r * iters + i) % n)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return sorted(ts)[2]
for (R, C) in [(4096, 1280), (8192, 1280), (16384, 640)]:
    n = int(400e6 / (R * C * 2)) + 2
    xs = [torch.randn(R, C, device="cuda").bfloat16() for _ in range(n)]
    dy = torch.randn(R, C, device="cuda").bfloat16(); add = torch.randn(R, C, device="cuda").bfloat16()
    dx = torch.empty(R, C, device="cuda", dtype=torch.bfloat16); y = torch.empty_like(dx)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); st = torch.empty(R, 2, device="cuda")
    check(L.pea_op_layernorm_fwd(ptr(xs[0]), ptr(g), ptr(b), ptr(y), ptr(st), R, C, 1e-5, stream_ptr()))
    bw = lambda i: check(L.pea_op_layernorm_bwd(ptr(xs[i]), ptr(dy), ptr(g), ptr(st), ptr(dx), None, None, R, C, 0, stream_ptr()))
    fw = lambda i: check(L.pea_op_layernorm_fwd(ptr(xs[i]), ptr(g), ptr(b), ptr(y), ptr(st), R, C, 1e-5, stream_ptr()))
    print(f"LN {R}x{C} ({n} sets): bwd hot {med(lambda i: bw(0), n):5.1f} us / cold {med(bw, n):5.1f} us | fwd hot {med(lambda i: fw(0), n):5.1f} / cold {med(fw, n):5.1f}", flush=True)
