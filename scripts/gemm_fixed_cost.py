"""Fixed cost of a one-round GEMM launch: time against K for M4096 N1280 (128 x 160 one-tile kernel) and M8192 N1280 (256 x 160),
operands hot, output preallocated (direct C-ABI calls: the host side stays under the kernel time)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd._lib import lib, check, ptr, stream_ptr
L = lib(); BF = torch.bfloat16
def med(fn, iters=200):
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
for (M, N) in [(4096, 1280), (8192, 1280)]:
    line = f"M{M} N{N}:"
    for K in [64, 128, 256, 512, 1280, 2560]:
        a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
        c = torch.empty(M, N, device="cuda", dtype=BF)
        sp = stream_ptr()
        f = lambda: L.pea_op_gemm(ptr(a), K, ptr(w), K, ptr(c), N, M, N, K, 1.0, None, None, 0, 1, 0, None, 0, None, 0, 0, 0, sp)
        line += f"  K{K} {med(f):5.1f} us"
    print(line, flush=True)
# host-side floor of this loop: an empty-ish kernel
x = torch.zeros(64, device="cuda")
print(f"host floor (torch add on 64 floats): {med(lambda: x.add_(1.0)):5.1f} us")
