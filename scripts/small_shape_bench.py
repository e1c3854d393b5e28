import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib
L = lib(); BF = torch.bfloat16
def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
# hot (same operands back to back) and "cold" (operands rotated through 24 buffer sets = more than the Infinity Cache holds)
for (M, N, K) in [(4096, 1280, 1280), (8192, 1280, 1280), (16384, 640, 640)]:
    sets = [(torch.randn(M, K, device="cuda").to(BF), (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)) for _ in range(24)]
    outs = [torch.empty(M, N, device="cuda", dtype=BF) for _ in range(24)]
    filler = torch.empty(300 << 20, device="cuda", dtype=torch.uint8)
    line = f"M{M} N{N} K{K}:"
    for v in (25, 24, 28, 29, 31, 36):
        L.pea_debug_set_gemm_variant(v)
        a, w = sets[0]
        th = timeit(lambda: ops.gemm(a, w))
        i = [0]
        def cold():
            i[0] = (i[0] + 1) % 24
            ops.gemm(sets[i[0]][0], sets[i[0]][1])
        tc = timeit(cold, 48)
        line += f"  v{v} hot {th*1e6:5.1f} / rotating {tc*1e6:5.1f} us"
    print(line, flush=True)
L.pea_debug_set_gemm_variant(-1)
