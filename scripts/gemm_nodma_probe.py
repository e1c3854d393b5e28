"""Is the main loop of the 16x16 loader/consumer GEMM limited by the operand stream (L2 -> LDS) or by the MFMA/LDS-read
loop?  debug bit 1 makes the loader waves skip every LDS-DMA refill after the prologue (results are wrong)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib
L = lib(); BF = torch.bfloat16
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N, K) in [(8192, 8192, 8192), (4096, 10240, 1280), (4096, 1280, 5120), (16384, 640, 2560)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    line = f"M{M} N{N} K{K}: "
    for v in (24, 25, 22):
        L.pea_debug_set_gemm_variant(v)
        for dbg, name in [(0, "full"), (1, "noDMA")]:
            L.pea_debug_set_gemm_debug(dbg)
            t = timeit(lambda: ops.gemm(a, w, out=out))
            line += f"v{v} {name} {2*M*N*K/t/1e6:5.0f} TF | "
        L.pea_debug_set_gemm_debug(0)
    print(line, flush=True)
L.pea_debug_set_gemm_variant(-1)
