"""Where does a K-step of the persistent loader/consumer GEMM go?  Timing-only probes (outputs are wrong while a probe is
set): debug bit 1 = no LDS-DMA refills after the prologue, 2 = no barriers, 4 = no fragment ds_reads, 16 = no epilogue."""
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
shapes = [(8192, 10240, 1280, 27), (8192, 10240, 1280, 28), (32768, 5120, 640, 27), (8192, 8192, 8192, 27)]
if len(sys.argv) > 1 and sys.argv[1] == "oneround":      # the one-round 128 x 160 shapes of the backward on the persistent 4 x 2-wave kernel
    shapes = [(4096, 1280, 10240, 28), (4096, 1280, 3840, 28), (4096, 1280, 1280, 28), (4096, 1280, 10240, 29)]
for (M, N, K, v) in shapes:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    L.pea_debug_set_gemm_variant(v)
    line = f"M{M} N{N} K{K} v{v}: "
    for dbg, name in [(0, "full"), (16, "noEpi"), (17, "noEpi+noDMA"), (18, "noEpi+noBarrier"), (19, "noEpi+noDMA+noBarrier"), (23, "MFMA only"), (20, "noEpi+noLDSread")]:
        L.pea_debug_set_gemm_debug(dbg)
        t = min(timeit(lambda: ops.gemm(a, w, out=out)) for _ in range(2))
        line += f"{name} {t:6.1f}us ({2*M*N*K/t/1e6:5.0f} TF) | "
    L.pea_debug_set_gemm_debug(0)
    print(line, flush=True)
L.pea_debug_set_gemm_variant(-1)
