"""Where does a K-step of the loader/consumer GEMM go?  Timing-only probes (outputs are wrong while a probe is set)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib
L = lib(); BF = torch.bfloat16
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N, K, v) in [(4096, 1280, 5120, 19), (4096, 1280, 5120, 20), (4096, 1280, 1280, 19), (4096, 1280, 1280, 20), (16384, 640, 2560, 19), (16384, 640, 2560, 20)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    L.pea_debug_set_gemm_variant(v)
    line = f"M{M} N{N} K{K} v{v}: "
    for dbg, name in [(0, "full"), (1, "noDMA"), (3, "noDMA+noBarrier"), (7, "MFMA only"), (2, "noBarrier")]:
        L.pea_debug_set_gemm_debug(dbg)
        t = timeit(lambda: ops.gemm(a, w, out=out))
        line += f"{name} {t:6.1f}us ({2*M*N*K/t/1e6:5.0f} TF) | "
    L.pea_debug_set_gemm_debug(0)
    print(line, flush=True)
L.pea_debug_set_gemm_variant(-1)
