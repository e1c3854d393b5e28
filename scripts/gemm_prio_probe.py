"""Issue priority of the persistent GEMM's DMA waves (s_setprio 3, the default) against equal priorities (debug bit 64):
interleaved rounds, min of 5.  Round-2 measurement with more arms (loaders at 1, MFMA waves 4-7 at 1, all MFMA waves at 1):
all within +-0.5 %; loaders at 3 was +0.3..1.0 % on every shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib
L = lib()
BF = torch.bfloat16
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
shapes = [(8192, 10240, 1280), (8192, 3840, 1280), (8192, 1280, 5120), (8192, 1280, 1280), (32768, 5120, 640), (32768, 640, 2560), (8192, 8192, 8192)]
modes = {"loaders prio3 (default)": 0, "equal priorities": 64}
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    L.pea_debug_set_gemm_variant(27)
    best = {k: 1e9 for k in modes}
    for rnd in range(5):
        for k, dbg in modes.items():
            L.pea_debug_set_gemm_debug(dbg)
            best[k] = min(best[k], timeit(lambda: ops.gemm(a, w)))
    L.pea_debug_set_gemm_debug(0)
    print(f"M{M} N{N} K{K} v27: " + " | ".join(f"{k} {2*M*N*K/t/1e12:7.1f} TF" for k, t in best.items()), flush=True)
L.pea_debug_set_gemm_variant(-1)
