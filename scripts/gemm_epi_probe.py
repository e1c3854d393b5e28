"""How much of a short-K persistent GEMM is its epilogue?  (debug bit 16 skips the epilogue of gemm_lcp_kernel)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pea_diffusion_amd", "libpea_hip.so"))
BF = torch.bfloat16
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N, K, v) in [(4096, 10240, 1280, 27), (4096, 3840, 1280, 28), (4096, 3840, 1280, 35), (16384, 5120, 640, 27), (16384, 5120, 640, 35), (16384, 1920, 640, 35), (16384, 1920, 640, 28), (8192, 8192, 8192, 27)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    L.pea_debug_set_gemm_variant(v)
    out = []
    for dbg in (0, 64 if v == 35 else 16):
        L.pea_debug_set_gemm_debug(dbg)
        out.append(timeit(lambda: ops.gemm(a, w)))
    L.pea_debug_set_gemm_debug(0)
    print(f"M{M} N{N} K{K} v{v}: {out[0]:7.1f} us ({2*M*N*K/out[0]/1e6:6.1f} TF)  without epilogue (v35: paired stores suppressed) {out[1]:7.1f} us ({2*M*N*K/out[1]/1e6:6.1f} TF)")
L.pea_debug_set_gemm_variant(-1)
