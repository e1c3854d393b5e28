"""timing of the four-wave GEMM (variant 40) on long-K and short-K shapes; for lib_multi.py probe builds (results may be wrong)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pea_diffusion_amd", "libpea_hip.so"))
BF = torch.bfloat16
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
line = ""
for (M, N, K) in [(8192, 8192, 8192), (8192, 10240, 1280), (8192, 3840, 1280), (32768, 5120, 640)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    for v in (40, 41, 27):
        L.pea_debug_set_gemm_variant(v)
        t = timeit(lambda: ops.gemm(a, w, out=out))
        line += f" M{M}N{N}K{K} v{v} {2*M*N*K/t/1e12:6.0f} |"
print(line)
