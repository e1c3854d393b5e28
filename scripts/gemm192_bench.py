import ctypes, os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib
L = lib()
BF = torch.bfloat16
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for (M, N, K) in [(6144, 1280, 1280), (6144, 1280, 5120), (6144, 3840, 1280), (6144, 10240, 1280), (5120, 1280, 1280), (3072, 1280, 1280), (3072, 5120, 1280), (24576, 640, 640), (24576, 5120, 640)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    line = f"M{M} N{N} K{K}:"
    for v in (-1, 24, 25, 27, 28, 39, 40):
        L.pea_debug_set_gemm_variant(v)
        t = timeit(lambda: ops.gemm(a, w))
        line += f"  v{v} {t*1e6:7.1f}us {2*M*N*K/t/1e12:6.0f}TF"
    print(line, flush=True)
L.pea_debug_set_gemm_variant(-1)
