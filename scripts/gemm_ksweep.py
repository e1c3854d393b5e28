"""Fixed cost vs per-k-iteration cost of the one-wave GEMMs (M=4096 rows: 256 tiles of 128x160 on 256 CUs)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pea_diffusion_amd", "libpea_hip.so"))
BF = torch.bfloat16
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "22,23,25,26".split(","))]
def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N) in [(4096, 1280), (4096, 3840), (16384, 640), (2048, 1280), (1024, 1280)]:
    for K in [64, 320, 640, 1280, 2560, 5120]:
        a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
        t0 = timeit(lambda: a @ w.T)
        line = f"M{M} N{N} K{K:5d}: torch {t0:7.1f} us |"
        for v in variants:
            L.pea_debug_set_gemm_variant(v)
            t = timeit(lambda: ops.gemm(a, w))
            line += f" v{v} {t:7.1f} us ({2*M*N*K/t/1e6:6.1f} TF)"
        print(line, flush=True)
L.pea_debug_set_gemm_variant(-1)
