"""Tile variants on the stacked cross-attention K|V projection (all 70 blocks' to_k / to_v in one launch:
M = 2B*77 rows, N = 166400, K = 2048) -- the tall-skinny shape the variant rule sends to the 64x160 tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops

L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pea_diffusion_amd", "libpea_hip.so"))
BF = torch.bfloat16

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

for (M, N, K) in [(616, 166400, 2048), (308, 166400, 2048), (1232, 166400, 2048), (616, 2560, 2048)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    ref = a @ w.T
    line = f"M{M} N{N} K{K}: torch {timeit(lambda: a @ w.T)*1e6:7.1f} us |"
    for v in (31, 29, 28, 27, 25, 24, 22):
        L.pea_debug_set_gemm_variant(v)
        out = ops.gemm(a, w)
        err = (out.float() - ref.float()).abs().max().item()
        line += f" v{v} {timeit(lambda: ops.gemm(a, w))*1e6:7.1f}{'' if err < 0.5 else ' ERR%.2g' % err}"
    print(line, flush=True)
L.pea_debug_set_gemm_variant(-1)
