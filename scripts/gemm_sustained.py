"""Does GEMM throughput drop under sustained load (DVFS)?  Runs one shape continuously and prints TF/s per window."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib
L = lib(); BF = torch.bfloat16
def sustained(M, N, K, v, secs=2.0, label=""):
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    L.pea_debug_set_gemm_variant(v)
    fn = (lambda: ops.gemm(a, w, out=out)) if v >= 0 else (lambda: torch.matmul(a, w.T, out=out))
    fn(); torch.cuda.synchronize()
    res = []
    t_end = time.time() + secs
    while time.time() < t_end:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200): fn()
        e.record(); torch.cuda.synchronize()
        res.append(2.0 * M * N * K * 200 / (s.elapsed_time(e) * 1e-3) / 1e12)
    print(f"{label} M{M} N{N} K{K} v{v}: first {res[0]:.0f} TF, windows: " + " ".join(f"{r:.0f}" for r in res[:: max(1, len(res) // 12)]) + f" last {res[-1]:.0f}", flush=True)
sustained(4096, 1280, 1280, 19, label="ours")
sustained(4096, 1280, 1280, -1, label="hipblaslt")
sustained(4096, 1280, 5120, 19, label="ours")
sustained(4096, 10240, 1280, 10, label="ours")
sustained(8192, 8192, 8192, 12, secs=3.0, label="ours")
sustained(8192, 8192, 8192, -1, secs=3.0, label="hipblaslt")
L.pea_debug_set_gemm_variant(-1)
os.system("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | head -4")
