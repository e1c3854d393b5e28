"""What the GEGLU epilogue of the FF projection costs: M8192 N10240 K1280 (32x32 level, 2B = 8 samples) and M32768 N5120 K640,
plain GEMM against GEGLU without stash, with the raw stash and with the backward-factor stash on the student half."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib
from pea_diffusion_amd.ops import ptr, stream_ptr, check
BF = torch.bfloat16

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for (M, N, K) in [(8192, 10240, 1280), (32768, 5120, 640)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    bias = torch.randn(N, device="cuda")
    y = torch.empty(M, N // 2, device="cuda", dtype=BF); st = torch.empty(M, N, device="cuda", dtype=BF)
    c = torch.empty(M, N, device="cuda", dtype=BF)
    fl = 2.0 * M * N * K
    def geglu(stash, grad, rows):
        check(lib().pea_op_gemm_geglu(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(y), ptr(st) if stash else None, M, N, K, grad, rows, stream_ptr()))
    t = timeit(lambda: ops.gemm(a, w, bias=bias, out=c) if "out" in ops.gemm.__code__.co_varnames else ops.gemm(a, w, bias=bias))
    print(f"M{M} N{N} K{K}: plain + bias (full-width bf16 output) {t:7.1f} us {fl/t/1e6:7.0f} TF")
    for name, args in (("GEGLU, no stash", (False, 0, 0)), ("GEGLU + raw stash, student half", (True, 0, M // 2)), ("GEGLU + factor stash, student half", (True, 1, M // 2)),
                       ("GEGLU + factor stash, all rows", (True, 1, 0))):
        t = timeit(lambda: geglu(*args))
        print(f"   {name:36s} {t:7.1f} us {fl/t/1e6:7.0f} TF", flush=True)
