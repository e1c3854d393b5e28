"""GEMM with COLD weights (rotating through > 256 MiB of distinct weight matrices, activations produced by the
previous launch) -- the situation inside the UNet tape -- with and without a side-stream Infinity-Cache prefetch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib, ptr
L = lib()
BF = torch.bfloat16
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,4,5".split(","))]

def run(M, N, K, v, prefetch, iters=3):
    nW = max(4, int(600e6 // (N * K * 2)))            # > 2x the Infinity Cache
    Ws = (torch.randn(nW, N, K, device="cuda") * K ** -0.5).to(BF)
    a = torch.randn(M, K, device="cuda").to(BF)
    outs = [torch.empty(M, N, device="cuda", dtype=BF) for _ in range(2)]
    L.pea_debug_set_gemm_variant(v)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    def sweep():
        evs = []
        for i in range(nW):
            if prefetch:
                ev = torch.cuda.Event(); ev.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    j = (i + 1) % nW
                    L.pea_op_prefetch(ctypes.c_void_p(Ws[j].data_ptr()), N * K * 2, ctypes.c_void_p(side.cuda_stream))
            ops.gemm(a, Ws[i], out=outs[i & 1])
    sweep(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): sweep()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) * 1e-3 / (iters * nW)
    return 2.0 * M * N * K / t / 1e12, t * 1e6

for (M, N, K) in [(4096, 1280, 1280), (4096, 1280, 5120), (4096, 10240, 1280), (16384, 640, 2560), (16384, 5120, 640)]:
    line = f"M{M} N{N} K{K}:"
    for v in variants:
        tf0, us0 = run(M, N, K, v, False)
        tf1, us1 = run(M, N, K, v, True)
        line += f" | v{v} cold {tf0:6.1f} TF ({us0:5.1f}us) +pf {tf1:6.1f} TF"
    print(line, flush=True)
L.pea_debug_set_gemm_variant(-1)
