"""The 64x64-level (K = 640) GEMMs of the step with their real epilogues, per tile form: does the two-workgroups-per-CU
persistent form (36: 128 x 160, 2 x 2 consumer waves, out of phase) pay where a tile's epilogue is as long as its 10 K-steps?
Operands rotate through NB buffers (cold-ish).  usage: python scripts/k640_bench.py [variants]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib
BF = torch.bfloat16
L = lib()
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "-1,27,28,29,36".split(","))]
NB = 3
def bench(fn, n=NB, reps=6):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for r in range(reps):
        for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * n) * 1e-3
cases = [("geglu-fwd+stash", 32768, 5120, 640), ("plain+res", 32768, 640, 640), ("qkv qscale", 32768, 1920, 640), ("plain", 16384, 640, 640),
         ("ff-out+res", 32768, 640, 2560), ("plain", 16384, 1920, 640), ("geglu-fwd+stash K1280", 8192, 10240, 1280), ("plain+res K1280", 8192, 1280, 1280)]
for (kind, M, N, K) in cases:
    a = [torch.randn(M, K, device="cuda").to(BF) for _ in range(NB)]
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    bias = torch.randn(N, device="cuda")
    res = [torch.randn(M, N, device="cuda").to(BF) for _ in range(NB)] if "res" in kind else None
    line = f"{kind:22s} M{M} N{N} K{K}:"
    for v in variants:
        L.pea_debug_set_gemm_variant(v)
        try:
            if kind.startswith("geglu"):
                y = torch.empty(M, N // 2, device="cuda", dtype=BF); st = torch.empty(M, N, device="cuda", dtype=BF)
                from pea_diffusion_amd._lib import check, ptr, stream_ptr
                f = lambda i: check(L.pea_op_gemm_geglu(ptr(a[i]), K, ptr(w), K, ptr(bias), ptr(y), ptr(st), M, N, K, 1, M // 2, stream_ptr()))
            elif kind.startswith("qkv"):
                f = lambda i: ops.gemm_qscale(a[i], w, bias, qscale_cols=N // 3, qscale=0.18)
            elif res is not None:
                out = torch.empty(M, N, device="cuda", dtype=BF)
                f = lambda i: ops.gemm(a[i], w, bias=bias, res=res[i], out=out)
            else:
                out = torch.empty(M, N, device="cuda", dtype=BF)
                f = lambda i: ops.gemm(a[i], w, out=out)
            t = bench(f)
            line += f"  v{v} {t*1e6:6.1f}us {2*M*N*K/t/1e12:5.0f}TF"
        except Exception as ex:
            line += f"  v{v} FAIL({str(ex)[:40]})"
    L.pea_debug_set_gemm_variant(-1)
    print(line, flush=True)
