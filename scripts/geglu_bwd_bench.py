"""FF-out dgrad with the fused GEGLU backward on the step's two shapes, operands rotated through more buffers than the
Infinity Cache holds (in situ the stash was written a whole forward pass earlier)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
BF = torch.bfloat16
for (M, N, K, NB) in [(4096, 5120, 1280, 8), (16384, 2560, 640, 4)]:
    a = [torch.randn(M, K, device="cuda").to(BF) for _ in range(NB)]
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    pre = [torch.randn(M, 2 * N, device="cuda").to(BF) for _ in range(NB)]
    import ctypes
    L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pea_diffusion_amd", "libpea_hip.so"))
    for form, var in ((1, -1), (1, 28)):
        L.pea_debug_set_gemm_variant(var)
        for i in range(NB): ops.gemm_geglu_bwd(a[i], w, pre[i], form)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        s.record()
        for r in range(reps):
            for i in range(NB): ops.gemm_geglu_bwd(a[i], w, pre[i], form)
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / (reps * NB) * 1e-3
        print(f"geglu-bwd gemm M{M} N{N} K{K} form{form} variant {var}: {t*1e6:7.1f} us  {2*M*N*K/t/1e12:7.1f} TF", flush=True)
    L.pea_debug_set_gemm_variant(-1)
    ref = ops.gemm(a[0], w)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for r in range(20): ops.gemm(a[r % NB], w)
    t1.record(); torch.cuda.synchronize()
    t = t0.elapsed_time(t1) / 20 * 1e-3
    print(f"   plain gemm same shape: {t*1e6:7.1f} us  {2*M*N*K/t/1e12:7.1f} TF", flush=True)
