"""Does the row pitch of Q / K / V matter?  Self-attention forward + backward on separate tensors (pitch C), on column
slices of one fused [B, S, 3C] buffer (the step's layout: pitch 3C) and on slices of padded buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
import ctypes
from pea_diffusion_amd._lib import lib, check, stream_ptr
P = lambda t: ctypes.c_void_p(t.data_ptr())
BF = torch.bfloat16
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for (B, H, S) in [(8, 10, 4096), (8, 20, 1024), (4, 10, 4096), (4, 20, 1024)]:
    C = H * 64
    line = f"B{B} H{H} S{S}:"
    for name, pitch in [("separate", C), ("fused 3C", 3 * C), ("3C+64", 3 * C + 64), ("3C+128", 3 * C + 128), ("4C", 4 * C)]:
        if name == "separate":
            q, k, v = (torch.randn(B, S, C, device="cuda").to(BF) for _ in range(3))
        else:
            buf = torch.randn(B, S, pitch, device="cuda").to(BF)
            q, k, v = buf[:, :, 0:C], buf[:, :, C:2 * C], buf[:, :, 2 * C:3 * C]
        o = torch.empty(B, S, C, device="cuda", dtype=BF); lse = torch.empty(B, H, S, device="cuda")
        do = torch.randn(B, S, C, device="cuda").to(BF)
        dbuf = torch.empty(B, S, pitch if name != "separate" else 3 * C, device="cuda", dtype=BF)
        ld = dbuf.shape[-1]
        dq, dk, dv = dbuf[:, :, 0:C], dbuf[:, :, C:2 * C], dbuf[:, :, 2 * C:3 * C]
        delta = torch.empty(2, B, H, S, device="cuda")
        fwd = lambda: check(lib().pea_op_attention_fwd(P(q), q.stride(1), P(k), k.stride(1), P(v), v.stride(1), P(o), C, P(lse), B, H, S, S, 0.125, 1, stream_ptr()))
        bwd = lambda: check(lib().pea_op_attention_bwd(P(q), q.stride(1), P(k), k.stride(1), P(v), v.stride(1), P(o), C, P(do), C, P(lse), P(delta),
                                                       P(dq), ld, P(dk), ld, P(dv), ld, B, H, S, S, 0.125, 0, 0, 1, None, stream_ptr()))
        tf = min(timeit(fwd) for _ in range(3))
        tb = min(timeit(bwd) for _ in range(3))
        line += f" | {name}: fwd {tf*1e6:6.1f} bwd {tb*1e6:6.1f} us"
    print(line, flush=True)
