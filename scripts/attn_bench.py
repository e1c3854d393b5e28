"""Attention kernels on the step's shapes (B=4): self-attention at 64x64 / 32x32 tokens, cross-attention over 77 keys."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
BF = torch.bfloat16
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
shapes = [(4, 10, 4096, 4096), (4, 20, 1024, 1024), (4, 10, 4096, 77), (4, 20, 1024, 77)]
if len(sys.argv) > 1 and sys.argv[1] == 'quant':
    shapes = [(4, 8, 4096, 4096), (4, 10, 4096, 4096), (4, 16, 4096, 4096), (2, 20, 4096, 4096), (4, 4, 4096, 4096), (4, 16, 1024, 1024), (4, 32, 1024, 1024), (8, 16, 1024, 1024)]
for (B, H, Sq, Skv) in shapes:
    C = H * 64
    q = torch.randn(B, Sq, C, device="cuda").to(BF); k = torch.randn(B, Skv, C, device="cuda").to(BF); v = torch.randn(B, Skv, C, device="cuda").to(BF)
    o, lse = ops.attention_fwd(q, k, v, H)
    ref = torch.nn.functional.scaled_dot_product_attention(q.view(B, Sq, H, 64).transpose(1, 2).float(), k.view(B, Skv, H, 64).transpose(1, 2).float(), v.view(B, Skv, H, 64).transpose(1, 2).float()).transpose(1, 2).reshape(B, Sq, C)
    err = (o.float() - ref).abs().max().item()
    do = torch.randn_like(o)
    tf = timeit(lambda: ops.attention_fwd(q, k, v, H))
    from pea_diffusion_amd._lib import lib
    fl = 4.0 * B * H * Sq * Skv * 64
    res = {}
    for rnd in range(3):                      # interleaved rounds in one process (fused launch vs two launches)
        for mode in (1, 0):
            lib().pea_debug_set_attn_fused_bwd(mode)
            tb = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, H))
            res.setdefault(mode, []).append(tb)
    lib().pea_debug_set_attn_fused_bwd(1)
    xline = ""
    if Skv <= 128:                            # one-pass cross-attention backward vs the general kernels
        tx = {0: 1e9, 1: 1e9}
        for rnd in range(3):
            for mode in (0, 1):
                lib().pea_debug_set_attn_xattn(mode)
                tx[mode] = min(tx[mode], timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, H), 20))
        lib().pea_debug_set_attn_xattn(1)
        xline = f" [one-pass {tx[1]*1e6:6.1f} us vs general kernels {tx[0]*1e6:6.1f} us]"
    g1 = ops.attention_bwd(q, k, v, o, do, lse, H)
    lib().pea_debug_set_attn_fused_bwd(0)
    g0 = ops.attention_bwd(q, k, v, o, do, lse, H)
    lib().pea_debug_set_attn_fused_bwd(1)
    same = all(torch.equal(a, b) for a, b in zip(g1, g0))
    t1, t0 = min(res[1]), min(res[0])
    print(f"attn B{B} H{H} Sq{Sq} Skv{Skv}: fwd {tf*1e6:7.1f} us {fl/tf/1e12:6.1f} TF | bwd fused {t1*1e6:7.1f} us {2.5*fl/t1/1e12:6.1f} TF, "
          f"two launches {t0*1e6:7.1f} us {2.5*fl/t0/1e12:6.1f} TF (bit-identical: {same}) | max err {err:.3e}{xline}", flush=True)
