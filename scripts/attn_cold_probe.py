"""Attention kernels with their operands in the Infinity Cache (hot: the same tensors every call) against operands from HBM
(cold: rotated through more sets than the cache holds) -- the backward reads Q / K / V / O written by the forward ~50 ms earlier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctypes
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib, check, stream_ptr
BF = torch.bfloat16
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None     # (column-block views of a fused Q|K|V tensor, as on the tape)

def attn_fwd(q, k, v, H):
    B, Sq, C = q.shape
    Skv = k.shape[1]
    o = torch.empty(B, Sq, C, device=q.device, dtype=BF)
    lse = torch.empty(B, H, Sq, device=q.device, dtype=torch.float32)
    check(lib().pea_op_attention_fwd(P(q), q.stride(1), P(k), k.stride(1), P(v), v.stride(1), P(o), C, P(lse), B, H, Sq, Skv, 0.125, 1, stream_ptr()))
    return o, lse

_scr = {}
def attn_bwd(q, k, v, o, do, lse, H):
    B, Sq, C = q.shape
    Skv = k.shape[1]
    key = (B, Sq, Skv, C)
    if key not in _scr:
        nb = lib().pea_op_attention_bwd_scratch_bytes(B, H, Sq, Skv, 1)
        _scr[key] = (torch.empty(B, Sq, C, device=q.device, dtype=BF), torch.empty(B, Skv, C, device=q.device, dtype=BF),
                     torch.empty(B, Skv, C, device=q.device, dtype=BF), torch.empty(2, B, H, Sq, device=q.device),
                     torch.empty(max(nb, 16), device=q.device, dtype=torch.uint8))
    dq, dk, dv, delta, scratch = _scr[key]
    check(lib().pea_op_attention_bwd(P(q), q.stride(1), P(k), k.stride(1), P(v), v.stride(1), P(o), C, P(do), C, P(lse), P(delta),
                                     P(dq), C, P(dk), C, P(dv), C, B, H, Sq, Skv, 0.125, 0, 0, 1, P(scratch), stream_ptr()))

def med(fn, n, iters=24):
    for i in range(3): fn(i % n)
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(iters): fn((r * iters + i) % n)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    ts.sort()
    return ts[2]

for (B, H, Sq, Skv) in [(4, 20, 1024, 1024), (4, 10, 4096, 4096), (4, 20, 1024, 77), (4, 10, 4096, 77), (8, 20, 1024, 1024), (8, 20, 1024, 77)]:
    C = H * 64
    per = B * Sq * C * 2 * (3 if Skv == Sq else 1) + 2 * B * Sq * C * 2
    n = max(2, int(500e6 / per) + 1)
    sets = []
    for i in range(n):
        if Skv == Sq:
            qkv = torch.randn(B, Sq, 3 * C, device="cuda").to(BF)
            q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
        else:
            q = torch.randn(B, Sq, C, device="cuda").to(BF)
            kv = torch.randn(B, Skv, 2 * C, device="cuda").to(BF)
            k, v = kv[:, :, :C], kv[:, :, C:]
        o, lse = attn_fwd(q, k, v, H)
        do = torch.randn_like(o)
        sets.append((q, k, v, o, do, lse))
    f = lambda i: attn_fwd(*sets[i][:3], H)
    b = lambda i: attn_bwd(*sets[i], H)
    fh, fc = med(lambda i: f(0), n), med(f, n)
    bh, bc = med(lambda i: b(0), n), med(b, n)
    print(f"attn B{B} H{H} Sq{Sq} Skv{Skv} ({n} sets): fwd hot {fh:7.1f} us / cold {fc:7.1f} us | bwd hot {bh:7.1f} us / cold {bc:7.1f} us", flush=True)
