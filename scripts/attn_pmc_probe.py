"""run each self-attention kernel a few times (for rocprofv3 --pmc passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
BF = torch.bfloat16
for (B, H, Sq, Skv) in [(4, 10, 4096, 4096), (4, 20, 1024, 1024)]:
    C = H * 64
    q = torch.randn(B, Sq, C, device="cuda").to(BF); k = torch.randn(B, Skv, C, device="cuda").to(BF); v = torch.randn(B, Skv, C, device="cuda").to(BF)
    for _ in range(3):
        o, lse = ops.attention_fwd(q, k, v, H)
    do = torch.randn_like(o)
    for _ in range(3):
        ops.attention_bwd(q, k, v, o, do, lse, H)
torch.cuda.synchronize()
