"""a few launches of the cross-attention forward / backward kernels (for rocprofv3 --pmc passes: scripts/xattn_pmc.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
from pea_diffusion_amd._lib import lib
BF = torch.bfloat16
B, H, Sq, Skv = 4, 20, 1024, 77
C = H * 64
q = (torch.randn(B, Sq, C, device="cuda") * 0.18).to(BF); k = torch.randn(B, Skv, C, device="cuda").to(BF); v = torch.randn(B, Skv, C, device="cuda").to(BF)
o, lse = ops.attention_fwd(q, k, v, H, q_prescaled=True)
do = torch.randn_like(o)
for ver in (0, 2, 3):
    lib().pea_debug_set_xattn_bwd_v2(ver)
    for _ in range(5):
        ops.attention_bwd(q, k, v, o, do, lse, H, q_prescaled=True)
for _ in range(5):
    ops.attention_fwd(q, k, v, H, q_prescaled=True)
torch.cuda.synchronize()
