"""a few launches of the step's main GEMM shapes (for rocprofv3 --pmc passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
BF = torch.bfloat16
for (M, N, K) in [(8192, 10240, 1280), (8192, 1280, 5120), (4096, 1280, 1280), (8192, 1280, 1280)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    for _ in range(4):
        ops.gemm(a, w)
torch.cuda.synchronize()
