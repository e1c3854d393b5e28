"""folded LayerNorm -> Linear GEMM against the plain GEMM of the same shape (run under rocprofv3 --kernel-trace --stats)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
BF = torch.bfloat16
for (M, N, K, geglu) in [(8192, 3840, 1280, False), (8192, 1280, 1280, False), (8192, 10240, 1280, True)]:
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    gamma = torch.ones(K, device="cuda"); beta = torch.zeros(K, device="cuda"); bias = torch.zeros(N, device="cuda")
    for _ in range(10):
        ops.ln_linear(x, gamma, beta, w, bias, geglu=geglu)
    for _ in range(10):
        if geglu: ops.gemm_geglu(x, w, bias, stash_grad=False)
        else: ops.gemm(x, w, bias=bias)
torch.cuda.synchronize()
