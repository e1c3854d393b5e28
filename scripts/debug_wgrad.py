import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops
BF = torch.bfloat16
for (M, N, K) in [(2048, 1280, 192), (2048, 1280, 64), (1280, 1024, 192), (154, 1280, 2048), (1024, 1024, 192), (2048, 1280, 256)]:
    a = torch.randn(M, K, device="cuda").to(BF); w = torch.randn(N, K, device="cuda").to(BF)
    ref = a.float() @ w.float().T
    o = ops.gemm(a, w, out_f32=True)
    acc = torch.zeros(M, N, device="cuda"); ops.gemm(a, w, out=acc, accum_f32=True)
    print(M, N, K, "f32 err %.3e accum err %.3e refmax %.2f nonzero %.3f" % ((o - ref).abs().max(), (acc - ref).abs().max(), ref.abs().max(), (o != 0).float().mean()))
from oracle.step_ref import AdapterRef
from pea_diffusion_amd.adapter import PEAAdapter
torch.manual_seed(0)
ref = AdapterRef(1024, 1280, 1024, 2048, False); hip = PEAAdapter(1024, 1280, 1024, 2048, False)
hip.load_state_dict(ref.state_dict()); hip = hip.cuda()
x = torch.randn(2, 77, 1024)
pr, tr_ = ref(x); ph, th = hip(x.cuda())
g1, g2 = torch.randn_like(pr), torch.randn_like(tr_)
torch.autograd.backward([pr, tr_], [g1, g2]); torch.autograd.backward([ph, th], [g1.cuda(), g2.cuda()])
for (k, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
    print(k, "rel_l2 %.3e |ref| %.3e |hip| %.3e" % (((p.grad.cpu() - q.grad).norm() / q.grad.norm()), q.grad.norm(), p.grad.norm()))
