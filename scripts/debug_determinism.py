"""debug aid: run ops / UNet forward twice on identical inputs and report the first mismatch"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import ops, config as pc
from pea_diffusion_amd.unet import HipUNet
from pea_diffusion_amd.adapter import PEAAdapter

BF = torch.bfloat16
def same(name, a, b):
    d = (a.float() - b.float()).abs().max().item()
    print(f"{name}: max diff {d:.3e} {'SAME' if d == 0 else 'DIFF'}")

torch.manual_seed(0)
# op level
a, w = torch.randn(300, 512).to(BF).cuda(), torch.randn(256, 512).to(BF).cuda()
same("gemm", ops.gemm(a, w), ops.gemm(a, w))
x = torch.randn(2, 16, 16, 64).to(BF).cuda(); wc = ops.pack_conv(torch.randn(128, 64, 3, 3).cuda())
same("conv", ops.conv3x3(x, wc), ops.conv3x3(x, wc))
same("conv s2", ops.conv3x3(x, wc, stride=2), ops.conv3x3(x, wc, stride=2))
same("conv ups", ops.conv3x3(x, wc, upsample2x=True), ops.conv3x3(x, wc, upsample2x=True))
xg = torch.randn(2, 256, 64).to(BF).cuda(); g = torch.ones(64).cuda(); b = torch.zeros(64).cuda()
y1, s1 = ops.groupnorm_fwd(xg, g, b, 32, 1e-5, True); y2, s2 = ops.groupnorm_fwd(xg, g, b, 32, 1e-5, True)
same("gn y", y1, y2); same("gn stats", s1, s2)
xl = torch.randn(100, 128).to(BF).cuda(); gl = torch.ones(128).cuda(); bl = torch.zeros(128).cuda()
same("ln", ops.layernorm_fwd(xl, gl, bl)[0], ops.layernorm_fwd(xl, gl, bl)[0])
q, k, v = [torch.randn(2, 64, 128).to(BF).cuda() for _ in range(3)]
same("attn self", ops.attention_fwd(q, k, v, 2)[0], ops.attention_fwd(q, k, v, 2)[0])
k2, v2 = [torch.randn(2, 12, 128).to(BF).cuda() for _ in range(2)]
same("attn cross12", ops.attention_fwd(q, k2, v2, 2)[0], ops.attention_fwd(q, k2, v2, 2)[0])
q16 = torch.randn(2, 16, 128).to(BF).cuda()
same("attn sq16", ops.attention_fwd(q16, q16, q16, 2)[0], ops.attention_fwd(q16, q16, q16, 2)[0])

# model level
cfg = pc.tiny_config()
for B, L in [(2, 12), (2, 77)]:
    u = HipUNet(cfg, B, 16, 16, L)
    u.init_random(1)
    xx = torch.randn(B, 4, 16, 16).cuda(); t = torch.tensor([10., 500.]).cuda(); ehs = torch.randn(B, L, 128).cuda()
    added = {"text_embeds": torch.randn(B, 128).cuda(), "time_ids": torch.tensor([[128., 128, 0, 0, 128, 128]] * B).cuda()}
    e1 = u(xx, t, ehs, added_cond_kwargs=added)[0].clone(); taps1 = [u.tap(i).clone() for i in range(u.num_taps)]
    e2 = u(xx, t, ehs, added_cond_kwargs=added)[0].clone(); taps2 = [u.tap(i).clone() for i in range(u.num_taps)]
    same(f"unet B{B} L{L} eps", e1, e2)
    for i in range(u.num_taps):
        same(f"   tap{i}", taps1[i], taps2[i])
ad = PEAAdapter(128, 128, 192, 128, False).cuda()
xe = torch.randn(4, 12, 128).cuda()
with torch.no_grad():
    p1, t1 = ad(xe); p2, t2 = ad(xe)
same("adapter pooled", p1, p2); same("adapter tokens", t1, t2)
