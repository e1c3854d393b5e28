import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import config as pc
from pea_diffusion_amd.unet import HipUNet
from pea_diffusion_amd._lib import lib, check, ptr, stream_ptr
cfg = pc.sdxl_config()
B, hw, L = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 77
u = HipUNet(cfg, B, hw, hw, L, needs_grad=True); u.init_random(3)
g = torch.Generator(device="cuda").manual_seed(5)
r = lambda *s: torch.randn(*s, generator=g, device="cuda")
x, t, ehs = r(B, 4, hw, hw), torch.tensor([500.], device="cuda"), r(B, L, 2048)
added = {"text_embeds": r(B, 1280), "time_ids": torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * B, device="cuda")}
eps = u(x, t, ehs, added_cond_kwargs=added)[0]
print("eps: mean %.3e std %.3e max %.3e finite %s" % (eps.mean(), eps.std(), eps.abs().max(), torch.isfinite(eps).all().item()))
for k in range(u.num_taps):
    tp = u.tap(k); print("tap", k, tuple(tp.shape), "std %.3e max %.3e" % (tp.std(), tp.abs().max()))
for scale in (1.0, 1e-4):
    d_eps = r(B, 4, hw, hw) * scale
    d_ehs, d_text = u.backward(d_eps, 0)
    print("scale", scale, "d_ehs: std %.3e max %.3e nonzero %.4f | d_text std %.3e nonzero %.4f" % (
        d_ehs.std(), d_ehs.abs().max(), (d_ehs != 0).float().mean(), d_text.std(), (d_text != 0).float().mean()))
