"""BASELINE config 5: SDXL + ControlNet 1024x1024 inference (tests/test_sdxl_zh_controlnet.py denoise loop), N images
per call with classifier-free guidance (UNet / ControlNet batch 2N), DPM-Solver++ steps.  Random-init weights,
synthetic prompt embeddings and canny image; reports seconds per generation, it/s and TFLOP/s."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import config as pc
from pea_diffusion_amd.controlnet import HipControlNet
from pea_diffusion_amd.sampler import DPMSolverMultistep, denoise
from pea_diffusion_amd.unet import HipUNet

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=4)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--no-controlnet", action="store_true")
ap.add_argument("--latent", type=int, default=128)
ap.add_argument("--end-to-end", action="store_true", help="also: Chinese-CLIP text tower + adapter in front, VAE decode behind")
a = ap.parse_args()
cfg = pc.sdxl_config()
N, hw = a.images, a.latent
dev = torch.device("cuda")
unet = HipUNet(cfg, 2 * N, hw, hw, 77, residual_inputs=not a.no_controlnet)
unet.init_random(1)
cn = None
if not a.no_controlnet:
    cn = HipControlNet(cfg, 2 * N, hw, hw, 77)
    cn.init_random(2)
g = torch.Generator(device="cpu").manual_seed(0)
lat = torch.randn(N, 4, hw, hw, generator=g).to(dev)
ehs = torch.randn(2 * N, 77, 2048, generator=g).to(dev, torch.bfloat16)
added = {"text_embeds": torch.randn(2 * N, 1280, generator=g).to(dev, torch.bfloat16),
         "time_ids": torch.tensor([[hw * 8, hw * 8, 0, 0, hw * 8, hw * 8]] * (2 * N)).to(dev)}
img = (torch.rand(N, 3, hw * 8, hw * 8, generator=g) > 0.9).float().to(dev)
img2 = torch.cat([img] * 2)

class Pipe:
    def __call__(self, x, t, encoder_hidden_states=None, added_cond_kwargs=None, return_dict=False):
        if cn is not None:
            cn.run(x, t, encoder_hidden_states, img2, added_cond_kwargs)
            cn.feed(unet, 0.5)
        return unet(x, t, encoder_hidden_states=encoder_hidden_states, added_cond_kwargs=added_cond_kwargs)

def gen(steps):
    out = denoise(Pipe(), DPMSolverMultistep(), lat.clone(), ehs, added, num_inference_steps=steps, guidance_scale=5.0)
    torch.cuda.synchronize()
    return out
gen(2)
t0 = time.perf_counter()
out = gen(a.steps)
dt = time.perf_counter() - t0
tf_img_step = 2 * (6.765 + (0.0 if cn is None else 3.02))
print(f"SDXL{'' if cn is None else ' + ControlNet'} {hw*8}x{hw*8}, {N} images (UNet batch {2*N}), {a.steps} steps: "
      f"{dt:.3f} s/generation = {dt/N:.3f} s/image, {a.steps/dt:.2f} it/s, "
      f"{tf_img_step*N*a.steps/dt:.0f} TFLOP/s; latents finite={bool(torch.isfinite(out).all())}; "
      f"unet mem {unet.memory()['activation_bytes']/2**30:.1f} GiB act" + ("" if cn is None else f", controlnet {cn.memory()['activation_bytes']/2**30:.1f} GiB act"))

if a.end_to_end:
    # tests/test_sdxl_zh.py:153-290 (encode_prompt through the adapter) and :408-431 (VAE decode)
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.text import HipTextEncoder
    from pea_diffusion_amd.vae import HipVAEDecoder
    zh = HipTextEncoder(pc.cnclip_bert_large_config(), 2 * N, 52); zh.init_random(3)
    proj = PEAAdapter(1024, 1280, 2048, 2048, False).to(dev)
    dec = HipVAEDecoder(pc.sdxl_vae_config(), N, hw, hw); dec.init_random(4)
    unet52 = None
    ids = torch.randint(1, 21000, (2 * N, 52), generator=g); ids[:, 30:] = 0
    ids = ids.to(dev)
    def front():
        tok, _ = zh.encode_text(ids)
        pooled, tokens = proj(tok.to(torch.float32))
        return pooled, tokens
    def back(lat_):
        return dec.decode(lat_, inv_scaling=1.0 / pc.sdxl_vae_config().scaling_factor)[0]
    front(); back(out); torch.cuda.synchronize()
    t0 = time.perf_counter(); pooled, tokens = front(); torch.cuda.synchronize(); t_front = time.perf_counter() - t0
    t0 = time.perf_counter(); img = back(out); torch.cuda.synchronize(); t_back = time.perf_counter() - t0
    print(f"end to end: text tower + adapter {t_front*1e3:.1f} ms (tokens {tuple(tokens.shape)}, pooled {tuple(pooled.shape)}), "
          f"denoise {dt:.3f} s, VAE decode {t_back*1e3:.1f} ms (image {tuple(img.shape)}, finite={bool(torch.isfinite(img).all())}) "
          f"=> {t_front + dt + t_back:.3f} s for {N} images")
