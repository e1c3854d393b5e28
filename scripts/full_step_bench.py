"""The reference's WHOLE training_step (train_sdxl_zh.py:305-449) on the HIP path, frozen encoders included: VAE encode of
the 1024x1024 pixels (:306-309), teacher text encoders CLIP-L + OpenCLIP-bigG on prompt and negative prompt
(encode_prompt, :410), student Chinese-CLIP BERT tower on ids and unconditional ids (:327-329), then the KD step and the
optimizer.  Random-init weights, synthetic pixels / token ids.  bench.py's headline metric starts from latents and
embeddings; this script reports what the frozen front end adds."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import config as pc
from pea_diffusion_amd.adapter import PEAAdapter
from pea_diffusion_amd.text import HipTextEncoder
from pea_diffusion_amd.train import PEATrainer
from pea_diffusion_amd.unet import HipUNet
from pea_diffusion_amd.vae import HipVAEEncoder

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--ctx", type=int, default=77, help="student context length (77 keeps the merged passes; cn_clip's default is 52)")
a = ap.parse_args()
B, L, dev = a.batch, a.ctx, torch.device("cuda")
cfg = pc.sdxl_config()
student = HipUNet(cfg, B, 128, 128, L, needs_grad=True); student.init_random(7)
teacher = HipUNet(cfg, B, 128, 128, 77, share_weights_from=student)
adapter = PEAAdapter(1024, cfg.pooled_dim, 1024, cfg.cross_attention_dim, False).to(dev)
trainer = PEATrainer(adapter, student, teacher)
vae = HipVAEEncoder(pc.sdxl_vae_config(), B); vae.init_random(1)
te1 = HipTextEncoder(pc.clip_l_config(), 2 * B, 77); te1.init_random(2)           # prompt | negative prompt in one batch
te2 = HipTextEncoder(pc.openclip_bigg_config(), 2 * B, 77); te2.init_random(3)
zh = HipTextEncoder(pc.cnclip_bert_large_config(), 2 * B, L); zh.init_random(4)   # ids | unconditional ids
g = torch.Generator().manual_seed(0)
pixels = torch.randn(B, 3, 1024, 1024, generator=g).clamp_(-1, 1).to(dev)
ids_t = torch.randint(0, 49000, (2 * B, 77), generator=g); ids_t[:, 0] = 49406; ids_t[:, 20:] = 49407
ids_t = ids_t.to(dev)
ids_z = torch.randint(1, 21000, (2 * B, L), generator=g); ids_z[:, 30:] = 0
ids_z = ids_z.to(dev)
time_ids = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * B).to(dev)

def step():
    latents = vae.encode_latents(pixels)                                                      # :306-309
    noise = torch.randn_like(latents) + 0.5 * torch.randn(B, 4, 1, 1, device=dev)             # :311-315 (noise_offset)
    t = torch.randint(0, 1000, (B,), device=dev)
    h1, _ = te1.encode(ids_t, hidden_index=-2)
    h2, pooled = te2.encode(ids_t, hidden_index=-2)
    pe = torch.cat([h1, h2], -1)                                                              # [2B, 77, 2048]
    enc, _ = zh.encode_text(ids_z)                                                            # [2B, L, 1024]
    batch = {"latents": latents, "noise": noise, "timesteps": t, "enc": enc[:B], "enc_uncond": enc[B:],
             "prompt_mask": torch.rand(B, device=dev) < 0.1, "zh_or_not": torch.randint(0, 2, (B,), device=dev),
             "teacher_ehs": pe[:B], "teacher_neg": pe[B:], "teacher_pooled": pooled[:B], "time_ids": time_ids}
    out = trainer.training_step(batch)
    trainer.optimizer_step()
    return out

for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps): out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
# the front end alone
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(a.steps):
    vae.encode_latents(pixels); te1.encode(ids_t); te2.encode(ids_t); zh.encode_text(ids_z)
torch.cuda.synchronize()
fe = (time.perf_counter() - t1) / a.steps
# the same with the frozen front end of the NEXT batch on a side HIP stream, overlapped with this batch's step
side = torch.cuda.Stream()
def front():
    latents = vae.encode_latents(pixels)
    h1, _ = te1.encode(ids_t, hidden_index=-2)
    h2, pooled = te2.encode(ids_t, hidden_index=-2)
    enc, _ = zh.encode_text(ids_z)
    return latents, torch.cat([h1, h2], -1), pooled, enc
def step_overlapped(cur):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        nxt = front()
    latents, pe, pooled, enc = cur
    noise = torch.randn_like(latents) + 0.5 * torch.randn(B, 4, 1, 1, device=dev)
    t = torch.randint(0, 1000, (B,), device=dev)
    batch = {"latents": latents, "noise": noise, "timesteps": t, "enc": enc[:B], "enc_uncond": enc[B:],
             "prompt_mask": torch.rand(B, device=dev) < 0.1, "zh_or_not": torch.randint(0, 2, (B,), device=dev),
             "teacher_ehs": pe[:B], "teacher_neg": pe[B:], "teacher_pooled": pooled[:B], "time_ids": time_ids}
    trainer.training_step(batch)
    trainer.optimizer_step()
    torch.cuda.current_stream().wait_stream(side)
    return nxt
cur = front()
for _ in range(2): cur = step_overlapped(cur)
torch.cuda.synchronize(); t2 = time.perf_counter()
for _ in range(a.steps): cur = step_overlapped(cur)
torch.cuda.synchronize()
ov = (time.perf_counter() - t2) / a.steps
print(f"front end of the next batch on a side stream: {ov*1e3:.1f} ms/step = {B/ov:.2f} images/s")
print(f"full reference training_step incl. VAE encode + 3 text encoders, SDXL 1024x1024, batch {B}, ctx {L}: "
      f"{dt*1e3:.1f} ms/step = {B/dt:.2f} images/s (loss {float(out['loss']):.4f}); frozen front end alone {fe*1e3:.1f} ms "
      f"(VAE + CLIP-L + OpenCLIP-bigG on 2B prompts + BERT-large on 2B prompts)")
