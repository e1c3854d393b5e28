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
from pea_diffusion_amd.frontend import PEAFrontEnd
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
ids_t = torch.randint(0, 49000, (B, 77), generator=g); ids_t[:, 0] = 49406; ids_t[:, 20:] = 49407
neg_t = torch.full((B, 77), 49407); neg_t[:, 0] = 49406
ids_z = torch.randint(1, 21000, (B, L), generator=g); ids_z[:, 30:] = 0
unc_z = torch.zeros(1, L, dtype=torch.int64); unc_z[:, 0] = 101; unc_z[:, 1] = 102
# the dataloader's dictionary (utils/custom_dataset_sdxl.py:397-407), English prompts pre-tokenised
batch = {"pixel_values": pixels, "input_ids": ids_z.to(dev), "input_ids_uncond": unc_z.to(dev),
         "original_size": [(1024, 1024)] * B, "crops_coords_top_left": [(0, 0)] * B, "bucket_id": [0] * B,
         "zh_or_not": [1] * B, "texts_en_ids": (ids_t.to(dev), ids_t.to(dev)), "neg_en_ids": (neg_t.to(dev), neg_t.to(dev))}
fe = PEAFrontEnd(vae, te1, te2, zh)
trainer.attach_frontend(fe)

def step():
    out = trainer.training_step_from_batch(batch)
    trainer.optimizer_step()
    return out

def timed(f, n):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, r

import pea_diffusion_amd.frontend as fmod
res = {}
combos = [(False, False, True), (True, False, True), (True, True, True)]
for conc, drain, acopy in combos:   # towers on their own streams / host drains first / pinned-ring copies
    fe.concurrent_towers, fe.drain_before_enqueue, fmod.ASYNC_COPIES = conc, drain, acopy
    dt, out = timed(step, a.steps)
    fd, _ = timed(lambda: fe.prepare(batch), a.steps)
    print(f"towers on their own streams={conc} host drains the stream first={drain}: step {dt*1e3:.1f} ms, front end alone {fd*1e3:.1f} ms", flush=True)
    res[conc] = (dt, fd, float(out["loss"]))
fe.concurrent_towers, fe.drain_before_enqueue, fmod.ASYNC_COPIES = True, True, True
for conc in (True, False):
    dt, fd, loss = res[conc]
    print(f"full reference training_step (PEATrainer.training_step_from_batch: VAE encode + 3 text encoders + KD step + AdamW), SDXL "
          f"1024x1024, batch {B}, ctx {L}, text towers {'on their own streams' if conc else 'on the VAE stream'}: {dt*1e3:.1f} ms/step = "
          f"{B/dt:.2f} images/s (loss {loss:.4f}); frozen front end alone {fd*1e3:.1f} ms (VAE + CLIP-L + OpenCLIP-bigG on 2B prompts "
          f"+ BERT-large on 2B prompts)")
