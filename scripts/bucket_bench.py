"""Side measurement: the KD step over the reference's nine aspect-ratio buckets (utils/custom_dataset_sdxl.py:30), one
bucket per batch in random order as its dataloader delivers them, through BucketedTrainer (one context per bucket, all
resident).  SDXL, bf16, per-GPU batch 4, synthetic post-encoder batches, AdamW included.
usage: python scripts/bucket_bench.py [steps] [batch]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import config as pc
from pea_diffusion_amd.adapter import PEAAdapter
from pea_diffusion_amd.frontend import BUCKETS
from pea_diffusion_amd.train import BucketedTrainer
from pea_diffusion_amd.unet import HipUNet

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 45
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = 77
cfg = pc.sdxl_config()
student = HipUNet(cfg, B, 80, 80, L, needs_grad=True)
student.init_random(3)
teacher = HipUNet(cfg, B, 80, 80, L, share_weights_from=student)
torch.manual_seed(0)
ad = PEAAdapter(1024, 1280, 1024, 2048, False).cuda()
tr = BucketedTrainer(ad, student, teacher)
g = torch.Generator(device="cuda").manual_seed(5)
r = lambda *s: torch.randn(*s, generator=g, device="cuda")
batches = []
for (H, W) in BUCKETS:
    h, w = H // 8, W // 8
    batches.append(dict(latents=r(B, 4, h, w), noise=r(B, 4, h, w), timesteps=torch.randint(0, 1000, (B,), device="cuda"),
                        enc=r(B, L, 1024), enc_uncond=r(B, L, 1024), prompt_mask=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                        zh_or_not=torch.tensor([1, 0] * (B // 2) + [1] * (B % 2), device="cuda"),
                        teacher_ehs=r(B, L, 2048), teacher_neg=r(B, L, 2048), teacher_pooled=r(B, 1280),
                        time_ids=torch.tensor([[H, W, 0, 0, H, W]] * B, dtype=torch.float32, device="cuda")))
for b in batches:                       # first touch of every bucket: context creation + arena allocation, untimed
    tr.training_step(b, 0)
    tr.optimizer_step()
torch.cuda.synchronize()
rng = random.Random(0)
order = [rng.randrange(9) for _ in range(steps)]
t0 = time.perf_counter()
for i in order:
    tr.training_step(batches[i], 0)
    tr.optimizer_step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
per = {}
for i in range(9):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(3):
        tr.training_step(batches[i], 0); tr.optimizer_step()
    torch.cuda.synchronize()
    per[i] = (time.perf_counter() - t1) / 3
print(f"nine buckets, random order, B={B}: {steps} steps in {dt:.3f} s -> {dt / steps * 1e3:.1f} ms/step, {B * steps / dt:.1f} images/s; "
      f"resident contexts {tr.resident_bytes() / 2**30:.0f} GiB")
print("per bucket (ms/step): " + ", ".join(f"{BUCKETS[i][0]}x{BUCKETS[i][1]} {per[i] * 1e3:.1f}" for i in range(9)))
