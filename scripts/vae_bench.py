"""VAE encode (train_sdxl_zh.py:306-309) timing at the training resolution: ms per batch, TFLOP/s, per-family split."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pea_diffusion_amd import config as pc
from pea_diffusion_amd._lib import lib
from pea_diffusion_amd.vae import HipVAEEncoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
vae = HipVAEEncoder(pc.sdxl_vae_config(), B, hw, hw)
vae.init_random(0)
x = torch.randn(B, 3, hw, hw, device="cuda").clamp(-1, 1)
for _ in range(2): vae.encode_latents(x)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
n = 5
for _ in range(n): vae.encode_latents(x)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
fl = pc.vae_encoder_flops(pc.sdxl_vae_config(), hw, hw) * B
print(f"VAE encode B={B} {hw}x{hw}: {ms:.2f} ms/batch  {fl/ms/1e9:.1f} TFLOP/s  ({fl/B/1e12:.3f} TFLOP/img)  memory {vae.memory()}")
L = lib()
L.pea_prof_reset(); L.pea_prof_enable(1)
vae.encode_latents(x); torch.cuda.synchronize()
L.pea_prof_enable(0)
for f in range(8):
    t, fl_, by, k = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
    L.pea_prof_report(f, ctypes.byref(t), ctypes.byref(fl_), ctypes.byref(by), ctypes.byref(k))
    if k.value: print(f"  {L.pea_prof_family_name(f).decode():28s} {t.value:8.2f} ms {k.value:5d} launches {fl_.value/max(t.value,1e-9)/1e9:8.1f} TFLOP/s {by.value/max(t.value,1e-9)/1e6:8.0f} GB/s")

# ---- decode (tests/test_sdxl_zh.py:430): 128x128 latents -> 1024x1024 image
from pea_diffusion_amd.vae import HipVAEDecoder
del vae
torch.cuda.empty_cache()
dec = HipVAEDecoder(pc.sdxl_vae_config(), B, hw // 8, hw // 8)
dec.init_random(1)
z = torch.randn(B, 4, hw // 8, hw // 8, device="cuda")
for _ in range(2): dec.decode(z)
torch.cuda.synchronize()
s.record()
for _ in range(n): dec.decode(z)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
print(f"VAE decode B={B} -> {hw}x{hw}: {ms:.2f} ms/batch  ({ms/B:.2f} ms/image)  memory {dec.memory()}")
L.pea_prof_reset(); L.pea_prof_enable(1)
dec.decode(z); torch.cuda.synchronize()
L.pea_prof_enable(0)
for f in range(8):
    t, fl_, by, k = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
    L.pea_prof_report(f, ctypes.byref(t), ctypes.byref(fl_), ctypes.byref(by), ctypes.byref(k))
    if k.value: print(f"  {L.pea_prof_family_name(f).decode():28s} {t.value:8.2f} ms {k.value:5d} launches {fl_.value/max(t.value,1e-9)/1e9:8.1f} TFLOP/s {by.value/max(t.value,1e-9)/1e6:8.0f} GB/s")
