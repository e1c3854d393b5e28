"""`HipVAEEncoder`: the VAE call that produces the training step's latents --
`latents = self.vae.encode(pixel_values.float()).latent_dist.sample(); latents = latents.half() * scaling_factor`
(train_sdxl_zh.py:306-309; train_sd_zh.py:188-189) -- on the HIP op tape of libpea_hip.so (bf16 storage, fp32
accumulation: bf16 has fp32's exponent range, so the fp16-overflow reason for the reference's `.to(float32)` does not
apply).  Weights are addressed by the diffusers AutoencoderKL keys (`encoder.*`, `quant_conv.*`); decoder keys in a
full checkpoint are ignored.  `encode(..., stream=...)` lets the caller put the next batch's encode on a side HIP
stream while the current step's gradient all-reduce runs (BASELINE north_star)."""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import config as _cfg
from ._lib import PeaError, check, lib, ptr, stream_ptr
from .unet import HipUNet


class DiagonalGaussian:
    """`.latent_dist` of the encode result: `sample()`, `mode()`, `mean`, `logvar`, `std`."""

    def __init__(self, vae: "HipVAEEncoder", moments: torch.Tensor):
        self._vae, self.moments = vae, moments
        self.mean, lv = moments.chunk(2, dim=1)
        self.logvar = lv.clamp(-30.0, 20.0)

    @property
    def std(self):
        return torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=torch.float32)
        return self.mean + self.std * noise.to(self.mean.device, torch.float32)

    def mode(self):
        return self.mean


class _EncodeOutput:
    def __init__(self, dist):
        self.latent_dist = dist

    def __getitem__(self, i):
        return (self.latent_dist,)[i]


class _Cfg:
    def __init__(self, cfg):
        self.__dict__.update(cfg.__dict__)


class HipVAEEncoder:
    def __init__(self, cfg, batch: int, height: Optional[int] = None, width: Optional[int] = None):
        if not torch.cuda.is_available():
            raise PeaError("HipVAEEncoder needs a MI355X (no CPU fallback)")
        self.cfg, self.config = cfg, _Cfg(cfg)
        self.B, self.H, self.W = batch, height or cfg.sample_size, width or cfg.sample_size
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.dtype = torch.float32
        self._h = ctypes.c_void_p()
        c = _cfg.vae_to_c(cfg)
        check(lib().pea_vae_encoder_create(ctypes.byref(c), self.B, self.H, self.W, ctypes.byref(self._h)))
        C, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().pea_vae_latent_shape(self._h, ctypes.byref(C), ctypes.byref(h), ctypes.byref(w)))
        self.latent_shape = (self.B, C.value, h.value, w.value)

    __del__ = HipUNet.__del__
    weight_table = HipUNet.weight_table
    memory = HipUNet.memory

    def to(self, *a, **k):          # `self.vae.to(dtype=torch.float32)` (train_sdxl_zh.py:307) is a no-op here
        return self

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """accepts a full AutoencoderKL state dict: `decoder.*` / `post_quant_conv.*` entries are not part of the
        encode path and are skipped"""
        sd = {k: v for k, v in sd.items() if not (k.startswith("decoder.") or k.startswith("post_quant_conv."))}
        return HipUNet.load_state_dict(self, sd, strict)

    def init_random(self, seed: int = 0):
        check(lib().pea_unet_init_random(self._h, seed, stream_ptr()))

    # ------------------------------------------------------------------ encode
    def _pixels(self, x):
        if tuple(x.shape) != (self.B, self.cfg.in_channels, self.H, self.W):
            raise PeaError(f"HipVAEEncoder built for {(self.B, self.cfg.in_channels, self.H, self.W)}, got {tuple(x.shape)}")
        return x.detach().to(self.device, torch.float32).contiguous()

    def encode(self, pixel_values, return_dict: bool = True):
        """`vae.encode(x).latent_dist` -- moments are materialised, sampling happens in torch on request."""
        x = self._pixels(pixel_values)
        mom = torch.empty(self.B, 2 * self.latent_shape[1], *self.latent_shape[2:], device=self.device)
        check(lib().pea_vae_encode(self._h, ptr(x), None, 1.0, ptr(mom), None, stream_ptr()))
        self._keep = x
        return _EncodeOutput(DiagonalGaussian(self, mom))

    def encode_latents(self, pixel_values, noise: Optional[torch.Tensor] = None,
                       generator: Optional[torch.Generator] = None, sample: bool = True):
        """The fused form of train_sdxl_zh.py:306-309: `encode(x).latent_dist.sample() * scaling_factor` in one call
        (posterior sampling and scaling run in the kernel that applies quant_conv).  fp32 [B, 4, H/8, W/8]."""
        x = self._pixels(pixel_values)
        if sample and noise is None:
            noise = torch.randn(self.latent_shape, generator=generator, device=self.device, dtype=torch.float32)
        if noise is not None:
            noise = noise.to(self.device, torch.float32).contiguous()
        out = torch.empty(self.latent_shape, device=self.device, dtype=torch.float32)
        check(lib().pea_vae_encode(self._h, ptr(x), ptr(noise), float(self.cfg.scaling_factor), None, ptr(out),
                                   stream_ptr()))
        self._keep = (x, noise)
        return out


class HipVAEDecoder:
    """`image = self.vae.decode(latents / self.vae.config.scaling_factor, return_dict=False)[0]`
    (tests/test_sdxl_zh.py:430; tests/test_sdxl_zh_controlnet.py:575) on the HIP tape.  Built for a LATENT size;
    weights by the AutoencoderKL keys `decoder.*`, `post_quant_conv.*` (encoder keys of a full checkpoint are skipped)."""

    def __init__(self, cfg, batch: int, latent_height: Optional[int] = None, latent_width: Optional[int] = None):
        if not torch.cuda.is_available():
            raise PeaError("HipVAEDecoder needs a MI355X (no CPU fallback)")
        self.cfg, self.config = cfg, _Cfg(cfg)
        f = 2 ** (len(cfg.block_out_channels) - 1)
        self.B, self.h, self.w = batch, latent_height or cfg.sample_size // f, latent_width or cfg.sample_size // f
        self.scale_factor = f
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.dtype = torch.float32
        self._h = ctypes.c_void_p()
        c = _cfg.vae_decoder_to_c(cfg)
        check(lib().pea_vae_decoder_create(ctypes.byref(c), self.B, self.h, self.w, ctypes.byref(self._h)))

    __del__ = HipUNet.__del__
    weight_table = HipUNet.weight_table
    memory = HipUNet.memory

    def to(self, *a, **k):
        return self

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        sd = {k: v for k, v in sd.items() if not (k.startswith("encoder.") or k.startswith("quant_conv."))}
        return HipUNet.load_state_dict(self, sd, strict)

    def init_random(self, seed: int = 0):
        check(lib().pea_unet_init_random(self._h, seed, stream_ptr()))

    def decode(self, z, return_dict: bool = False, inv_scaling: float = 1.0):
        """z: [B, 4, h, w] (already divided by the scaling factor, as the reference passes it; or pass the raw latents
        with inv_scaling = 1 / scaling_factor and the division happens in the post_quant kernel) -> ([B, 3, 8h, 8w],)"""
        if tuple(z.shape) != (self.B, self.cfg.latent_channels, self.h, self.w):
            raise PeaError(f"HipVAEDecoder built for {(self.B, self.cfg.latent_channels, self.h, self.w)}, got {tuple(z.shape)}")
        zz = z.detach().to(self.device, torch.float32).contiguous()
        img = torch.empty(self.B, 3, self.h * self.scale_factor, self.w * self.scale_factor, device=self.device)
        check(lib().pea_vae_decode(self._h, ptr(zz), float(inv_scaling), ptr(img), stream_ptr()))
        self._keep = zz
        return (img,)
