"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): torch fp32 restatement of the VAE encode that precedes the
training step -- `self.vae.encode(pixel_values).latent_dist.sample() * self.vae.config.scaling_factor`
(train_sdxl_zh.py:306-309; train_sd_zh.py:188-189).  `AutoencoderKL` lives in diffusers==0.23.0
(requirements.txt:25), which is absent from /root/reference and from this image: the module graph, the state-dict
keys (`encoder.*`, `quant_conv.*`) and `DiagonalGaussianDistribution` are restated from its published definition --
**parity unpinned** at that boundary (the reference's tests hold no VAE vectors).  Structural known-answer: the SDXL
VAE encoder + quant_conv has 34 163 664 parameters (tests/test_vae_cpu.py)."""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VAEConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.13025          # SDXL; SD1.5: 0.18215
    sample_size: int = 1024
    name: str = "sdxl_vae"


def sdxl_vae_config() -> VAEConfig:
    return VAEConfig()


def sd15_vae_config() -> VAEConfig:
    return VAEConfig(scaling_factor=0.18215, sample_size=512, name="sd15_vae")


def tiny_vae_config() -> VAEConfig:
    return VAEConfig(block_out_channels=(64, 128, 128), sample_size=64, name="tiny_vae")


class ResnetBlock(nn.Module):               # ResnetBlock2D(temb_channels=None)
    def __init__(self, cin, cout, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (self.conv_shortcut(x) if self.conv_shortcut is not None else x) + h


class Downsample(nn.Module):                # Downsample2D(use_conv=True, padding=0)
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class DownEncoderBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, eps, down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if j == 0 else cout, cout, groups, eps) for j in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample(cout)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class MidAttention(nn.Module):              # Attention(heads=1, dim_head=C, residual_connection=True, bias=True)
    def __init__(self, c, groups, eps):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x.view(B, C, H * W)).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        p = torch.softmax(q @ k.transpose(1, 2) * (C ** -0.5), dim=-1)
        o = self.to_out[0](p @ v)
        return o.transpose(1, 2).reshape(B, C, H, W) + x


class MidBlock(nn.Module):
    def __init__(self, c, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(c, c, groups, eps), ResnetBlock(c, c, groups, eps)])
        self.attentions = nn.ModuleList([MidAttention(c, groups, eps)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g, e = cfg.block_out_channels, cfg.norm_num_groups, cfg.norm_eps
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = boc[0]
        for i, co in enumerate(boc):
            self.down_blocks.append(DownEncoderBlock(c, co, cfg.layers_per_block, g, e, down=(i != len(boc) - 1)))
            c = co
        self.mid_block = MidBlock(c, g, e)
        self.conv_norm_out = nn.GroupNorm(g, c, eps=e)
        self.conv_out = nn.Conv2d(c, 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DiagonalGaussianRef:
    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, noise=None, generator=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class _EncodeOut:
    def __init__(self, d):
        self.latent_dist = d


class VAEEncoderRef(nn.Module):
    """The `encode` half of AutoencoderKL: `vae.encode(x).latent_dist.sample()`."""

    def __init__(self, cfg: VAEConfig):
        super().__init__()
        self.config = cfg
        self.encoder = Encoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)

    def moments(self, x):
        return self.quant_conv(self.encoder(x))

    def encode(self, x):
        return _EncodeOut(DiagonalGaussianRef(self.moments(x)))


class Upsample(nn.Module):                  # Upsample2D(use_conv=True): nearest 2x, then conv 3x3
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class UpDecoderBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, eps, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if j == 0 else cout, cout, groups, eps) for j in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample(cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig, out_channels: int = 3):
        super().__init__()
        boc, g, e = cfg.block_out_channels, cfg.norm_num_groups, cfg.norm_eps
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(cfg.latent_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlock(rev[0], g, e)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(UpDecoderBlock(c, co, cfg.layers_per_block + 1, g, e, up=(i != len(rev) - 1)))
            c = co
        self.conv_norm_out = nn.GroupNorm(g, c, eps=e)
        self.conv_out = nn.Conv2d(c, out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class VAEDecoderRef(nn.Module):
    """The `decode` half of AutoencoderKL: `vae.decode(latents / scaling_factor, return_dict=False)[0]`
    (tests/test_sdxl_zh.py:430).  Published totals: decoder 49 490 179 + post_quant_conv 20 parameters."""

    def __init__(self, cfg: VAEConfig):
        super().__init__()
        self.config = cfg
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)
        self.decoder = Decoder(cfg)

    def decode(self, z, return_dict=False):
        return (self.decoder(self.post_quant_conv(z)),)


def vae_encoder_flops(cfg: VAEConfig, H: int, W: int) -> float:
    """analytic FLOPs of one encode (2*MACs of convs, linears and the mid attention), per image"""
    fl, c, h, w = 0.0, cfg.block_out_channels[0], H, W
    fl += 2.0 * h * w * c * 9 * cfg.in_channels
    for i, co in enumerate(cfg.block_out_channels):
        for j in range(cfg.layers_per_block):
            ci = c if j == 0 else co
            fl += 2.0 * h * w * 9 * (ci * co + co * co) + (2.0 * h * w * ci * co if ci != co else 0.0)
        c = co
        if i != len(cfg.block_out_channels) - 1:
            h, w = h // 2, w // 2
            fl += 2.0 * h * w * 9 * c * c
    fl += 2 * (2.0 * h * w * 9 * 2 * c * c)                                   # two mid resnets
    fl += 4 * 2.0 * h * w * c * c + 2 * 2.0 * (h * w) ** 2 * c                # q,k,v,out + scores + PV
    fl += 2.0 * h * w * 9 * c * 2 * cfg.latent_channels
    return fl
