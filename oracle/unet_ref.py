"""CPU ORACLE (test infrastructure, NOT product code) -- UNet2DConditionModel restatement.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

The reference never defines the UNet itself: it loads `diffusers==0.23.0`'s
`UNet2DConditionModel` (reference requirements.txt:25; loaded at
train_sdxl_zh.py:138,151 and train_sd_zh.py:100,111; called at train_sdxl_zh.py:397,415
and train_sd_zh.py:215,231).  diffusers is absent from this image, so this file restates
its published forward algorithm in plain torch fp32, keeping the diffusers state-dict key
names (`down_blocks.1.attentions.0.transformer_blocks.3.attn2.to_k.weight`, ...) so real
checkpoints would load, and keeping `down_blocks[i]` / `mid_block` / `up_blocks[i]` as
hook-able nn.Modules because the reference taps features with `register_forward_hook`
(train_sdxl_zh.py:69-84).  Down blocks return `(hidden, res_samples)` so that the
reference's `output[0]` hook (train_sdxl_zh.py:72-74) selects the hidden state.

PARITY STATUS: "parity unpinned" at this boundary -- no golden tensors exist for the
diffusers arithmetic.  Structural known answers pinned in tests/test_oracle_unet.py:
exact parameter totals (SDXL 2 567 463 684, SD1.5 859 520 964), per-block totals, tap
shapes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

try:                                               # (imported as oracle.unet_ref by the tests, as a plain module by scripts)
    from .bf16_store import bf16_storage, enabled as _bf16_on, st as _st  # noqa: F401  (identity unless inside `with bf16_storage():`)
except ImportError:                                # pragma: no cover
    from bf16_store import bf16_storage, enabled as _bf16_on, st as _st  # noqa: F401


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    sample_size: int = 128
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D")
    up_block_types: Tuple[str, ...] = ("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D")
    layers_per_block: int = 2
    # per down block: an int, or one int per attention of the block (diffusers >= 0.22 [ext], e.g. SSD-1B)
    transformer_layers_per_block: Tuple[Union[int, Tuple[int, ...]], ...] = (1, 2, 10)
    # per UP block, int or one per attention; None = the down list reversed (required when the down list is nested)
    reverse_transformer_layers_per_block: Optional[Tuple[Union[int, Tuple[int, ...]], ...]] = None
    mid_block_type: Optional[str] = "UNetMidBlock2DCrossAttn"   # None: `unet.mid_block is None`
    num_attention_heads: Tuple[int, ...] = (5, 10, 20)   # diffusers calls this `attention_head_dim`
    cross_attention_dim: int = 2048
    use_linear_projection: bool = True
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    addition_embed_type: Optional[str] = "text_time"
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    name: str = "sdxl"

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def pooled_dim(self) -> int:
        return self.projection_class_embeddings_input_dim - 6 * self.addition_time_embed_dim


def sdxl_config() -> UNetConfig:
    return UNetConfig()


def sd15_config() -> UNetConfig:
    return UNetConfig(
        sample_size=64, block_out_channels=(320, 640, 1280, 1280),
        down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
        up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
        transformer_layers_per_block=(1, 1, 1, 1), num_attention_heads=(8, 8, 8, 8),
        cross_attention_dim=768, use_linear_projection=False, addition_embed_type=None,
        addition_time_embed_dim=0, projection_class_embeddings_input_dim=0, name="sd15")


def ssd1b_config() -> UNetConfig:
    """SSD-1B (BASELINE config #4; the downstream UNet of tests/test_sdxl_zh.py:449-454).  segmind/SSD-1B
    `unet/config.json` [ext, recalled: no network here]: SDXL widths, attention stacks pruned per position
    (down [.., [2, 2], [4, 4]], up [[4, 4, 10], [2, 1, 1], ..]) and `mid_block_type: null`.  Known answer: the model
    card's 1.3 B parameters."""
    return UNetConfig(transformer_layers_per_block=(1, (2, 2), (4, 4)),
                      reverse_transformer_layers_per_block=((4, 4, 10), (2, 1, 1), 1),
                      mid_block_type=None, name="ssd1b")


def ssd1b_uniform_config() -> UNetConfig:
    """round-1 stand-in: uniform depths 1/2/4 with a mid block (an asymmetric shape case only)"""
    return UNetConfig(transformer_layers_per_block=(1, 2, 4), name="ssd1b_uniform")


def _depth_row(v, k):
    r = [int(v)] * k if isinstance(v, int) else [int(a) for a in v]
    assert len(r) == k, (v, k)
    return r


def tiny_config(heads64: bool = True) -> UNetConfig:
    """Small SDXL-shaped config for fast parity tests (all channel counts are multiples
    of 64 and head_dim is 64, the shapes the HIP kernels are tiled for)."""
    return UNetConfig(
        sample_size=16, block_out_channels=(64, 128, 128),
        transformer_layers_per_block=(1, 1, 2), num_attention_heads=(1, 2, 2),
        cross_attention_dim=128, addition_time_embed_dim=32,
        projection_class_embeddings_input_dim=128 + 6 * 32, name="tiny")


def tiny15_config() -> UNetConfig:
    """small SD1.5-shaped config: 4 levels, 8 heads per level (head dims 8 / 16, stored zero-padded to 64 on the
    HIP path), 1x1-conv projections, no added conditioning"""
    return UNetConfig(sample_size=16, block_out_channels=(64, 128, 128, 128),
                      down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                      up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
                      transformer_layers_per_block=(1, 1, 1, 1), num_attention_heads=(8, 8, 8, 8),
                      cross_attention_dim=128, use_linear_projection=False, addition_embed_type=None,
                      addition_time_embed_dim=0, projection_class_embeddings_input_dim=0, name="tiny15")


# ----------------------------------------------------------------------------- embeddings
def timestep_embedding(t: torch.Tensor, dim: int, flip_sin_to_cos: bool = True,
                       freq_shift: float = 0.0, max_period: float = 10000.0) -> torch.Tensor:
    """diffusers `get_timestep_embedding` (SDXL/SD1.5 use flip_sin_to_cos=True, shift 0):
    returns [N, dim] = (cos | sin) of t * exp(-ln(max_period) * i / half)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - freq_shift)
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(_st(F.silu(_st(self.linear_1(x)))))


# ----------------------------------------------------------------------------- blocks
class ResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, temb_dim: int, groups: int, eps: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        # (_st: a tensor the HIP path stores in bf16 -- identity outside oracle.bf16_store.bf16_storage())
        h = self.conv1(_st(F.silu(self.norm1(x))))
        h = _st(h + _st(self.time_emb_proj(_st(F.silu(temb))))[:, :, None, None])
        h = self.conv2(_st(F.silu(self.norm2(h))))
        if self.conv_shortcut is not None:
            x = _st(self.conv_shortcut(x))
        return _st(x + h)


class Attention(nn.Module):
    def __init__(self, query_dim: int, heads: int, cross_dim: Optional[int]):
        super().__init__()
        self.heads = heads
        kv_dim = cross_dim if cross_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, query_dim, bias=False)
        self.to_k = nn.Linear(kv_dim, query_dim, bias=False)
        self.to_v = nn.Linear(kv_dim, query_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim), nn.Identity()])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        B, S, C = x.shape
        H = self.heads
        q = _st(self.to_q(x)).view(B, S, H, C // H).transpose(1, 2)
        k = _st(self.to_k(ctx)).view(B, -1, H, C // H).transpose(1, 2)
        v = _st(self.to_v(ctx)).view(B, -1, H, C // H).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) * (C // H) ** -0.5, dim=-1)
        o = _st((_st(a) @ v).transpose(1, 2).reshape(B, S, C))          # (P is a bf16 MFMA operand of the P.V product)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim: int, inner: int):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return _st(h * F.gelu(gate))


class FeedForward(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, cross_dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, None)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        x = _st(self.attn1(_st(self.norm1(x))) + x)
        x = _st(self.attn2(_st(self.norm2(x)), ctx) + x)
        return _st(self.ff(_st(self.norm3(x))) + x)


class Transformer2DModel(nn.Module):
    def __init__(self, dim: int, heads: int, depth: int, cross_dim: int, groups: int, linear_proj: bool):
        super().__init__()
        self.linear_proj = linear_proj
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim) if linear_proj else nn.Conv2d(dim, dim, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(dim, heads, cross_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(dim, dim) if linear_proj else nn.Conv2d(dim, dim, 1)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        res = x
        h = _st(self.norm(x))
        if self.linear_proj:
            h = _st(self.proj_in(h.permute(0, 2, 3, 1).reshape(B, H * W, C)))
        else:
            h = _st(self.proj_in(h)).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        if self.linear_proj:
            h = self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = self.proj_out(h.reshape(B, H, W, C).permute(0, 3, 1, 2))
        return _st(h + res)


class Downsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return _st(self.conv(x))


class Upsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        if _bf16_on():
            return _st(self._subpixel(x))
        return _st(self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest")))

    def _subpixel(self, x):
        """bf16-storage mode only: the product STORES this conv's weights as four 2 x 2 kernels of summed taps, one per output
        parity (output row 2s + py reads source rows {s-1: w0, s: w1 + w2} for py = 0 and {s: w0 + w1, s+1: w2} for py = 1; the
        same along x), each sum rounded to bf16 (pea_diffusion_amd/csrc/elementwise.hip: pack_conv_subpix_kernel).  In exact
        arithmetic this equals conv(interpolate(x)); here the merged taps carry their storage rounding."""
        w, b = self.conv.weight, self.conv.bias
        B, _, H, W = x.shape
        y = x.new_empty(B, w.shape[0], 2 * H, 2 * W)
        taps = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
        for py in (0, 1):
            for px in (0, 1):
                k = torch.stack([torch.stack([sum(w[:, :, ky, kx] for ky in taps[py][dy] for kx in taps[px][dx])
                                              for dx in (0, 1)], -1) for dy in (0, 1)], -2)          # [Co][Ci][dy][dx]
                xp = F.pad(x, (1 - px, px, 1 - py, py))                                               # (left, right, top, bottom)
                y[:, :, py::2, px::2] = F.conv2d(xp, _st(k), b)
        return y


class DownBlock(nn.Module):
    """DownBlock2D / CrossAttnDownBlock2D: returns (hidden, res_samples)."""

    def __init__(self, cfg: UNetConfig, cin: int, cout: int, depth: int, heads: int, cross: bool, down: bool):
        super().__init__()
        g, e, t = cfg.norm_num_groups, cfg.norm_eps, cfg.time_embed_dim
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if j == 0 else cout, cout, t, g, e) for j in range(cfg.layers_per_block)])
        if cross:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, heads, d, cfg.cross_attention_dim, g, cfg.use_linear_projection)
                 for d in _depth_row(depth, cfg.layers_per_block)])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if down else None

    def forward(self, x, temb, ctx):
        outs = ()
        for j, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[j](x, ctx)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, cfg: UNetConfig, c: int, depth: int, heads: int):
        super().__init__()
        g, e, t = cfg.norm_num_groups, cfg.norm_eps, cfg.time_embed_dim
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, t, g, e), ResnetBlock2D(c, c, t, g, e)])
        self.attentions = nn.ModuleList(
            [Transformer2DModel(c, heads, depth, cfg.cross_attention_dim, g, cfg.use_linear_projection)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    """UpBlock2D / CrossAttnUpBlock2D."""

    def __init__(self, cfg: UNetConfig, cin: int, cout: int, cprev: int, depth: int, heads: int,
                 cross: bool, up: bool):
        super().__init__()
        g, e, t = cfg.norm_num_groups, cfg.norm_eps, cfg.time_embed_dim
        n = cfg.layers_per_block + 1
        self.resnets = nn.ModuleList()
        for j in range(n):
            skip = cin if j == n - 1 else cout
            rin = cprev if j == 0 else cout
            self.resnets.append(ResnetBlock2D(rin + skip, cout, t, g, e))
        if cross:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, heads, d, cfg.cross_attention_dim, g, cfg.use_linear_projection)
                 for d in _depth_row(depth, n)])
        else:
            self.attentions = None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if up else None

    def forward(self, x, res_samples, temb, ctx):
        for j, r in enumerate(self.resnets):
            x = torch.cat([x, res_samples[-1 - j]], dim=1)
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[j](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNet2DConditionRef(nn.Module):
    """`unet(sample, t, encoder_hidden_states, added_cond_kwargs=..., return_dict=False)[0]`
    -- the call made at train_sdxl_zh.py:397,415 and tests/test_sdxl_zh.py:384-391."""

    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.config = cfg
        self.in_channels = cfg.in_channels
        boc = cfg.block_out_channels
        nb = len(boc)
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], cfg.time_embed_dim)
        if cfg.addition_embed_type == "text_time":
            self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, cfg.time_embed_dim)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, ty in enumerate(cfg.down_block_types):
            cin, out = out, boc[i]
            self.down_blocks.append(DownBlock(cfg, cin, out, cfg.transformer_layers_per_block[i],
                                              cfg.num_attention_heads[i], ty.startswith("CrossAttn"),
                                              down=(i != nb - 1)))
        # diffusers >= 0.22 [ext]: UNet2DConditionModel hands transformer_layers_per_block[-1] to UNetMidBlock2DCrossAttn,
        # which broadcasts an int over its num_layers = 1 attention and otherwise reads element [0] (same rule as
        # pea_diffusion_amd/config.py:depth_tables); `mid_block_type: null` leaves `self.mid_block = None`
        if cfg.mid_block_type is None:
            self.mid_block = None
        else:
            last = cfg.transformer_layers_per_block[-1]
            self.mid_block = MidBlock(cfg, boc[-1], last if isinstance(last, int) else last[0], cfg.num_attention_heads[-1])
        self.up_blocks = nn.ModuleList()
        rboc = list(reversed(boc))
        if cfg.reverse_transformer_layers_per_block is not None:
            rdepth = list(cfg.reverse_transformer_layers_per_block)
        else:
            assert all(isinstance(t, int) for t in cfg.transformer_layers_per_block), \
                "reverse_transformer_layers_per_block is required with nested transformer_layers_per_block"
            rdepth = list(reversed(cfg.transformer_layers_per_block))
        rheads = list(reversed(cfg.num_attention_heads))
        out = rboc[0]
        for i, ty in enumerate(cfg.up_block_types):
            prev, out = out, rboc[i]
            cin = rboc[min(i + 1, nb - 1)]
            self.up_blocks.append(UpBlock(cfg, cin, out, prev, rdepth[i], rheads[i],
                                          ty.startswith("CrossAttn"), up=(i != nb - 1)))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, boc[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def embed(self, timesteps, added_cond_kwargs, B):
        cfg = self.config
        t = timesteps
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.int64)
        if t.dim() == 0:
            t = t[None]
        t = t.expand(B)
        emb = _st(self.time_embedding(_st(timestep_embedding(t, cfg.block_out_channels[0]).to(self.dtype))))
        if cfg.addition_embed_type == "text_time":
            text_embeds = added_cond_kwargs["text_embeds"]
            time_ids = added_cond_kwargs["time_ids"]
            te = _st(timestep_embedding(time_ids.flatten(), cfg.addition_time_embed_dim)).reshape(B, -1)
            add = torch.cat([_st(text_embeds), te.to(text_embeds.dtype)], dim=-1)
            emb = _st(emb + self.add_embedding(add.to(self.dtype)))
        return emb

    def forward(self, sample, timesteps, encoder_hidden_states, added_cond_kwargs=None,
                cross_attention_kwargs=None, return_dict=False, down_block_additional_residuals=None,
                mid_block_additional_residual=None):
        """ControlNet extras (tests/test_sdxl_zh_controlnet.py:534-535), diffusers 0.23 semantics [ext]: each
        down-path skip tensor gets its residual added AFTER the down path ran (only the copies consumed by the
        up blocks change), the mid residual is added to the mid-block output."""
        B = sample.shape[0]
        emb = self.embed(timesteps, added_cond_kwargs, B)
        x = _st(self.conv_in(sample))
        res = (x,)
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states)
            res += outs
        if down_block_additional_residuals is not None:
            assert len(down_block_additional_residuals) == len(res)
            res = tuple(r + a for r, a in zip(res, down_block_additional_residuals))
        if self.mid_block is not None:
            x = self.mid_block(x, emb, encoder_hidden_states)
        if mid_block_additional_residual is not None:
            x = x + mid_block_additional_residual
        for blk in self.up_blocks:
            n = len(blk.resnets)
            take, res = res[-n:], res[:-n]
            x = blk(x, take, emb, encoder_hidden_states)
        x = self.conv_out(_st(F.silu(self.conv_norm_out(x))))
        return (x,)


def tap_names(cfg: UNetConfig):
    """Feature-tap order used by the reference's `cast_hook` (train_sdxl_zh.py:79-84,
    train_sd_zh.py:69-74): d0..d{n-1}, m, u0..u{n-1} with n = NUM_blocks = number of
    UNet levels (3 for SDXL, 4 for SD1.5)."""
    n = len(cfg.block_out_channels)
    return [f"d{i}" for i in range(n)] + (["m"] if cfg.mid_block_type is not None else []) + [f"u{i}" for i in range(n)]


def cast_hook_ref(unet: UNet2DConditionRef, store: dict):
    """Restatement of `cast_hook` + `getActivation` (train_sdxl_zh.py:69-84)."""
    n = len(unet.config.block_out_channels)
    hs = []
    for i in range(n):
        hs.append(unet.down_blocks[i].register_forward_hook(
            lambda m, inp, out, k=f"d{i}": store.__setitem__(k, out[0])))
    if unet.mid_block is not None:      # (the reference's cast_hook would raise on `None.register_forward_hook`)
        hs.append(unet.mid_block.register_forward_hook(lambda m, inp, out: store.__setitem__("m", out)))
    for i in range(n):
        hs.append(unet.up_blocks[i].register_forward_hook(
            lambda m, inp, out, k=f"u{i}": store.__setitem__(k, out)))
    return hs


def count_params_analytic(cfg: UNetConfig) -> int:
    """Parameter total without allocating the model (meta device)."""
    with torch.device("meta"):
        m = UNet2DConditionRef(cfg)
    return sum(p.numel() for p in m.parameters())
