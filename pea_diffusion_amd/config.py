"""UNet2DConditionModel configurations (diffusers 0.23 field names) and the ctypes mirror of
`pea_unet_config` (include/pea_hip.h).  The reference never states these numbers itself: it loads
them with `UNet2DConditionModel.from_pretrained(args.model_path, subfolder="unet")`
(train_sdxl_zh.py:138,151); the values below are the published SDXL-base / SD1.5 / SSD-1B configs."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple, Union


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    sample_size: int = 128
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D")
    up_block_types: Tuple[str, ...] = ("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D")
    layers_per_block: int = 2
    # per down block: an int (every attention of the block) or one int per attention (diffusers >= 0.22, e.g. SSD-1B)
    transformer_layers_per_block: Tuple[Union[int, Tuple[int, ...]], ...] = (1, 2, 10)
    # per UP block (up order), int or one per attention (layers_per_block + 1); None = the down list reversed
    reverse_transformer_layers_per_block: Optional[Tuple[Union[int, Tuple[int, ...]], ...]] = None
    mid_block_type: Optional[str] = "UNetMidBlock2DCrossAttn"    # None: the UNet has no mid block
    num_attention_heads: Tuple[int, ...] = (5, 10, 20)   # diffusers: `attention_head_dim`
    cross_attention_dim: int = 2048
    use_linear_projection: bool = True
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    addition_embed_type: Optional[str] = "text_time"
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    name: str = "sdxl"

    @property
    def pooled_dim(self) -> int:
        return self.projection_class_embeddings_input_dim - 6 * self.addition_time_embed_dim


def sdxl_config() -> UNetConfig:
    return UNetConfig()


def ssd1b_config() -> UNetConfig:
    """segmind/SSD-1B `unet/config.json` (the downstream UNet of tests/test_sdxl_zh.py:449-454; README.md:58) [ext,
    recalled -- no network in the build image]: SDXL widths, layer-pruned attention stacks given per position
    (`transformer_layers_per_block` [0, [2, 2], [4, 4]] on the way down, `reverse_transformer_layers_per_block`
    [[4, 4, 10], [2, 1, 1], 0] on the way up) and no mid block.  Known answer: 1.3 B parameters (model card);
    this layout has 1 300 195 844 (tests/test_oracle_unet.py).  `unet_config_from_diffusers()` reads the real file."""
    return UNetConfig(transformer_layers_per_block=(1, (2, 2), (4, 4)),
                      reverse_transformer_layers_per_block=((4, 4, 10), (2, 1, 1), 1),
                      mid_block_type=None, name="ssd1b")


def ssd1b_uniform_config() -> UNetConfig:
    """round-1 stand-in (uniform depths 1/2/4 with a mid block); kept as an asymmetric shape case"""
    return UNetConfig(transformer_layers_per_block=(1, 2, 4), name="ssd1b_uniform")


def depth_tables(cfg):
    """(down[i][j], up[i][j], mid) transformer depths from the diffusers fields, following UNet2DConditionModel.__init__
    (diffusers >= 0.22 [ext]): ints are broadcast over a block's attentions, the up path defaults to the reversed down
    list, the mid block takes the LAST down entry (element [0] when that is a list: UNetMidBlock2DCrossAttn indexes
    its depth list by attention, and it has one)."""
    n = len(cfg.block_out_channels)
    lpb = cfg.layers_per_block
    tl = cfg.transformer_layers_per_block
    if isinstance(tl, int):
        tl = (tl,) * n
    tl = tuple(tl)
    rev = getattr(cfg, "reverse_transformer_layers_per_block", None)
    if rev is None:
        if any(not isinstance(t, int) for t in tl):
            raise ValueError("reverse_transformer_layers_per_block is required when transformer_layers_per_block is nested")
        rev = tuple(reversed(tl))
    if isinstance(rev, int):
        rev = (rev,) * n

    def row(v, k):
        r = [int(v)] * k if isinstance(v, int) else [int(a) for a in v]
        if len(r) != k:
            raise ValueError(f"transformer depth list {v} must have {k} entries")
        return r
    down = [row(tl[i], lpb) for i in range(n)]
    up = [row(rev[i], lpb + 1) for i in range(n)]
    if getattr(cfg, "mid_block_type", "UNetMidBlock2DCrossAttn") is None:
        mid = -1
    else:
        last = tl[-1]
        mid = int(last) if isinstance(last, int) else int(last[0])     # diffusers: UNetMidBlock2DCrossAttn indexes [i], i < num_layers = 1
    return down, up, mid


def unet_config_from_diffusers(d: dict, name: str = "from_json") -> UNetConfig:
    """UNetConfig from a diffusers `unet/config.json` dict (what `UNet2DConditionModel.from_pretrained(model_id,
    subfolder="unet")` reads, train_sdxl_zh.py:138,151 / tests/test_sdxl_zh.py:144-146)."""
    def tup(v):
        return tuple(tup(a) if isinstance(a, (list, tuple)) else a for a in v) if isinstance(v, (list, tuple)) else v
    n = len(d["block_out_channels"])
    heads = d.get("num_attention_heads") or d.get("attention_head_dim")
    if isinstance(heads, int):
        heads = (heads,) * n
    tl = d.get("transformer_layers_per_block", 1)
    if isinstance(tl, int):
        tl = (tl,) * n
    return UNetConfig(
        in_channels=d.get("in_channels", 4), out_channels=d.get("out_channels", 4), sample_size=d.get("sample_size", 128),
        block_out_channels=tuple(d["block_out_channels"]), down_block_types=tuple(d["down_block_types"]),
        up_block_types=tuple(d["up_block_types"]), layers_per_block=d.get("layers_per_block", 2),
        transformer_layers_per_block=tup(tl), reverse_transformer_layers_per_block=tup(d.get("reverse_transformer_layers_per_block")),
        mid_block_type=d.get("mid_block_type", "UNetMidBlock2DCrossAttn"), num_attention_heads=tuple(heads),
        cross_attention_dim=d.get("cross_attention_dim", 1280), use_linear_projection=bool(d.get("use_linear_projection", False)),
        norm_num_groups=d.get("norm_num_groups", 32), norm_eps=d.get("norm_eps", 1e-5),
        addition_embed_type=d.get("addition_embed_type"), addition_time_embed_dim=d.get("addition_time_embed_dim") or 0,
        projection_class_embeddings_input_dim=d.get("projection_class_embeddings_input_dim") or 0, name=name)


def sd15_config() -> UNetConfig:
    return UNetConfig(
        sample_size=64, block_out_channels=(320, 640, 1280, 1280),
        down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
        up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
        transformer_layers_per_block=(1, 1, 1, 1), num_attention_heads=(8, 8, 8, 8),
        cross_attention_dim=768, use_linear_projection=False, addition_embed_type=None,
        addition_time_embed_dim=0, projection_class_embeddings_input_dim=0, name="sd15")


def tiny_config() -> UNetConfig:
    return UNetConfig(sample_size=16, block_out_channels=(64, 128, 128), transformer_layers_per_block=(1, 1, 2),
                      num_attention_heads=(1, 2, 2), cross_attention_dim=128, addition_time_embed_dim=32,
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


class CUNetConfig(ctypes.Structure):
    _fields_ = [("in_channels", ctypes.c_int), ("out_channels", ctypes.c_int), ("n_levels", ctypes.c_int),
                ("block_out", ctypes.c_int * 4), ("down_cross", ctypes.c_int * 4), ("up_cross", ctypes.c_int * 4),
                ("layers_per_block", ctypes.c_int), ("depth", ctypes.c_int * 4), ("heads", ctypes.c_int * 4),
                ("cross_dim", ctypes.c_int), ("linear_proj", ctypes.c_int), ("groups", ctypes.c_int),
                ("eps", ctypes.c_float), ("text_time", ctypes.c_int), ("add_time_dim", ctypes.c_int),
                ("proj_in_dim", ctypes.c_int), ("per_layer_depth", ctypes.c_int),
                ("depth_down", (ctypes.c_int * 4) * 4), ("depth_up", (ctypes.c_int * 4) * 4), ("depth_mid", ctypes.c_int)]


def to_c(cfg) -> CUNetConfig:
    """accepts this module's UNetConfig or any object with the same (diffusers) field names."""
    n = len(cfg.block_out_channels)
    c = CUNetConfig()
    c.in_channels, c.out_channels, c.n_levels = cfg.in_channels, cfg.out_channels, n
    for i in range(n):
        c.block_out[i] = cfg.block_out_channels[i]
        c.down_cross[i] = int(cfg.down_block_types[i].startswith("CrossAttn"))
        c.up_cross[i] = int(cfg.up_block_types[i].startswith("CrossAttn"))
        c.heads[i] = cfg.num_attention_heads[i]
    down, up, mid = depth_tables(cfg)
    c.per_layer_depth, c.depth_mid = 1, mid
    for i in range(n):
        c.depth[i] = max(down[i])
        for j, v in enumerate(down[i]):
            c.depth_down[i][j] = v
        for j, v in enumerate(up[i]):
            c.depth_up[i][j] = v
    c.layers_per_block = cfg.layers_per_block
    c.cross_dim = cfg.cross_attention_dim
    c.linear_proj = int(cfg.use_linear_projection)
    c.groups, c.eps = cfg.norm_num_groups, cfg.norm_eps
    c.text_time = int(cfg.addition_embed_type == "text_time")
    c.add_time_dim = cfg.addition_time_embed_dim
    c.proj_in_dim = cfg.projection_class_embeddings_input_dim
    return c


# ---- VAE encoder (AutoencoderKL.encode, train_sdxl_zh.py:137,306-309)
@dataclass
class VAEConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.13025
    sample_size: int = 1024
    name: str = "sdxl_vae"


def sdxl_vae_config() -> VAEConfig:
    return VAEConfig()


def sd15_vae_config() -> VAEConfig:
    return VAEConfig(scaling_factor=0.18215, sample_size=512, name="sd15_vae")


def tiny_vae_config() -> VAEConfig:
    return VAEConfig(block_out_channels=(64, 128, 128), sample_size=64, name="tiny_vae")


def vae_to_c(cfg) -> CUNetConfig:
    c = CUNetConfig()
    n = len(cfg.block_out_channels)
    c.in_channels, c.out_channels, c.n_levels = cfg.in_channels, 2 * cfg.latent_channels, n
    for i in range(n):
        c.block_out[i] = cfg.block_out_channels[i]
    c.layers_per_block = cfg.layers_per_block
    c.groups, c.eps = cfg.norm_num_groups, cfg.norm_eps
    return c


def vae_encoder_flops(cfg: VAEConfig, H: int, W: int) -> float:
    """analytic FLOPs of one encode per image (2 x MACs of the convs, linears and the mid-block attention)"""
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
    fl += 2 * (2.0 * h * w * 9 * 2 * c * c)
    fl += 4 * 2.0 * h * w * c * c + 2 * 2.0 * (h * w) ** 2 * c
    fl += 2.0 * h * w * 9 * c * 2 * cfg.latent_channels
    return fl


def vae_decoder_to_c(cfg, out_channels: int = 3) -> CUNetConfig:
    c = CUNetConfig()
    n = len(cfg.block_out_channels)
    c.in_channels, c.out_channels, c.n_levels = cfg.latent_channels, out_channels, n
    for i in range(n):
        c.block_out[i] = cfg.block_out_channels[i]
    c.layers_per_block = cfg.layers_per_block
    c.groups, c.eps = cfg.norm_num_groups, cfg.norm_eps
    return c


# ---- text encoders (train_sdxl_zh.py:103-107,147-150)
@dataclass
class TextConfig:
    vocab_size: int = 49408
    max_position_embeddings: int = 77
    hidden_size: int = 768
    num_attention_heads: int = 12
    num_hidden_layers: int = 12
    intermediate_size: int = 3072
    hidden_act: str = "quick_gelu"          # "quick_gelu" | "gelu"
    flavor: str = "clip"                    # "clip" (pre-LN, causal, EOS pooling) | "bert" (post-LN, padding mask) | "t5" (encoder stack)
    projection_dim: int = 0
    layer_norm_eps: float = 1e-5
    eos_token_id: int = -1                  # clip: EOS id (-1: argmax of the ids, the original CLIP vocabulary); bert: pad id
    position_offset: int = 0                # RoBERTa / XLM-R: 2 (position = padding_idx + 1 + index)
    relative_attention_num_buckets: int = 32    # t5 only
    relative_attention_max_distance: int = 128  # t5 only
    name: str = "clip_l"


def clip_l_config() -> TextConfig:           # SDXL text_encoder (CLIP ViT-L/14 text tower)
    return TextConfig()


def openclip_bigg_config() -> TextConfig:    # SDXL text_encoder_2 (OpenCLIP ViT-bigG/14 text tower, CLIPTextModelWithProjection)
    return TextConfig(hidden_size=1280, num_attention_heads=20, num_hidden_layers=32, intermediate_size=5120,
                      hidden_act="gelu", projection_dim=1280, name="openclip_bigg")


def cnclip_bert_large_config() -> TextConfig:   # Chinese-CLIP ViT-H/14 text tower (RoBERTa-wwm-ext-large), 52 tokens
    return TextConfig(vocab_size=21128, max_position_embeddings=512, hidden_size=1024, num_attention_heads=16,
                      num_hidden_layers=24, intermediate_size=4096, hidden_act="gelu", flavor="bert",
                      layer_norm_eps=1e-12, eos_token_id=0, name="cnclip_bert_large")


def xlm_roberta_large_config() -> TextConfig:   # text tower of xlm-roberta-large-ViT-H-14 (mul_clip) and of AltCLIP
    return TextConfig(vocab_size=250002, max_position_embeddings=514, hidden_size=1024, num_attention_heads=16,
                      num_hidden_layers=24, intermediate_size=4096, hidden_act="gelu", flavor="bert", layer_norm_eps=1e-5,
                      eos_token_id=1, position_offset=2, name="xlm_roberta_large")


def tiny_xlmr_config() -> TextConfig:
    return TextConfig(vocab_size=1000, max_position_embeddings=66, hidden_size=128, num_attention_heads=2,
                      num_hidden_layers=2, intermediate_size=512, hidden_act="gelu", flavor="bert", layer_norm_eps=1e-5,
                      eos_token_id=1, position_offset=2, name="tiny_xlmr")


def mt5_xl_config() -> TextConfig:
    """google/mt5-xl encoder (`T5EncoderModel.from_pretrained('mt5-xl')`, train_sdxl_zh.py:108-112): d_model 2048,
    32 heads x d_kv 64, d_ff 5120, 24 blocks, gated gelu_new, pad id 0; the trainer tokenises to 77 tokens (:333-339)"""
    return TextConfig(vocab_size=250112, max_position_embeddings=77, hidden_size=2048, num_attention_heads=32,
                      num_hidden_layers=24, intermediate_size=5120, hidden_act="gelu_new", flavor="t5", layer_norm_eps=1e-6,
                      eos_token_id=0, name="mt5_xl")


def tiny_t5_config() -> TextConfig:
    return TextConfig(vocab_size=1000, max_position_embeddings=40, hidden_size=128, num_attention_heads=3,
                      num_hidden_layers=2, intermediate_size=256, hidden_act="gelu_new", flavor="t5", layer_norm_eps=1e-6,
                      eos_token_id=0, relative_attention_num_buckets=8, relative_attention_max_distance=20, name="tiny_t5")


def tiny_clip_config() -> TextConfig:
    return TextConfig(vocab_size=1000, max_position_embeddings=77, hidden_size=128, num_attention_heads=2,
                      num_hidden_layers=3, intermediate_size=512, projection_dim=64, eos_token_id=999, name="tiny_clip")


def tiny_bert_config() -> TextConfig:
    return TextConfig(vocab_size=1000, max_position_embeddings=64, hidden_size=128, num_attention_heads=2,
                      num_hidden_layers=2, intermediate_size=512, hidden_act="gelu", flavor="bert", layer_norm_eps=1e-12,
                      eos_token_id=0, name="tiny_bert")


class CTextConfig(ctypes.Structure):
    _fields_ = [("vocab", ctypes.c_int), ("max_pos", ctypes.c_int), ("width", ctypes.c_int), ("heads", ctypes.c_int),
                ("layers", ctypes.c_int), ("intermediate", ctypes.c_int), ("act", ctypes.c_int), ("flavor", ctypes.c_int),
                ("proj_dim", ctypes.c_int), ("eps", ctypes.c_float), ("pos_offset", ctypes.c_int), ("eos_id", ctypes.c_longlong),
                ("rel_buckets", ctypes.c_int), ("rel_max_dist", ctypes.c_int)]


def text_to_c(cfg) -> CTextConfig:
    c = CTextConfig()
    c.vocab, c.max_pos, c.width = cfg.vocab_size, cfg.max_position_embeddings, cfg.hidden_size
    c.heads, c.layers, c.intermediate = cfg.num_attention_heads, cfg.num_hidden_layers, cfg.intermediate_size
    c.act = {"quick_gelu": 3, "gelu": 1, "gelu_new": 1}[cfg.hidden_act]      # t5: the gate's gelu_new is fixed by the flavor
    c.flavor = {"clip": 0, "bert": 1, "t5": 2}[cfg.flavor]
    c.rel_buckets = getattr(cfg, "relative_attention_num_buckets", 32)
    c.rel_max_dist = getattr(cfg, "relative_attention_max_distance", 128)
    c.proj_dim, c.eps, c.eos_id = cfg.projection_dim, cfg.layer_norm_eps, cfg.eos_token_id
    c.pos_offset = getattr(cfg, "position_offset", 0)
    return c
