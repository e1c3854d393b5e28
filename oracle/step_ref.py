"""CPU ORACLE (test infrastructure, NOT product code) -- adapter, DDPM add_noise, loss
composition and the whole KD training step, restated in plain torch fp32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Follows (reference file:line):
  * MLP adapter SDXL      train_sdxl_zh.py:43-67 (twin tests/test_sdxl_zh.py:59-84)
  * MLP adapter SD1.5     train_sd_zh.py:41-56
  * noise + add_noise     train_sdxl_zh.py:311-323 (DDPMScheduler scaled_linear, :140)
  * CFG-dropout `where`   train_sdxl_zh.py:392-395, teacher :413
  * loss composition      train_sdxl_zh.py:399-441 ; SD1.5 twin train_sd_zh.py:217-276
Pinned by tests/golden/*.npz captured from the reference's own MLP / training_step
(oracle/make_golden.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

try:
    from .bf16_store import st as _st      # identity unless inside `with oracle.bf16_store.bf16_storage():`
except ImportError:                        # pragma: no cover
    from bf16_store import st as _st


def _bf16_on() -> bool:
    try:
        from .bf16_store import enabled
    except ImportError:                    # pragma: no cover
        from bf16_store import enabled
    return enabled()


class AdapterRef(nn.Module):
    """Restates `MLP` (train_sdxl_zh.py:43-67).  out_dim1=None gives the SD1.5 variant
    (train_sd_zh.py:41-56) that returns only the token tensor."""

    def __init__(self, in_dim=1024, out_dim=1280, hidden_dim=2048, out_dim1: Optional[int] = 2048,
                 use_residual=False):
        super().__init__()
        self.layernorm = nn.LayerNorm(in_dim)
        self.projector = nn.Sequential(
            nn.Linear(in_dim, hidden_dim, bias=False), nn.GELU(),
            nn.Linear(hidden_dim, hidden_dim, bias=False), nn.GELU(),
            nn.Linear(hidden_dim, out_dim, bias=False))
        self.fc = nn.Linear(out_dim, out_dim1) if out_dim1 is not None else None
        self.use_residual = use_residual

    def forward(self, x):
        residual = x
        if _st is not None and _bf16_on():
            # bf16-storage mode: the HIP adapter keeps the normalised rows, every pre-activation and every activation in bf16
            h = _st(self.layernorm(x))
            for m in self.projector:
                h = _st(m(h))
            x = h
        else:
            x = self.projector(self.layernorm(x))
        if self.fc is None:
            return x
        x2 = _st(self.fc(_st(F.gelu(x))))
        if self.use_residual:
            x = _st(x + residual)
        return _st(x.mean(1)), x2


def ddpm_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012) -> torch.Tensor:
    """DDPMScheduler(beta_schedule="scaled_linear") as built at train_sdxl_zh.py:140."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(x0: torch.Tensor, noise: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """`noise_scheduler.add_noise` (train_sdxl_zh.py:322)."""
    ac = ddpm_alphas_cumprod().to(x0.dtype)
    a = ac[t] ** 0.5
    s = (1 - ac[t]) ** 0.5
    return a.view(-1, 1, 1, 1) * x0 + s.view(-1, 1, 1, 1) * noise


def kd_losses(noise_pred, noise, noise_pred_teacher, taps_s: List[torch.Tensor],
              taps_t: List[torch.Tensor], zh_or_not: torch.Tensor, nan_guard: bool = False):
    """Loss composition of train_sdxl_zh.py:399-441.  Returns (total, loss_noise,
    loss_logits, loss_features).  `nan_guard` = SD1.5 behaviour (train_sd_zh.py:246-268):
    a feature term containing NaN/Inf is skipped."""
    z = zh_or_not.to(noise_pred.dtype).view(-1, 1, 1, 1)
    l_noise = (F.mse_loss(noise_pred, noise, reduction="none") * z).mean([1, 2, 3]).mean()
    l_logit = (F.mse_loss(noise_pred, noise_pred_teacher, reduction="none") * (1 - z)).mean([1, 2, 3]).mean()
    l_feat = 0
    for fs, ft in zip(taps_s, taps_t):
        term = F.mse_loss(ft, fs, reduction="none") * (1 - z)
        if nan_guard and (torch.isinf(term).any() or torch.isnan(term).any()):
            continue
        l_feat = l_feat + term.mean([1, 2, 3]).mean()
    total = l_noise + l_logit + 0.1 * l_feat
    return total, l_noise, l_logit, l_feat


def training_step_ref(adapter: AdapterRef, unet_s, unet_t, batch: Dict[str, torch.Tensor],
                      tap_fn, nan_guard: bool = False):
    """The post-encoder hot path of `training_step` (train_sdxl_zh.py:311-441).

    batch keys (synthetic post-encoder form): latents, noise (offset already
    applied), timesteps, enc [B,L,in], enc_uncond, prompt_mask [B] bool, zh_or_not [B],
    teacher_ehs, teacher_neg [B,77,ctx], teacher_pooled [B,pooled] (SDXL), time_ids [B,6].
    `tap_fn(unet, store)` installs feature taps.  Returns dict of scalars + tensors.
    """
    x_t = add_noise(batch["latents"], batch["noise"], batch["timesteps"])
    sdxl = adapter.fc is not None
    if sdxl:
        pooled, ehs = adapter(batch["enc"])
        _unused_pooled_uncond, ehs_u = adapter(batch["enc_uncond"])      # :383-384 (pooled_uncond unused)
        added_s = {"text_embeds": pooled, "time_ids": batch["time_ids"]}
        added_t = {"text_embeds": batch["teacher_pooled"], "time_ids": batch["time_ids"]}
    else:
        ehs, ehs_u = adapter(batch["enc"]), adapter(batch["enc_uncond"])
        added_s = added_t = None
    m = batch["prompt_mask"].view(-1, 1, 1)
    ehs = torch.where(m, ehs_u, ehs)                                        # :395
    ks, kt = {}, {}
    hs = tap_fn(unet_s, ks)
    ht = tap_fn(unet_t, kt)
    try:
        pred = unet_s(x_t, batch["timesteps"], ehs, added_cond_kwargs=added_s, return_dict=False)[0]
        with torch.no_grad():
            t_ehs = torch.where(m, batch["teacher_neg"], batch["teacher_ehs"])  # :413
            pred_t = unet_t(x_t, batch["timesteps"], t_ehs, added_cond_kwargs=added_t, return_dict=False)[0]
    finally:
        for h in hs + ht:
            h.remove()
    names = list(ks.keys())
    total, l0, l1, l2 = kd_losses(pred, batch["noise"], pred_t, [ks[k] for k in names],
                                  [kt[k] for k in names], batch["zh_or_not"], nan_guard)
    return {"loss": total, "train_loss": l0, "train_loss_logits": l1, "train_loss_features": l2,
            "noise_pred": pred, "noise_pred_teacher": pred_t, "taps_s": ks, "taps_t": kt, "ehs": ehs}


def rescale_noise_cfg_ref(noise_cfg, noise_pred_text, guidance_rescale=0.0):
    """tests/test_sdxl_zh.py:45-56."""
    std_text = noise_pred_text.std(dim=list(range(1, noise_pred_text.ndim)), keepdim=True)
    std_cfg = noise_cfg.std(dim=list(range(1, noise_cfg.ndim)), keepdim=True)
    rescaled = noise_cfg * (std_text / std_cfg)
    return guidance_rescale * rescaled + (1 - guidance_rescale) * noise_cfg


def synthetic_batch(cfg, B: int, L: int = 77, enc_dim: int = 1024, seed: int = 0,
                    latent_hw=None, force_mask: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded synthetic post-encoder batch; identical tensors feed the
    oracle and the device path."""
    g = lambda s: torch.Generator().manual_seed(seed * 100 + s)
    hw = latent_hw or cfg.sample_size
    lh, lw = (hw, hw) if isinstance(hw, int) else hw           # (height, width): the reference's aspect-ratio buckets
    lat = torch.randn(B, 4, lh, lw, generator=g(0))
    noise = torch.randn(B, 4, lh, lw, generator=g(1))
    if cfg.addition_embed_type == "text_time":                         # SDXL: noise_offset 0.5 (universal.py:31)
        noise = noise + 0.5 * torch.randn(B, 4, 1, 1, generator=g(11))
    t = torch.randint(0, 1000, (B,), generator=g(2))
    enc = torch.randn(B, L, enc_dim, generator=g(3))
    enc_u = torch.randn(1, L, enc_dim, generator=g(13)).repeat(B, 1, 1)
    ctx = cfg.cross_attention_dim
    te = torch.randn(B, 77, ctx, generator=g(4))
    tn = torch.randn(1, 77, ctx, generator=g(14)).repeat(B, 1, 1)
    pm = torch.rand(B, generator=g(5)) < 0.1
    if force_mask and B > 1:
        pm[B - 1] = True
    zh = (torch.rand(B, generator=g(6)) < 0.5).to(torch.int64)
    if B >= 4:
        zh[:4] = torch.tensor([1, 0, 0, 1])
    elif B >= 2:
        zh[:2] = torch.tensor([1, 0])
    else:
        zh[:] = 0
    out = dict(latents=lat, noise=noise, timesteps=t, enc=enc, enc_uncond=enc_u, prompt_mask=pm,
               zh_or_not=zh, teacher_ehs=te, teacher_neg=tn)
    if cfg.addition_embed_type == "text_time":
        out["teacher_pooled"] = torch.randn(B, cfg.pooled_dim, generator=g(7))
        out["time_ids"] = torch.tensor([[lh * 8, lw * 8, 0, 0, lh * 8, lw * 8]] * B, dtype=torch.int64)
    return out
