"""The reference batch-dict entry of `training_step` (train_sdxl_zh.py:305-396): everything between the dataloader's
`collate_fn` dictionary (utils/custom_dataset_sdxl.py:384-409) and the KD hot path, on the HIP tape.

  batch["pixel_values"]            fp32 [B,3,H,W]   -> VAE encode, `.latent_dist.sample() * scaling_factor`      (:306-309)
  batch["input_ids"] / ["input_ids_uncond"]  int64 [B,L]  -> student text tower, per-token states               (:327-329)
  batch["texts_en"]                the teacher's English captions -> CLIP-L + OpenCLIP-bigG `hidden_states[-2]`,
                                   pooled output of the second tower, same for the empty negative prompt (:410, :170-285)
  batch["original_size"], ["crops_coords_top_left"], ["bucket_id"] -> add_time_ids = (original | crop | BUCKETS[id]) (:386-390)
  batch["zh_or_not"]               int [B]
plus the step's own random draws: noise (+ noise_offset), timesteps, the CFG-dropout mask (uncond = 0.1).

Tokenisers are host-side string processing and stay with the caller: `texts_en` is accepted either as strings together
with `tokenize_en(list[str]) -> (ids_clip_l, ids_bigg)` or pre-tokenised as batch["texts_en_ids"] = (ids_1, ids_2)
([B,77] int64 each) and batch["neg_en_ids"] (the tokenised empty prompt, [1,77] or [B,77] each)."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple

import torch

from ._lib import PeaError

# utils/custom_dataset_sdxl.py:30 -- (height, width) of the nine aspect buckets; `target_size` of add_time_ids
BUCKETS = [[448, 896], [448, 832], [512, 768], [576, 704], [640, 640], [704, 576], [768, 512], [832, 448], [896, 448]]


def add_time_ids_from_batch(batch: Dict, device, buckets: Sequence[Sequence[int]] = BUCKETS) -> torch.Tensor:
    """`torch.cat([original_size, crops_coords_top_left, target_size], 1)` with
    `target_size = [BUCKETS[bucket_id]] * B` (train_sdxl_zh.py:386-389) -> fp32 [B,6]"""
    osz = torch.as_tensor(batch["original_size"]).to(device).reshape(-1, 2)
    crop = torch.as_tensor(batch["crops_coords_top_left"]).to(device).reshape(-1, 2)
    bid = int(torch.as_tensor(batch["bucket_id"]).reshape(-1)[0])
    if not 0 <= bid < len(buckets):
        raise PeaError(f"bucket_id {bid} outside the {len(buckets)} buckets")
    tgt = torch.tensor([list(buckets[bid])] * osz.shape[0], device=device)
    return torch.cat([osz, crop, tgt], 1).to(torch.float32)


class PEAFrontEnd:
    """The frozen modules in front of the KD step: `vae`, the teacher's two CLIP text towers, the student text tower
    (HipVAEEncoder / HipTextEncoder, or anything with the same call surface)."""

    def __init__(self, vae, text_encoder_1, text_encoder_2, student_text_encoder, noise_offset: float = 0.5,
                 uncond: float = 0.1, num_train_timesteps: int = 1000,
                 tokenize_en: Optional[Callable[[Sequence[str]], Tuple[torch.Tensor, torch.Tensor]]] = None,
                 buckets: Sequence[Sequence[int]] = BUCKETS):
        self.vae, self.te1, self.te2, self.zh = vae, text_encoder_1, text_encoder_2, student_text_encoder
        self.noise_offset, self.uncond, self.T = noise_offset, uncond, num_train_timesteps
        self.tokenize_en, self.buckets = tokenize_en, buckets

    def _teacher_ids(self, batch, B):
        if "texts_en_ids" in batch:
            ids1, ids2 = batch["texts_en_ids"]
        elif self.tokenize_en is not None:
            ids1, ids2 = self.tokenize_en(list(batch["texts_en"]))
        else:
            raise PeaError("batch['texts_en'] needs a tokenizer: pass tokenize_en=... or batch['texts_en_ids'] = (ids_1, ids_2)")
        if "neg_en_ids" in batch:
            n1, n2 = batch["neg_en_ids"]
        elif self.tokenize_en is not None:
            n1, n2 = self.tokenize_en([""] * B)
        else:
            raise PeaError("batch['neg_en_ids'] (the tokenised empty prompt) is required without a tokenizer")
        exp = lambda t: t if t.shape[0] == B else t.expand(B, -1)
        return ids1, ids2, exp(n1), exp(n2)

    def encode_prompt(self, ids1, ids2, neg1, neg2):
        """`encode_prompt` of train_sdxl_zh.py:170-285 on token ids: prompt and negative prompt go through each tower as
        ONE batch of 2B rows.  -> prompt_embeds [B,77,w1+w2], negative_prompt_embeds, pooled_prompt_embeds [B,proj]"""
        B = ids1.shape[0]
        h1, _ = self.te1.encode(torch.cat([ids1, neg1]), hidden_index=-2)
        h2, pooled = self.te2.encode(torch.cat([ids2, neg2]), hidden_index=-2)
        pe = torch.cat([h1, h2], -1)
        return pe[:B], pe[B:], pooled[:B]

    def prepare(self, batch: Dict, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
        """reference batch dict -> the post-encoder batch of PEATrainer.training_step.  Keys `_noise`, `_timesteps`,
        `_prompt_mask`, `_vae_noise` override the random draws (parity tests feed the oracle the same values)."""
        dev = self.vae.device
        px = batch["pixel_values"].to(dev, torch.float32)
        B = px.shape[0]
        latents = self.vae.encode_latents(px, noise=batch.get("_vae_noise"), generator=generator)       # :306-309
        if "_noise" in batch:
            noise = batch["_noise"].to(dev, torch.float32)
        else:
            noise = torch.randn(latents.shape, device=dev, generator=generator)                         # :311-315
            if self.noise_offset:
                noise = noise + self.noise_offset * torch.randn(B, latents.shape[1], 1, 1, device=dev, generator=generator)
        t = batch["_timesteps"].to(dev) if "_timesteps" in batch else \
            torch.randint(0, self.T, (B,), device=dev, generator=generator)                             # :318-319
        ids = batch["input_ids"].to(dev)
        ids_u = batch["input_ids_uncond"].to(dev)
        if ids_u.shape[0] != B:
            ids_u = ids_u.expand(B, -1)
        enc2, _ = self.zh.encode_text(torch.cat([ids, ids_u]))                                          # :327-329
        ids1, ids2, n1, n2 = self._teacher_ids(batch, B)
        pe, npe, pooled = self.encode_prompt(ids1.to(dev), ids2.to(dev), n1.to(dev), n2.to(dev))        # :410
        pm = batch["_prompt_mask"].to(dev) if "_prompt_mask" in batch else \
            (torch.rand(B, device=dev, generator=generator) < self.uncond)                              # :392-394
        return {"latents": latents, "noise": noise, "timesteps": t, "enc": enc2[:B], "enc_uncond": enc2[B:],
                "prompt_mask": pm, "zh_or_not": torch.as_tensor(batch["zh_or_not"]).to(dev),
                "teacher_ehs": pe, "teacher_neg": npe, "teacher_pooled": pooled,
                "time_ids": add_time_ids_from_batch(batch, dev, self.buckets)}
