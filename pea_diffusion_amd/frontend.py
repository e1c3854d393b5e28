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


class _PinnedRing:
    """Small host tensors (token ids, size / crop lists, flags) go to the device through a ring of pinned slots allocated
    once: a copy from pageable memory blocks the caller until the stream's earlier work is done -- with the GPU a step
    behind the host that serialises the host's enqueueing with the GPU -- and pinning a fresh tensor per call goes through
    the host allocator (measured slower than the blocking copy).  A slot is reused `slots` copies later, after its event."""

    def __init__(self, slots: int = 32, nbytes: int = 1 << 16):
        self.buf = torch.empty(slots, nbytes, dtype=torch.uint8).pin_memory()
        self.events, self.next, self.nbytes = [None] * slots, 0, nbytes

    def copy(self, t: torch.Tensor, device) -> torch.Tensor:
        t = t.contiguous()
        n = t.numel() * t.element_size()
        k, self.next = self.next, (self.next + 1) % len(self.events)
        if self.events[k] is not None:
            self.events[k].synchronize()
        view = self.buf[k, :n].view(t.dtype).view(t.shape)
        view.copy_(t)
        out = view.to(device, non_blocking=True)
        self.events[k] = torch.cuda.Event()
        self.events[k].record(torch.cuda.current_stream(device))
        return out


_ring = None
ASYNC_COPIES = True      # False: plain (blocking) copies of pageable host tensors, for an A/B


def _to_device(t, device, dtype=None) -> torch.Tensor:
    """host -> device without stalling the host where that is possible: device tensors pass through, pinned tensors (a
    DataLoader with pin_memory=True) are copied asynchronously, small pageable ones go through the pinned ring."""
    global _ring
    t = torch.as_tensor(t)
    if ASYNC_COPIES and t.device.type == "cpu" and torch.device(device).type == "cuda":
        if t.is_pinned():
            t = t.to(device, non_blocking=True)
        elif 0 < t.numel() * t.element_size() <= (1 << 16):
            if _ring is None:
                _ring = _PinnedRing()
            t = _ring.copy(t, device)
    return t.to(device=device, dtype=dtype)


def add_time_ids_from_batch(batch: Dict, device, buckets: Sequence[Sequence[int]] = BUCKETS) -> torch.Tensor:
    """`torch.cat([original_size, crops_coords_top_left, target_size], 1)` with
    `target_size = [BUCKETS[bucket_id]] * B` (train_sdxl_zh.py:386-389) -> fp32 [B,6]"""
    osz = torch.as_tensor(batch["original_size"]).reshape(-1, 2)
    crop = torch.as_tensor(batch["crops_coords_top_left"]).reshape(-1, 2)
    bid = int(torch.as_tensor(batch["bucket_id"]).reshape(-1)[0])
    if not 0 <= bid < len(buckets):
        raise PeaError(f"bucket_id {bid} outside the {len(buckets)} buckets")
    tgt = torch.tensor([list(buckets[bid])] * osz.shape[0])
    if osz.device.type != "cpu" or crop.device.type != "cpu":        # already on the device: assemble there
        return torch.cat([osz.to(device), crop.to(device), tgt.to(device)], 1).to(torch.float32)
    return _to_device(torch.cat([osz, crop, tgt], 1).to(torch.float32), device)    # assembled on the host, ONE async copy


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
        # The three text towers run on their own (high-priority) HIP streams beside the VAE encode: at 2B x 77 = 616 rows
        # their ~800 launches are latency-bound and fill a third of the CUs at best, so one after the other on the VAE's
        # stream they cost 8.6 ms of a 34.7 ms front end; concurrently the front end takes 29.7 ms and the whole
        # training_step_from_batch 136.3 instead of 141.8 ms (scripts/full_step_bench.py).
        self.concurrent_towers = True
        self.tower_streams = 3          # streams the three towers are spread over (set before the first prepare)
        # With the towers on their own streams the host first waits for the current stream to drain (one stream
        # synchronise per call, the KD step of the previous batch): measured, enqueueing the towers while the host is
        # still ~1000 launches ahead of the GPU loses more (their launches trickle in at the pace the queue frees up:
        # 145.0 ms per step) than the short idle gap costs (136.3-138.5 ms; 141.8-143.2 with the towers on the VAE stream).
        self.drain_before_enqueue = True
        self._tower_streams = None
        self._tower_outs = None

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
        px = _to_device(batch["pixel_values"], dev, torch.float32)
        B = px.shape[0]
        # every host -> device copy goes through pinned memory (_to_device): the call never waits for the GPU, so the host
        # keeps enqueueing (this call's towers, the KD step behind it) while the GPU works
        time_ids = add_time_ids_from_batch(batch, dev, self.buckets)
        zh_or_not = _to_device(batch["zh_or_not"], dev)
        ids = _to_device(batch["input_ids"], dev)
        ids_u = _to_device(batch["input_ids_uncond"], dev)
        if ids_u.shape[0] != B:
            ids_u = ids_u.expand(B, -1)
        ids1, ids2, n1, n2 = self._teacher_ids(batch, B)
        zh_in = torch.cat([ids, ids_u])
        t1_in = torch.cat([_to_device(ids1, dev), _to_device(n1, dev)])
        t2_in = torch.cat([_to_device(ids2, dev), _to_device(n2, dev)])
        towers = ((self.zh.encode_text, zh_in, {}), (self.te1.encode, t1_in, {"hidden_index": -2}),
                  (self.te2.encode, t2_in, {"hidden_index": -2}))
        side = self.concurrent_towers and torch.device(dev).type == "cuda"
        outs = [None, None, None]
        if side:
            if self.drain_before_enqueue:
                torch.cuda.current_stream(dev).synchronize()
            if self._tower_streams is None:
                n = max(1, min(len(towers), int(self.tower_streams)))
                self._tower_streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(n)]
            main = torch.cuda.current_stream(dev)
            fork = torch.cuda.Event()
            fork.record(main)
        # the VAE's ~60 launches are enqueued first: the GPU works on them while the host enqueues the towers' ~800
        latents = self.vae.encode_latents(px, noise=batch.get("_vae_noise"), generator=generator)       # :306-309
        if side:
            for st in self._tower_streams:
                st.wait_event(fork)
            for i, (fn, arg, kw) in enumerate(towers):
                with torch.cuda.stream(self._tower_streams[i % len(self._tower_streams)]):
                    outs[i] = fn(arg, **kw)
        if "_noise" in batch:
            noise = batch["_noise"].to(dev, torch.float32)
        else:
            noise = torch.randn(latents.shape, device=dev, generator=generator)                         # :311-315
            if self.noise_offset:
                noise = noise + self.noise_offset * torch.randn(B, latents.shape[1], 1, 1, device=dev, generator=generator)
        t = batch["_timesteps"].to(dev) if "_timesteps" in batch else \
            torch.randint(0, self.T, (B,), device=dev, generator=generator)                             # :318-319
        if side:
            for st in self._tower_streams:
                main.wait_stream(st)
            # The tower outputs were allocated on the side streams and are consumed on `main`.  They stay referenced until
            # the next call: their memory then returns to the side streams' pools, whose next use is ordered behind that
            # call's fork event, i.e. behind every consumer enqueued on `main` (what record_stream would arrange through
            # the allocator's event polling).
            self._tower_outs = outs
        else:
            outs = [fn(arg, **kw) for fn, arg, kw in towers]
        (enc2, _), (h1, _), (h2, pooled) = outs                                                         # :327-329, :410
        pe_all = torch.cat([h1, h2], -1)
        pe, npe, pooled = pe_all[:B], pe_all[B:], pooled[:B]
        pm = batch["_prompt_mask"].to(dev) if "_prompt_mask" in batch else \
            (torch.rand(B, device=dev, generator=generator) < self.uncond)                              # :392-394
        return {"latents": latents, "noise": noise, "timesteps": t, "enc": enc2[:B], "enc_uncond": enc2[B:],
                "prompt_mask": pm, "zh_or_not": zh_or_not,
                "teacher_ehs": pe, "teacher_neg": npe, "teacher_pooled": pooled, "time_ids": time_ids}
