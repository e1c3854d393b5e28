"""`HipTextEncoder`: the frozen text encoders in front of the training step and of the inference loop, on the HIP op
tape.  CLIP flavour = the teacher's two encoders as diffusers' `encode_prompt` calls them
(`out = text_encoder(ids, output_hidden_states=True); pooled = out[0]; prompt_embeds = out.hidden_states[-2]`,
train_sdxl_zh.py:170-285) -- CLIP-L and OpenCLIP-bigG, HF `CLIPTextModel[WithProjection]` state-dict keys.  BERT flavour =
the Chinese-CLIP text tower whose per-token states feed the adapter (`encode_text(batch["input_ids"])`,
train_sdxl_zh.py:327-329; HF `BertModel` keys, an optional `bert.` prefix is stripped).  T5 flavour = the mT5 student
option (`self.text_encoder.encoder(ids, attention_mask=ids.ne(pad), output_hidden_states=True)[0]`, train_sdxl_zh.py:108-112,
331-345; HF `T5EncoderModel` keys; the padding mask is derived from the ids on the device, right padding with the pad id).
Tokenisation stays on the host."""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import config as _cfg
from ._lib import PeaError, check, lib, ptr, stream_ptr
from .unet import HipUNet


class _HiddenStates:
    def __init__(self, enc, ids):
        self._enc, self._ids = enc, ids

    def __getitem__(self, k: int):
        return self._enc.encode(self._ids, hidden_index=k)[0]


class _Output:
    """what `encode_prompt` reads: `out[0]` (pooled / text_embeds for the projection model, else the last state) and
    `out.hidden_states[-2]`"""

    def __init__(self, enc, ids, last, pooled):
        self.last_hidden_state, self.text_embeds, self.pooler_output = last, pooled, pooled
        self.hidden_states = _HiddenStates(enc, ids)

    def __getitem__(self, i):
        return (self.text_embeds if self.text_embeds is not None else self.last_hidden_state, self.last_hidden_state)[i]


class HipTextEncoder:
    def __init__(self, cfg, batch: int, ctx_len: Optional[int] = None):
        if not torch.cuda.is_available():
            raise PeaError("HipTextEncoder needs a MI355X (no CPU fallback)")
        self.cfg, self.config = cfg, cfg
        self.B, self.L = batch, ctx_len or cfg.max_position_embeddings
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.dtype = torch.bfloat16
        self._h = ctypes.c_void_p()
        c = _cfg.text_to_c(cfg)
        check(lib().pea_text_create(ctypes.byref(c), self.B, self.L, ctypes.byref(self._h)))

    __del__ = HipUNet.__del__
    weight_table = HipUNet.weight_table
    memory = HipUNet.memory

    @property
    def encoder(self):
        """`T5EncoderModel.encoder` -- the reference calls the stack directly (train_sdxl_zh.py:341)"""
        return self

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        table = self.weight_table()
        fixed = {}
        for k, v in sd.items():
            k2 = k[5:] if k.startswith("bert.") else k
            if k2 == "encoder.embed_tokens.weight":       # T5: tied to `shared.weight`
                if "shared.weight" in sd:
                    continue
                k2 = "shared.weight"
            if k2 in table:
                fixed[k2] = v[: table[k2][0]] if (k2.endswith("position_embeddings.weight") or k2.endswith("token_type_embeddings.weight")) and v.shape[0] > table[k2][0] else v
        return HipUNet.load_state_dict(self, fixed, strict)

    def init_random(self, seed: int = 0):
        check(lib().pea_unet_init_random(self._h, seed, stream_ptr()))

    def encode(self, input_ids, hidden_index: int = -2):
        """-> (hidden fp32 [B, L, width], pooled fp32 [B, proj] or None).  hidden_index: -1 last state (CLIP: after the
        final LayerNorm), -2 = `hidden_states[-2]`, k >= 0 = `hidden_states[k]`."""
        if tuple(input_ids.shape) != (self.B, self.L):
            raise PeaError(f"HipTextEncoder built for ids {(self.B, self.L)}, got {tuple(input_ids.shape)}")
        ids = input_ids.detach().to(self.device, torch.int64).contiguous()
        hid = torch.empty(self.B, self.L, self.cfg.hidden_size, device=self.device, dtype=torch.float32)
        pooled = None
        if self.cfg.flavor == "clip":
            pooled = torch.empty(self.B, self.cfg.projection_dim or self.cfg.hidden_size, device=self.device, dtype=torch.float32)
        check(lib().pea_text_forward(self._h, ptr(ids), int(hidden_index), ptr(hid), ptr(pooled), stream_ptr()))
        self._keep = ids
        return hid, pooled

    def __call__(self, input_ids, attention_mask=None, output_hidden_states: bool = False, **kw):
        last, pooled = self.encode(input_ids, hidden_index=-1)
        return _Output(self, input_ids, last, pooled)

    def encode_text(self, input_ids):
        """Chinese-CLIP's (privately modified) `encode_text`: per-token states first (train_sdxl_zh.py:327-329); for the T5
        flavour the same call returns `encoder(ids, attention_mask=ids.ne(pad))[0]`"""
        return self.encode(input_ids, hidden_index=-1)[0], None
