"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): torch fp32 restatement of the frozen text encoders in front of the
step -- the teacher's `CLIPTextModel` / `CLIPTextModelWithProjection` (transformers==4.31.0, requirements.txt:23; call
sites train_sdxl_zh.py:147-150,170-285) and the BERT text tower of Chinese-CLIP (`cn_clip`, un-pinned private fork;
train_sdxl_zh.py:103-107,327-329).  transformers is third-party and absent from /root/reference, but a release of it
(5.15) IS installed in the authoring container: oracle/make_golden_text.py runs ITS CLIPTextModelWithProjection and
BertModel on seeded tiny configurations and stores weights, ids and outputs in tests/golden/text_*.npz -- this
restatement is pinned against those vectors (tests/test_text_cpu.py)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _ClipLayer(nn.Module):
    def __init__(self, w, heads, inter, act, eps):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(w, eps=eps)
        self.self_attn = nn.ModuleDict({k: nn.Linear(w, w) for k in ("q_proj", "k_proj", "v_proj", "out_proj")})
        self.layer_norm2 = nn.LayerNorm(w, eps=eps)
        self.mlp = nn.ModuleDict({"fc1": nn.Linear(w, inter), "fc2": nn.Linear(inter, w)})
        self.heads, self.act = heads, act

    def forward(self, x, mask):
        B, L, W = x.shape
        h = self.layer_norm1(x)
        sp = lambda t: t.view(B, L, self.heads, W // self.heads).transpose(1, 2)
        q, k, v = (sp(self.self_attn[n](h)) for n in ("q_proj", "k_proj", "v_proj"))
        s = q @ k.transpose(-1, -2) / math.sqrt(W // self.heads) + mask
        a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, W)
        x = x + self.self_attn["out_proj"](a)
        h = self.mlp["fc1"](self.layer_norm2(x))
        h = h * torch.sigmoid(1.702 * h) if self.act == "quick_gelu" else F.gelu(h)
        return x + self.mlp["fc2"](h)


class CLIPTextRef(nn.Module):
    """keys as HF: text_model.embeddings.{token,position}_embedding, text_model.encoder.layers.N.*, text_model.final_layer_norm,
    text_projection"""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        w = cfg.hidden_size
        tm = nn.Module()
        tm.embeddings = nn.Module()
        tm.embeddings.token_embedding = nn.Embedding(cfg.vocab_size, w)
        tm.embeddings.position_embedding = nn.Embedding(cfg.max_position_embeddings, w)
        tm.encoder = nn.Module()
        tm.encoder.layers = nn.ModuleList([_ClipLayer(w, cfg.num_attention_heads, cfg.intermediate_size, cfg.hidden_act,
                                                      cfg.layer_norm_eps) for _ in range(cfg.num_hidden_layers)])
        tm.final_layer_norm = nn.LayerNorm(w, eps=cfg.layer_norm_eps)
        self.text_model = tm
        if cfg.projection_dim:
            self.text_projection = nn.Linear(w, cfg.projection_dim, bias=False)

    def forward(self, ids):
        B, L = ids.shape
        tm = self.text_model
        x = tm.embeddings.token_embedding(ids) + tm.embeddings.position_embedding(torch.arange(L))[None]
        mask = torch.full((L, L), float("-inf")).triu(1)[None, None]
        hs = [x]
        for lyr in tm.encoder.layers:
            x = lyr(x, mask)
            hs.append(x)
        last = tm.final_layer_norm(x)
        if self.cfg.eos_token_id >= 0:
            pos = (ids == self.cfg.eos_token_id).int().argmax(-1)
        else:
            pos = ids.argmax(-1)
        pooled = last[torch.arange(B), pos]
        if self.cfg.projection_dim:
            pooled = self.text_projection(pooled)
        return {"hidden_states": hs, "last_hidden_state": last, "pooled": pooled}


class _BertLayer(nn.Module):
    def __init__(self, w, heads, inter, eps):
        super().__init__()
        self.attention = nn.Module()
        self.attention.self = nn.Module()
        self.attention.self.query, self.attention.self.key, self.attention.self.value = nn.Linear(w, w), nn.Linear(w, w), nn.Linear(w, w)
        self.attention.output = nn.Module()
        self.attention.output.dense, self.attention.output.LayerNorm = nn.Linear(w, w), nn.LayerNorm(w, eps=eps)
        self.intermediate = nn.Module()
        self.intermediate.dense = nn.Linear(w, inter)
        self.output = nn.Module()
        self.output.dense, self.output.LayerNorm = nn.Linear(inter, w), nn.LayerNorm(w, eps=eps)
        self.heads = heads

    def forward(self, x, mask):
        B, L, W = x.shape
        sp = lambda t: t.view(B, L, self.heads, W // self.heads).transpose(1, 2)
        q, k, v = sp(self.attention.self.query(x)), sp(self.attention.self.key(x)), sp(self.attention.self.value(x))
        s = q @ k.transpose(-1, -2) / math.sqrt(W // self.heads) + mask
        a = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, W)
        x = self.attention.output.LayerNorm(x + self.attention.output.dense(a))
        return self.output.LayerNorm(x + self.output.dense(F.gelu(self.intermediate.dense(x))))


class BertTextRef(nn.Module):
    """keys as HF BertModel: embeddings.{word,position,token_type}_embeddings, embeddings.LayerNorm, encoder.layer.N.*"""

    def __init__(self, cfg, type_vocab_size=None):
        super().__init__()
        self.cfg = cfg
        self.pos_offset = getattr(cfg, "position_offset", 0)      # RoBERTa / XLM-R: 2
        type_vocab_size = type_vocab_size or (1 if self.pos_offset else 2)
        w = cfg.hidden_size
        self.embeddings = nn.Module()
        self.embeddings.word_embeddings = nn.Embedding(cfg.vocab_size, w)
        self.embeddings.position_embeddings = nn.Embedding(cfg.max_position_embeddings, w)
        self.embeddings.token_type_embeddings = nn.Embedding(type_vocab_size, w)
        self.embeddings.LayerNorm = nn.LayerNorm(w, eps=cfg.layer_norm_eps)
        self.encoder = nn.Module()
        self.encoder.layer = nn.ModuleList([_BertLayer(w, cfg.num_attention_heads, cfg.intermediate_size, cfg.layer_norm_eps)
                                            for _ in range(cfg.num_hidden_layers)])

    def forward(self, ids):
        B, L = ids.shape
        e = self.embeddings
        x = e.LayerNorm(e.word_embeddings(ids) + e.position_embeddings(torch.arange(L) + self.pos_offset)[None] + e.token_type_embeddings.weight[0])
        valid = ids != self.cfg.eos_token_id                      # pad id; right padding
        mask = torch.zeros(B, 1, 1, L).masked_fill(~valid[:, None, None, :], float("-inf"))
        hs = [x]
        for lyr in self.encoder.layer:
            x = lyr(x, mask)
            hs.append(x)
        return {"hidden_states": hs, "last_hidden_state": x}
