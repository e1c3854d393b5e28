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


class _T5Norm(nn.Module):
    """T5LayerNorm: x * rsqrt(mean(x^2) + eps) * weight (no mean subtraction, no bias)"""

    def __init__(self, w, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(w))
        self.eps = eps

    def forward(self, x):
        return self.weight * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps))


def t5_relative_bucket(rel, num_buckets, max_distance):
    """bidirectional bucket of rel = key_pos - query_pos (T5Attention._relative_position_bucket, encoder form): half the
    buckets per sign; within a sign the first half are exact offsets, the rest log-spaced up to max_distance"""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(n < max_exact, n, large)


class _T5Block(nn.Module):
    def __init__(self, w, heads, d_kv, d_ff, eps, buckets):
        super().__init__()
        inner = heads * d_kv
        att = nn.Module()
        att.SelfAttention = nn.Module()
        for n in ("q", "k", "v"):
            setattr(att.SelfAttention, n, nn.Linear(w, inner, bias=False))
        att.SelfAttention.o = nn.Linear(inner, w, bias=False)
        if buckets:
            att.SelfAttention.relative_attention_bias = nn.Embedding(buckets, heads)
        att.layer_norm = _T5Norm(w, eps)
        ff = nn.Module()
        ff.DenseReluDense = nn.Module()
        ff.DenseReluDense.wi_0, ff.DenseReluDense.wi_1 = nn.Linear(w, d_ff, bias=False), nn.Linear(w, d_ff, bias=False)
        ff.DenseReluDense.wo = nn.Linear(d_ff, w, bias=False)
        ff.layer_norm = _T5Norm(w, eps)
        self.layer = nn.ModuleList([att, ff])
        self.heads, self.d_kv = heads, d_kv

    def forward(self, x, bias):
        B, L, _ = x.shape
        a, f = self.layer[0], self.layer[1]
        n = a.layer_norm(x)
        sp = lambda t: t.view(B, L, self.heads, self.d_kv).transpose(1, 2)
        q, k, v = sp(a.SelfAttention.q(n)), sp(a.SelfAttention.k(n)), sp(a.SelfAttention.v(n))
        s = q @ k.transpose(-1, -2) + bias                          # no 1/sqrt(d): folded into T5's initialisation
        o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, self.heads * self.d_kv)
        x = x + a.SelfAttention.o(o)
        n = f.layer_norm(x)
        g = F.gelu(f.DenseReluDense.wi_0(n), approximate="tanh") * f.DenseReluDense.wi_1(n)     # gated gelu_new
        return x + f.DenseReluDense.wo(g)


class T5EncoderRef(nn.Module):
    """the encoder stack of HF `T5EncoderModel` in its T5 v1.1 / mT5 form (gated-gelu FF), the student text tower of
    `--text_encoder mt5` (train_sdxl_zh.py:108-112; call site :331-345 -- `encoder(ids, attention_mask=ids.ne(pad))[0]`).
    keys: shared.weight, encoder.block.N.layer.{0,1}.*, encoder.final_layer_norm.weight (d_kv = 64)"""

    def __init__(self, cfg, d_kv=64):
        super().__init__()
        self.cfg = cfg
        w = cfg.hidden_size
        self.shared = nn.Embedding(cfg.vocab_size, w)
        self.encoder = nn.Module()
        self.encoder.block = nn.ModuleList([
            _T5Block(w, cfg.num_attention_heads, d_kv, cfg.intermediate_size, cfg.layer_norm_eps,
                     cfg.relative_attention_num_buckets if i == 0 else 0) for i in range(cfg.num_hidden_layers)])
        self.encoder.final_layer_norm = _T5Norm(w, cfg.layer_norm_eps)

    def forward(self, ids):
        B, L = ids.shape
        c = self.cfg
        x = self.shared(ids)
        pos = torch.arange(L)
        bucket = t5_relative_bucket(pos[None, :] - pos[:, None], c.relative_attention_num_buckets, c.relative_attention_max_distance)
        bias = self.encoder.block[0].layer[0].SelfAttention.relative_attention_bias(bucket).permute(2, 0, 1)[None]   # [1,H,L,L]
        valid = ids != c.eos_token_id                              # pad id; right padding
        bias = bias + torch.zeros(B, 1, 1, L).masked_fill(~valid[:, None, None, :], float("-inf"))
        hs = [x]
        for blk in self.encoder.block:
            x = blk(x, bias)
            hs.append(x)
        return {"hidden_states": hs, "last_hidden_state": self.encoder.final_layer_norm(x)}
