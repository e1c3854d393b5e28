"""The text-encoder restatement (oracle/text_ref.py) against golden vectors produced by the installed HF transformers
release (oracle/make_golden_text.py): CLIPTextModelWithProjection and BertModel on seeded tiny configurations."""
import os

import numpy as np
import torch

from oracle.text_ref import BertTextRef, CLIPTextRef, T5EncoderRef
from pea_diffusion_amd import config as pc


def _load(ref, z):
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}
    sd = {k: v for k, v in sd.items() if "position_ids" not in k}
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)


def test_clip_text_restatement_matches_transformers(golden_dir):
    z = np.load(os.path.join(golden_dir, "text_clip.npz"))
    ref = CLIPTextRef(pc.tiny_clip_config())
    _load(ref, z)
    with torch.no_grad():
        o = ref(torch.from_numpy(z["ids"]))
    for i, h in enumerate(o["hidden_states"]):
        assert torch.allclose(h, torch.from_numpy(z[f"hidden_{i}"]), rtol=1e-4, atol=1e-4), i
    assert torch.allclose(o["last_hidden_state"], torch.from_numpy(z["last_hidden_state"]), rtol=1e-4, atol=1e-4)
    assert torch.allclose(o["pooled"], torch.from_numpy(z["text_embeds"]), rtol=1e-4, atol=1e-4)


def test_bert_text_restatement_matches_transformers(golden_dir):
    z = np.load(os.path.join(golden_dir, "text_bert.npz"))
    ref = BertTextRef(pc.tiny_bert_config())
    _load(ref, z)
    ids = torch.from_numpy(z["ids"])
    with torch.no_grad():
        o = ref(ids)
    valid = ids != 0                       # padded positions are don't-care in HF's output as well
    for i, h in enumerate(o["hidden_states"]):
        assert torch.allclose(h[valid], torch.from_numpy(z[f"hidden_{i}"])[valid], rtol=1e-4, atol=1e-4), i


def test_xlm_roberta_restatement_matches_transformers(golden_dir):
    """RoBERTa-family tower (mul_clip's xlm-roberta-large, AltCLIP): BERT layers, positions offset by padding_idx + 1"""
    z = np.load(os.path.join(golden_dir, "text_xlmr.npz"))
    ref = BertTextRef(pc.tiny_xlmr_config())
    _load(ref, z)
    ids = torch.from_numpy(z["ids"])
    with torch.no_grad():
        o = ref(ids)
    valid = ids != 1
    assert torch.allclose(o["last_hidden_state"][valid], torch.from_numpy(z["last_hidden_state"])[valid], rtol=1e-4, atol=1e-4)


def test_t5_encoder_restatement_matches_transformers(golden_dir):
    """mT5-form T5 encoder (RMSNorm, bucketed relative position bias, unscaled scores, gated gelu_new FF) against HF
    T5EncoderModel.encoder(ids, attention_mask=ids.ne(pad), output_hidden_states=True) -- the call of train_sdxl_zh.py:341"""
    z = np.load(os.path.join(golden_dir, "text_t5.npz"))
    cfg = pc.tiny_t5_config()
    ref = T5EncoderRef(cfg)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.") and k != "w.encoder.embed_tokens.weight"}
    assert np.array_equal(z["w.encoder.embed_tokens.weight"], z["w.shared.weight"])
    ref.load_state_dict(sd, strict=True)
    ids = torch.from_numpy(z["ids"])
    with torch.no_grad():
        o = ref(ids)
    valid = ids != 0
    for i in range(cfg.num_hidden_layers):          # HF: hidden_states[i] = input of block i; the last entry is the normed output
        assert torch.allclose(o["hidden_states"][i][valid], torch.from_numpy(z[f"hidden_{i}"])[valid], rtol=1e-4, atol=1e-4), i
    assert torch.allclose(o["last_hidden_state"][valid], torch.from_numpy(z["last_hidden_state"])[valid], rtol=1e-4, atol=1e-4)
    assert np.array_equal(z[f"hidden_{cfg.num_hidden_layers}"], z["last_hidden_state"])


def test_t5_relative_buckets_known_answers():
    """bucket table of the published T5 configuration (32 buckets, max distance 128), bidirectional; expected values are
    the output of transformers' T5Attention._relative_position_bucket(rel, True, 32, 128) captured in the authoring container"""
    from oracle.text_ref import t5_relative_bucket
    rel = torch.tensor([0, 1, -1, 7, 8, -8, 15, 16, 50, 127, 128, 500, -500])
    #  positive offsets start at 16; 0..7 exact, then log-spaced: 8 + floor(log(n/8)/log(16) * 8), capped at 15
    assert t5_relative_bucket(rel, 32, 128).tolist() == [0, 17, 1, 23, 24, 8, 25, 26, 29, 31, 31, 31, 15]


def test_text_parameter_totals():
    """structural known-answers of the published encoders"""
    with torch.device("meta"):
        a = CLIPTextRef(pc.clip_l_config())
        b = CLIPTextRef(pc.openclip_bigg_config())
    assert sum(p.numel() for p in a.parameters()) == 123_060_480          # CLIP ViT-L/14 text model (SDXL text_encoder)
    assert sum(p.numel() for p in b.parameters()) == 694_659_840          # OpenCLIP bigG text model + projection (text_encoder_2)
    with torch.device("meta"):
        t = T5EncoderRef(pc.mt5_xl_config())
    # mt5-xl encoder: 250112*2048 embedding + 24 * (4*2048*2048 + 3*2048*5120 + 2*2048) + 32*32 bias table + final norm
    assert sum(p.numel() for p in t.parameters()) == 250112 * 2048 + 24 * (4 * 2048 * 2048 + 3 * 2048 * 5120 + 2 * 2048) + 32 * 32 + 2048
