"""The text-encoder restatement (oracle/text_ref.py) against golden vectors produced by the installed HF transformers
release (oracle/make_golden_text.py): CLIPTextModelWithProjection and BertModel on seeded tiny configurations."""
import os

import numpy as np
import torch

from oracle.text_ref import BertTextRef, CLIPTextRef
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


def test_text_parameter_totals():
    """structural known-answers of the published encoders"""
    with torch.device("meta"):
        a = CLIPTextRef(pc.clip_l_config())
        b = CLIPTextRef(pc.openclip_bigg_config())
    assert sum(p.numel() for p in a.parameters()) == 123_060_480          # CLIP ViT-L/14 text model (SDXL text_encoder)
    assert sum(p.numel() for p in b.parameters()) == 694_659_840          # OpenCLIP bigG text model + projection (text_encoder_2)
