"""-m gpu: text encoders on the HIP tape (C ABI pea_text_*) against the golden vectors of the installed HF transformers
release and against oracle/text_ref.py at the real CLIP-L width."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from test_model_gpu import gpu, rel_l2, round_weights_bf16_  # noqa: E402,F401


def _golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.") and "position_ids" not in k}
    return z, sd


def test_clip_text_encoder_vs_transformers_golden(gpu, golden_dir):
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.text import HipTextEncoder
    z, sd = _golden(golden_dir, "text_clip.npz")
    enc = HipTextEncoder(pc.tiny_clip_config(), 2, 77)
    assert set(enc.weight_table()) == set(sd)
    enc.load_state_dict(sd)
    ids = torch.from_numpy(z["ids"]).cuda()
    for idx, key in [(-2, "hidden_2"), (-1, "last_hidden_state"), (0, "hidden_0"), (3, "hidden_3")]:
        hid, pooled = enc.encode(ids, hidden_index=idx)
        e = rel_l2(hid, torch.from_numpy(z[key]))
        print(f"[clip text tiny] hidden_index {idx}: rel_l2={e:.3e}")
        assert e < 2e-2, key
    e = rel_l2(pooled, torch.from_numpy(z["text_embeds"]))
    print(f"[clip text tiny] text_embeds rel_l2={e:.3e}")
    assert e < 2e-2
    out = enc(ids, output_hidden_states=True)            # the encode_prompt access pattern
    assert rel_l2(out[0], torch.from_numpy(z["text_embeds"])) < 2e-2
    assert rel_l2(out.hidden_states[-2], torch.from_numpy(z["hidden_2"])) < 2e-2


def test_bert_text_encoder_vs_transformers_golden(gpu, golden_dir):
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.text import HipTextEncoder
    z, sd = _golden(golden_dir, "text_bert.npz")
    enc = HipTextEncoder(pc.tiny_bert_config(), 2, 52)
    enc.load_state_dict({"bert." + k: v for k, v in sd.items()})       # cn_clip checkpoints carry the `bert.` prefix
    ids = torch.from_numpy(z["ids"])
    tokens, _ = enc.encode_text(ids.cuda())
    valid = ids != 0
    want = torch.from_numpy(z["last_hidden_state"])
    e = rel_l2(tokens.cpu()[valid], want[valid])
    print(f"[bert text tiny] per-token states rel_l2={e:.3e} (valid positions)")
    assert e < 2e-2 and torch.isfinite(tokens).all()


def test_xlm_roberta_text_encoder_vs_transformers_golden(gpu, golden_dir):
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.text import HipTextEncoder
    z, sd = _golden(golden_dir, "text_xlmr.npz")
    enc = HipTextEncoder(pc.tiny_xlmr_config(), 2, 64)
    assert set(enc.weight_table()) == set(sd)
    enc.load_state_dict(sd)
    ids = torch.from_numpy(z["ids"])
    tokens, _ = enc.encode_text(ids.cuda())
    valid = ids != 1
    e = rel_l2(tokens.cpu()[valid], torch.from_numpy(z["last_hidden_state"])[valid])
    print(f"[xlm-r text tiny] per-token states rel_l2={e:.3e} (valid positions)")
    assert e < 2e-2


def test_clip_l_width_vs_oracle(gpu):
    """CLIP-L width and depth (12 x 768, 123 M parameters), 77 tokens, against the fp32 restatement"""
    from oracle.text_ref import CLIPTextRef
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.text import HipTextEncoder
    cfg = pc.clip_l_config()
    torch.manual_seed(0)
    ref = CLIPTextRef(cfg)
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() >= 2:
                p.mul_(0.5)
    round_weights_bf16_(ref)
    enc = HipTextEncoder(cfg, 4, 77)
    enc.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 49000, (4, 77), generator=g)
    ids[:, 0] = 49406
    for b, n in enumerate([5, 20, 50, 76]):
        ids[b, n] = 49407
        ids[b, n + 1:] = 49407
    with torch.no_grad():
        o = ref(ids)
    hid, pooled = enc.encode(ids.cuda(), hidden_index=-2)
    e1, e2 = rel_l2(hid, o["hidden_states"][-2]), rel_l2(pooled, o["pooled"])
    print(f"[clip-l] hidden_states[-2] rel_l2={e1:.3e} pooled rel_l2={e2:.3e}")
    assert e1 < 2e-2 and e2 < 2e-2
