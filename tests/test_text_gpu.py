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


def test_t5_encoder_vs_transformers_golden(gpu, golden_dir):
    """mT5-form encoder against HF T5EncoderModel.encoder(ids, attention_mask=ids.ne(pad), output_hidden_states=True):
    RMSNorm, bucketed relative position bias, unscaled scores, gated gelu_new FF, heads * d_kv != d_model"""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.text import HipTextEncoder
    z, sd = _golden(golden_dir, "text_t5.npz")
    enc = HipTextEncoder(pc.tiny_t5_config(), 2, 40)
    assert set(enc.weight_table()) == set(sd) - {"encoder.embed_tokens.weight"}
    enc.load_state_dict(sd)
    ids = torch.from_numpy(z["ids"])
    valid = ids != 0
    for idx, key in [(0, "hidden_0"), (1, "hidden_1"), (-1, "last_hidden_state")]:
        hid, pooled = enc.encode(ids.cuda(), hidden_index=idx)
        e = rel_l2(hid.cpu()[valid], torch.from_numpy(z[key])[valid])
        print(f"[t5 tiny] hidden_index {idx}: rel_l2={e:.3e} (valid positions)")
        assert e < 2e-2 and pooled is None and torch.isfinite(hid).all(), key
    # the reference's call (train_sdxl_zh.py:339-342)
    mask = ids.ne(0)
    out = enc.encoder(ids.cuda(), attention_mask=mask.cuda(), output_hidden_states=True)
    assert rel_l2(out[0].cpu()[valid], torch.from_numpy(z["last_hidden_state"])[valid]) < 2e-2
    tok, _ = enc.encode_text(ids.cuda())
    assert torch.equal(tok, out[0])


def test_t5_relative_bias_table_bit_exact(gpu):
    """the [heads][L][L] additive bias built on the device = table[bucket(k - q)][h] * log2(e), bucket indices bit-exact
    with the restatement (which is pinned against transformers' _relative_position_bucket) over 512 positions"""
    import ctypes
    from oracle.text_ref import t5_relative_bucket
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.text import HipTextEncoder
    cfg = pc.tiny_t5_config()
    cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance, cfg.num_hidden_layers = 32, 128, 1
    L = 512
    enc = HipTextEncoder(cfg, 1, L)
    enc.init_random(0)
    H = cfg.num_attention_heads
    # a table whose entries identify their bucket: rel[b][h] = b + h / 8
    tab = (torch.arange(32, dtype=torch.float32)[:, None] + torch.arange(H, dtype=torch.float32)[None] / 8).cuda()
    from pea_diffusion_amd._lib import check, lib, ptr, stream_ptr
    check(lib().pea_unet_load_weight(enc._h, b"encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", ptr(tab), tab.numel(), stream_ptr()))
    out = torch.empty(H, L, L, device="cuda")
    check(lib().pea_text_rel_bias(enc._h, ptr(out), stream_ptr()))
    torch.cuda.synchronize()
    pos = torch.arange(L)
    want_bucket = t5_relative_bucket(pos[None, :] - pos[:, None], 32, 128)
    got = out.cpu() / 1.4426950408889634
    for h in range(H):
        assert torch.equal(torch.round(got[h] - h / 8).long(), want_bucket), h


def test_mt5_xl_width_vs_oracle(gpu):
    """mt5-xl block geometry (d_model 2048, 32 heads x 64, d_ff 5120; 4 of the 24 blocks and a 32 k vocabulary to bound
    the CPU side), 77 tokens, B = 8 as the trainer feeds it (prompts + unconditional), against the fp32 restatement"""
    from oracle.text_ref import T5EncoderRef
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.text import HipTextEncoder
    cfg = pc.mt5_xl_config()
    cfg.num_hidden_layers, cfg.vocab_size = 4, 32000
    torch.manual_seed(0)
    ref = T5EncoderRef(cfg)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if ".q." in n or ".k." in n:
                p.mul_(0.35)                     # nn.Linear init has |w| ~ 1/sqrt(K); unscaled scores want q.k of order one
            elif "relative_attention_bias" in n or "shared" in n:
                pass
            elif p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    round_weights_bf16_(ref)
    enc = HipTextEncoder(cfg, 8, 77)
    enc.load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(2, 32000, (8, 77), generator=g)
    for b, n in enumerate([1, 5, 20, 33, 50, 64, 76, 77]):
        if n < 77:
            ids[b, n - 1] = 1
            ids[b, n:] = 0
    with torch.no_grad():
        o = ref(ids)
    valid = ids != 0
    hid, _ = enc.encode(ids.cuda(), hidden_index=-1)
    e1 = rel_l2(hid.cpu()[valid], o["last_hidden_state"][valid])
    h2, _ = enc.encode(ids.cuda(), hidden_index=2)
    e2 = rel_l2(h2.cpu()[valid], o["hidden_states"][2][valid])
    print(f"[mt5-xl width] last_hidden_state rel_l2={e1:.3e} hidden_states[2] rel_l2={e2:.3e}")
    assert e1 < 2e-2 and e2 < 2e-2 and torch.isfinite(hid).all()
