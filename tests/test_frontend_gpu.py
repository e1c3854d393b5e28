"""-m gpu: the reference batch-dict entry (utils/custom_dataset_sdxl.py:384-409 -> train_sdxl_zh.py:305-449) on tiny
models: VAE encode + teacher CLIP towers + student BERT tower + add_time_ids from BUCKETS + the KD step through
PEATrainer.training_step_from_batch, against the same chain on the CPU oracles."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from test_model_gpu import gpu, rel_l2, round_weights_bf16_  # noqa: E402,F401


def test_training_step_from_reference_batch_dict(gpu):
    import oracle.text_ref as ot
    import oracle.vae_ref as ov
    from oracle.step_ref import AdapterRef, training_step_ref
    from oracle.unet_ref import UNet2DConditionRef, cast_hook_ref, tiny_config
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.frontend import BUCKETS, PEAFrontEnd
    from pea_diffusion_amd.text import HipTextEncoder
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    from pea_diffusion_amd.vae import HipVAEEncoder
    B, L, Lt = 2, 52, 77
    cfg = tiny_config()                                  # cross_attention_dim 128, pooled 128, latent 16x16
    vcfg_o, vcfg = ov.tiny_vae_config(), pc.tiny_vae_config()     # 3 levels: pixels = 4 x latent -> 64x64 pixels
    c1 = pc.TextConfig(vocab_size=1000, hidden_size=64, num_attention_heads=1, num_hidden_layers=2, intermediate_size=256,
                       eos_token_id=999, name="t1")
    c2 = pc.TextConfig(vocab_size=1000, hidden_size=64, num_attention_heads=1, num_hidden_layers=2, intermediate_size=256,
                       hidden_act="gelu", projection_dim=128, eos_token_id=999, name="t2")
    cz = pc.tiny_bert_config()
    torch.manual_seed(0)
    vae_r, t1_r, t2_r, zh_r = ov.VAEEncoderRef(vcfg_o), ot.CLIPTextRef(c1), ot.CLIPTextRef(c2), ot.BertTextRef(cz)
    us = UNet2DConditionRef(cfg)
    ad_ref = AdapterRef(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    for m in (vae_r, t1_r, t2_r, zh_r, us, ad_ref):
        round_weights_bf16_(m)
    for p in us.parameters():
        p.requires_grad_(False)
    vae = HipVAEEncoder(vcfg, B, 64, 64)
    vae.load_state_dict(vae_r.state_dict())
    te1 = HipTextEncoder(c1, 2 * B, Lt)
    te1.load_state_dict(t1_r.state_dict())
    te2 = HipTextEncoder(c2, 2 * B, Lt)
    te2.load_state_dict(t2_r.state_dict())
    zh = HipTextEncoder(cz, 2 * B, L)
    zh.load_state_dict(zh_r.state_dict())
    hs = HipUNet(pc.tiny_config(), B, 16, 16, L, needs_grad=True)
    hs.load_state_dict(us.state_dict())
    ht = HipUNet(pc.tiny_config(), B, 16, 16, Lt, share_weights_from=hs)      # teacher == student checkpoint
    ad = PEAAdapter(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    ad.load_state_dict(ad_ref.state_dict())
    ad = ad.cuda()
    tr = PEATrainer(ad, hs, ht)
    tr.attach_frontend(PEAFrontEnd(vae, te1, te2, zh))
    g = torch.Generator().manual_seed(3)
    ids_en = torch.randint(1, 998, (B, Lt), generator=g)
    ids_en[:, 0] = 998
    ids_en[0, 9:] = 999
    ids_en[1, 30:] = 999
    neg = torch.full((1, Lt), 999)
    neg[0, 0] = 998
    ids_zh = torch.randint(1, 1000, (B, L), generator=g)
    ids_zh[0, 20:] = 0
    ids_zh[1, 41:] = 0
    ids_zh_u = torch.zeros(1, L, dtype=torch.int64)
    ids_zh_u[0, :2] = torch.tensor([101, 102])
    batch = {"pixel_values": torch.randn(B, 3, 64, 64, generator=g).clamp(-1, 1), "instance_prompt_ids": ["一只猫", "a dog"],
             "original_size": torch.tensor([[700, 900], [512, 512]]), "crops_coords_top_left": torch.tensor([[0, 12], [4, 0]]),
             "bucket_id": torch.tensor(6), "input_ids": ids_zh, "input_ids_uncond": ids_zh_u.repeat(B, 1),
             "zh_or_not": torch.tensor([1, 0]), "texts_en": ["a cat", "a dog"],
             "texts_en_ids": (ids_en, ids_en.clone()), "neg_en_ids": (neg, neg.clone()),
             # the step's random draws, fixed so the oracle chain sees the same values
             "_vae_noise": torch.randn(B, 4, 16, 16, generator=g), "_noise": torch.randn(B, 4, 16, 16, generator=g),
             "_timesteps": torch.tensor([250, 999]), "_prompt_mask": torch.tensor([False, True])}
    out = tr.training_step_from_batch(batch, 0, sync=True)
    assert out["loss"].ndim == 0
    # ---- the same chain on the oracles
    with torch.no_grad():
        mom = vae_r.moments(batch["pixel_values"])
        lat = (mom[:, :4] + torch.exp(0.5 * mom[:, 4:].clamp(-30, 20)) * batch["_vae_noise"]) * vcfg_o.scaling_factor
        o1, o2 = t1_r(torch.cat([ids_en, neg.repeat(B, 1)])), t2_r(torch.cat([ids_en, neg.repeat(B, 1)]))
        pe = torch.cat([o1["hidden_states"][-2], o2["hidden_states"][-2]], -1)
        enc = zh_r(torch.cat([ids_zh, ids_zh_u.repeat(B, 1)]))["last_hidden_state"]
    q = lambda t: t.to(torch.bfloat16).float()
    bq = {"latents": lat, "noise": batch["_noise"], "timesteps": batch["_timesteps"], "enc": q(enc[:B]), "enc_uncond": q(enc[B:]),
          "prompt_mask": batch["_prompt_mask"], "zh_or_not": batch["zh_or_not"], "teacher_ehs": q(pe[:B]), "teacher_neg": q(pe[B:]),
          "teacher_pooled": q(o2["pooled"][:B]),
          "time_ids": torch.tensor([[700, 900, 0, 12] + BUCKETS[6], [512, 512, 4, 0] + BUCKETS[6]])}
    import copy
    ref = training_step_ref(ad_ref, us, copy.deepcopy(us), bq, cast_hook_ref)
    ref["loss"].backward()
    for k in tr.LOG_KEYS:
        h, r = float(out[k]), float(ref[k])
        print(f"[batch-dict step] {k}: hip={h:.6f} oracle={r:.6f}")
        tol = 5e-2 if k in ("train_loss_features", "train_loss_logits") else 1.5e-2   # teacher == student: differences of near-equal tensors
        assert abs(h - r) <= tol * max(abs(r), 1e-3), k
    g_ref = torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()])
    e = rel_l2(ad.flat_grad, g_ref)
    print(f"   adapter grad rel_l2={e:.3e}")
    assert e < 4e-2
    # without the overrides the step draws its own noise / timesteps / CFG mask and still runs
    plain = {k: v for k, v in batch.items() if not k.startswith("_")}
    out2 = tr.training_step_from_batch(plain, 1, generator=torch.Generator(device="cuda").manual_seed(1), sync=True)
    assert torch.isfinite(out2["loss"]).item()
    # the text towers on their own streams (default) and on the VAE's stream give the same post-encoder batch, bit for bit
    fe = tr.frontend
    prepared = {}
    for conc in (True, False):
        fe.concurrent_towers = conc
        prepared[conc] = fe.prepare(batch)
        torch.cuda.synchronize()
    fe.concurrent_towers = True
    for k, v in prepared[True].items():
        assert torch.equal(v, prepared[False][k]), f"front end output '{k}' differs between concurrent and serial towers"
