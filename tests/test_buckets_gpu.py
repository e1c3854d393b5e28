"""-m gpu: the reference's aspect-ratio buckets (utils/custom_dataset_sdxl.py:30: nine (height, width) pairs, one bucket per
batch, train_sdxl_zh.py:386-389).  Non-square, ragged latent shapes (56x104 latents -> 14x26 = 364 tokens at the deepest
level: no multiple of 64 anywhere) through the whole KD step against the oracle, the bucket-switching trainer, and every
bucket at full SDXL size by properties."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from test_model_gpu import gpu, rel_l2, round_weights_bf16_  # noqa: E402,F401


def _tiny_models(B, L, hw, seed=0):
    from oracle.step_ref import AdapterRef
    from oracle.unet_ref import UNet2DConditionRef, tiny_config
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.unet import HipUNet
    cfg = tiny_config()
    torch.manual_seed(seed)
    us = UNet2DConditionRef(cfg)
    round_weights_bf16_(us)
    for p in us.parameters():
        p.requires_grad_(False)
    import copy
    ut = copy.deepcopy(us)
    ad_ref = AdapterRef(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    ad = PEAAdapter(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    ad.load_state_dict(ad_ref.state_dict())
    ad = ad.cuda()
    round_weights_bf16_(ad_ref)
    hs = HipUNet(pc.tiny_config(), B, hw[0], hw[1], L, needs_grad=True)
    hs.load_state_dict(us.state_dict())
    ht = HipUNet(pc.tiny_config(), B, hw[0], hw[1], 77, share_weights_from=hs)
    return cfg, us, ut, ad_ref, ad, hs, ht


def _check_step(tr, ad, ad_ref, us, ut, cfg, B, L, hw, seed, tag):
    from oracle.step_ref import synthetic_batch, training_step_ref
    from oracle.unet_ref import cast_hook_ref
    batch = synthetic_batch(cfg, B, L=L, enc_dim=128, seed=seed, latent_hw=hw)
    assert batch["time_ids"][0].tolist() == [hw[0] * 8, hw[1] * 8, 0, 0, hw[0] * 8, hw[1] * 8]
    out = tr.training_step(batch, 0, sync=True)
    bq = dict(batch)
    for k in ("enc", "enc_uncond", "teacher_ehs", "teacher_neg", "teacher_pooled"):
        bq[k] = batch[k].to(torch.bfloat16).float()
    for p in ad_ref.parameters():
        p.grad = None
    ref = training_step_ref(ad_ref, us, ut, bq, cast_hook_ref)
    ref["loss"].backward()
    e = rel_l2(tr.export("eps_student"), ref["noise_pred"])
    total = abs(float(ref["loss"]))
    print(f"[{tag} {hw[0]}x{hw[1]}] eps_student rel_l2={e:.3e} loss hip={float(out['loss']):.6f} oracle={float(ref['loss']):.6f}")
    assert e < 2e-2
    for k in tr.LOG_KEYS:
        assert abs(float(out[k]) - float(ref[k])) <= 2e-2 * abs(float(ref[k])) + 5e-3 * total, k
    g_ref = torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()])
    eg = rel_l2(ad.flat_grad, g_ref)
    print(f"   adapter grad rel_l2={eg:.3e}")
    assert eg < 4e-2
    return out


@pytest.mark.parametrize("hw", [(56, 104), (72, 88), (112, 56)])
def test_bucket_shapes_tiny_vs_oracle(gpu, hw):
    """latents of the buckets 448x832, 576x704 and 896x448 through the whole step (merged passes, 52-token student context)"""
    from pea_diffusion_amd.train import PEATrainer
    B, L = 2, 52
    cfg, us, ut, ad_ref, ad, hs, ht = _tiny_models(B, L, hw)
    tr = PEATrainer(ad, hs, ht)
    _check_step(tr, ad, ad_ref, us, ut, cfg, B, L, hw, 3, "bucket tiny")
    from test_model_gpu import lib_merge_state
    assert lib_merge_state(tr) == 1


def test_bucketed_trainer_switches_and_releases(gpu):
    """one trainer surface over several latent shapes: contexts are created per shape, share the weights, each matches the
    oracle; with a memory cap of zero every switch releases the previous context and results stay bit-identical"""
    from oracle.step_ref import synthetic_batch
    from pea_diffusion_amd.train import BucketedTrainer
    B, L = 2, 52
    shapes = [(16, 16), (24, 8), (8, 24)]
    cfg, us, ut, ad_ref, ad, hs, ht = _tiny_models(B, L, shapes[0])
    tr = BucketedTrainer(ad, hs, ht)
    first = {}
    for i, hw in enumerate(shapes + shapes[::-1]):
        out = _check_step(tr, ad, ad_ref, us, ut, cfg, B, L, hw, 5 + (i % 3 if i < 3 else 2 - (i % 3)), "bucketed")
        key = (hw, 5 + (i % 3 if i < 3 else 2 - (i % 3)))
        sig = (float(out["loss"]), ad.flat_grad.clone())
        if key in first:
            assert first[key][0] == sig[0] and torch.equal(first[key][1], sig[1]), hw     # same batch, same context: bit-repro
        first[key] = sig
    assert sorted(tr.shapes) == sorted(shapes) and tr.resident_bytes() > 0
    # memory cap 0: every switch frees the other contexts first
    tr.max_resident_bytes = 0
    for hw in shapes:
        batch = synthetic_batch(cfg, B, L=L, enc_dim=128, seed=9, latent_hw=hw)
        o1 = tr.training_step(batch, 0, sync=True)
        g1 = ad.flat_grad.clone()
        others = [tr._resident_bytes(c) for k, c in tr._ctx.items() if k != hw]
        assert sum(others) == 0 and tr._resident_bytes(tr._ctx[hw]) > 0
        o2 = tr.training_step(batch, 0, sync=True)
        assert float(o1["loss"]) == float(o2["loss"]) and torch.equal(g1, ad.flat_grad)
    # the optimizer state is one: steps on different buckets advance the same schedule
    tr.optimizer_step()
    tr.training_step(synthetic_batch(cfg, B, L=L, enc_dim=128, seed=9, latent_hw=shapes[1]), 0)
    tr.optimizer_step()
    assert tr.global_step == 2
    hs.release_activations()                      # an explicitly released context reallocates on its next forward
    tr.max_resident_bytes = 1 << 40
    _check_step(tr, ad, ad_ref, us, ut, cfg, B, L, shapes[0], 11, "after release")


def test_all_nine_buckets_full_sdxl_properties(gpu):
    """every bucket of utils/custom_dataset_sdxl.py:30 at full SDXL size, B = 2, through BucketedTrainer: idempotence
    (teacher fed the student's own conditioning -> all KD terms exactly 0), the masked noise term, finite non-zero
    gradients, bit-reproducibility; the nine contexts stay resident together"""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.frontend import BUCKETS, add_time_ids_from_batch
    from pea_diffusion_amd.train import BucketedTrainer
    from pea_diffusion_amd.unet import HipUNet
    cfg = pc.sdxl_config()
    B, L = 2, 77
    h0, w0 = BUCKETS[4][0] // 8, BUCKETS[4][1] // 8
    student = HipUNet(cfg, B, h0, w0, L, needs_grad=True)
    student.init_random(3)
    teacher = HipUNet(cfg, B, h0, w0, L, share_weights_from=student)
    torch.manual_seed(0)
    ad = PEAAdapter(1024, 1280, 1024, 2048, False).cuda()
    tr = BucketedTrainer(ad, student, teacher)
    g = torch.Generator(device="cuda").manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g, device="cuda")
    enc = r(B, L, 1024)
    with torch.no_grad():
        pooled, tokens = ad(enc)
    for bid, (H, W) in enumerate(BUCKETS):
        h, w = H // 8, W // 8
        tid = add_time_ids_from_batch({"original_size": torch.tensor([[H, W]] * B), "crops_coords_top_left": torch.zeros(B, 2),
                                       "bucket_id": torch.tensor([bid] * B)}, "cuda")
        assert tid[0].tolist() == [H, W, 0, 0, H, W]
        base = dict(latents=r(B, 4, h, w), noise=r(B, 4, h, w), timesteps=torch.tensor([500] * B, device="cuda"),
                    enc=enc, enc_uncond=r(B, L, 1024), prompt_mask=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                    teacher_ehs=tokens.float(), teacher_neg=r(B, L, 2048), teacher_pooled=pooled.float(), time_ids=tid)
        out = tr.training_step(dict(base, zh_or_not=torch.zeros(B, dtype=torch.int64, device="cuda")), 0, sync=True)
        assert float(out["train_loss_logits"]) == 0.0 and float(out["train_loss_features"]) == 0.0 and float(out["loss"]) == 0.0, (H, W)
        assert torch.equal(tr.export("eps_student"), tr.export("eps_teacher"))
        b1 = dict(base, zh_or_not=torch.ones(B, dtype=torch.int64, device="cuda"))
        out1 = tr.training_step(b1, 0, sync=True)
        want = ((tr.export("eps_student") - base["noise"]) ** 2).mean().item()
        assert abs(float(out1["train_loss"]) - want) <= 1e-4 * want, (H, W)
        g1 = ad.flat_grad.clone()
        assert torch.isfinite(g1).all() and (g1 != 0).float().mean() > 0.9, (H, W)
        out2 = tr.training_step(b1, 0, sync=True)
        assert torch.equal(g1, ad.flat_grad) and float(out2["loss"]) == float(out1["loss"]), (H, W)
        print(f"[bucket {H}x{W}] loss={float(out1['loss']):.5f} resident={tr.resident_bytes() / 2**30:.1f} GiB")
    assert len(tr.shapes) == 9 and all(tr._resident_bytes(c) > 0 for c in tr._ctx.values())
