"""-m gpu: the BASELINE.json configurations at the sizes they are quoted on, one test each.

  C2  SDXL 1024x1024, per-GPU batch 4, teacher == student checkpoint -> the merged 2B-row launch set bench.py times
  C3  SDXL 1024x1024, per-rank batch 8 (global 64 = 8 x 8)           -> merged passes (default) and the two-stream path at B = 8
  C4  SSD-1B student (real per-position layout, no mid block) under the SDXL teacher: full width vs the fp32 CPU oracle
      at 512x512, and at 1024x1024 through size-independent properties
The oracle cannot finish 1024x1024 in seconds, so full-size cases are checked through properties the step offers:
idempotence (a teacher fed the student's own conditioning reproduces the student bit for bit), the zh_or_not masks,
bit-reproducibility, finite non-zero adapter gradients."""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
from test_model_gpu import (_fast_fill_, check_layer_grads, gpu, layer_grad_hooks, lib_merge_state, rel_l2,  # noqa: E402,F401
                            round_weights_bf16_)


def _properties(tr, ad, B, hw, L, same_weights):
    g = torch.Generator(device="cuda").manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g, device="cuda")
    enc = r(B, L, 1024)
    with torch.no_grad():
        pooled, tokens = ad(enc)                                     # the student's own conditioning
    base = dict(latents=r(B, 4, hw, hw), noise=r(B, 4, hw, hw),
                timesteps=torch.tensor([10, 250, 500, 999, 3, 700, 42, 901][:B], device="cuda"),
                enc=enc, enc_uncond=r(B, L, 1024), prompt_mask=torch.zeros(B, dtype=torch.uint8, device="cuda"),
                teacher_ehs=tokens.float(), teacher_neg=r(B, 77, 2048), teacher_pooled=pooled.float(),
                time_ids=torch.tensor([[hw * 8., hw * 8, 0, 0, hw * 8, hw * 8]] * B, device="cuda"))
    if same_weights:
        b0 = dict(base, zh_or_not=torch.zeros(B, dtype=torch.int64, device="cuda"))
        out = tr.training_step(b0, 0, sync=True)
        assert float(out["train_loss_logits"]) == 0.0 and float(out["train_loss_features"]) == 0.0
        assert float(out["loss"]) == 0.0
        assert torch.equal(tr.export("eps_student"), tr.export("eps_teacher"))
    b1 = dict(base, zh_or_not=torch.ones(B, dtype=torch.int64, device="cuda"))
    out1 = tr.training_step(b1, 0, sync=True)
    eps_s = tr.export("eps_student")
    want = ((eps_s - base["noise"]) ** 2).mean().item()
    assert abs(float(out1["train_loss"]) - want) <= 1e-4 * want and float(out1["loss"]) == float(out1["train_loss"])
    assert float(out1["train_loss_logits"]) == 0.0 and float(out1["train_loss_features"]) == 0.0
    g1 = ad.flat_grad.clone()
    assert torch.isfinite(g1).all() and (g1 != 0).float().mean() > 0.9
    out2 = tr.training_step(b1, 0, sync=True)
    assert torch.equal(g1, ad.flat_grad) and float(out2["loss"]) == float(out1["loss"]), "step is not bit-reproducible"
    # mixed masks + one CFG-dropped sample: every term is live, per-sample independence: a step on the same samples in
    # the same slots gives the same eps for them whatever the other samples' flags are
    zh = torch.tensor([1, 0, 0, 1, 0, 1, 1, 0][:B], device="cuda")
    pm = torch.zeros(B, dtype=torch.uint8, device="cuda")
    pm[B - 1] = 1
    out3 = tr.training_step(dict(base, zh_or_not=zh, prompt_mask=pm), 0, sync=True)
    eps3 = tr.export("eps_student")
    assert torch.equal(eps3[: B - 1], eps_s[: B - 1]) and not torch.equal(eps3[B - 1], eps_s[B - 1])
    for k in tr.LOG_KEYS:
        assert torch.isfinite(out3[k]).item()
    if not same_weights:
        assert float(out3["train_loss_logits"]) > 0 and float(out3["train_loss_features"]) > 0
    assert abs(float(out3["loss"]) - (float(out3["train_loss"]) + float(out3["train_loss_logits"])
                                      + 0.1 * float(out3["train_loss_features"]))) <= 1e-5 * abs(float(out3["loss"]))


@pytest.mark.parametrize("B,two_stream", [(4, False), (8, False), (8, True)])
def test_sdxl_1024_bench_workloads(gpu, B, two_stream):
    """C2 (B = 4: exactly the launch set bench.py times -- merged passes, 2B = 8 rows per forward launch) and C3's
    per-rank workload (B = 8) on both of its paths: merged passes (the default) and the teacher forward on the side stream."""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    cfg, hw, L = pc.sdxl_config(), 128, 77
    student = HipUNet(cfg, B, hw, hw, L, needs_grad=True)
    student.init_random(3)
    teacher = HipUNet(cfg, B, hw, hw, L, share_weights_from=student)
    torch.manual_seed(0)
    ad = PEAAdapter(1024, 1280, 1024, 2048, False).cuda()
    tr = PEATrainer(ad, student, teacher)
    if two_stream:
        tr.set_option("merge_passes", 0)
    _properties(tr, ad, B, hw, L, same_weights=True)
    assert (lib_merge_state(tr) == 1) == (not two_stream)
    del tr, student, teacher
    torch.cuda.empty_cache()


def _ssd1b_weight_table_checks(hip, ref):
    table = hip.weight_table()
    sd = ref.state_dict()
    assert set(table) == set(sd)
    assert not any(k.startswith("mid_block.") for k in table)
    for k, shape in table.items():
        assert tuple(sd[k].shape) == tuple(shape) or sd[k].numel() == torch.Size(shape).numel(), k


def test_ssd1b_tiny_layout_vs_oracle(gpu):
    """nested transformer depths + reverse list + no mid block on a small model: forward (every tap) and the KD step
    under a teacher that HAS a mid block (its 'm' tap has no partner and leaves the feature loss)."""
    from oracle import unet_ref as ou
    from oracle.step_ref import AdapterRef, synthetic_batch, training_step_ref
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    B, L = 2, 12
    kw = dict(transformer_layers_per_block=(1, (1, 2), (2, 1)), reverse_transformer_layers_per_block=((1, 2, 3), (2, 0, 1), 1),
              mid_block_type=None)
    cfg_s, cfg_t = ou.tiny_config(), ou.tiny_config()
    pcs, pct = pc.tiny_config(), pc.tiny_config()
    for k, v in kw.items():
        setattr(cfg_s, k, v)
        setattr(pcs, k, v)
    torch.manual_seed(1)
    us, ut = ou.UNet2DConditionRef(cfg_s), ou.UNet2DConditionRef(cfg_t)
    for m in (us, ut):
        round_weights_bf16_(m)
        for p in m.parameters():
            p.requires_grad_(False)
    hs = HipUNet(pcs, B, 16, 16, L, needs_grad=True)
    ht = HipUNet(pct, B, 16, 16, 77)
    _ssd1b_weight_table_checks(hs, us)
    hs.load_state_dict(us.state_dict())
    ht.load_state_dict(ut.state_dict())
    assert hs.mid_block is None and hs.tap_names == ["d0", "d1", "d2", "u0", "u1", "u2"] and "m" in ht.tap_names
    ad_ref = AdapterRef(128, cfg_s.pooled_dim, 192, cfg_s.cross_attention_dim, False)
    ad = PEAAdapter(128, cfg_s.pooled_dim, 192, cfg_s.cross_attention_dim, False)
    ad.load_state_dict(ad_ref.state_dict())
    ad = ad.cuda()
    round_weights_bf16_(ad_ref)
    batch = synthetic_batch(cfg_s, B, L=L, enc_dim=128, seed=2)
    tr = PEATrainer(ad, hs, ht)
    out = tr.training_step(batch, 0, sync=True)
    bq = dict(batch)
    for k in ("enc", "enc_uncond", "teacher_ehs", "teacher_neg", "teacher_pooled"):
        bq[k] = batch[k].to(torch.bfloat16).float()
    ref = training_step_ref(ad_ref, us, ut, bq, ou.cast_hook_ref)
    ref["loss"].backward()
    assert list(ref["taps_s"]) == hs.tap_names
    for i, k in enumerate(hs.tap_names):
        e = rel_l2(hs.tap(i), ref["taps_s"][k])
        assert e < 2e-2, (k, e)
    for k in tr.LOG_KEYS:
        assert abs(float(out[k]) - float(ref[k])) <= 1e-2 * max(abs(float(ref[k])), 1e-3), k
    g_ref = torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()])
    e = rel_l2(ad.flat_grad, g_ref)
    print(f"[ssd1b-layout tiny] loss hip={float(out['loss']):.6f} oracle={float(ref['loss']):.6f} grad rel_l2={e:.3e}")
    assert e < 3e-2


def test_nested_last_entry_with_mid_block_vs_oracle(gpu):
    """A nested LAST transformer_layers_per_block entry with unequal values AND a mid block: diffusers hands that entry to
    UNetMidBlock2DCrossAttn, which reads element [0].  The oracle and the product build their graphs independently from the
    same config; the weight tables must agree key for key and the forward (every tap, mid included) must match."""
    from oracle import unet_ref as ou
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.unet import HipUNet
    B, L = 2, 12
    kw = dict(transformer_layers_per_block=(1, 1, (2, 1)), reverse_transformer_layers_per_block=((1, 2, 1), 1, 1))
    cfg_o, cfg_p = ou.tiny_config(), pc.tiny_config()
    for k, v in kw.items():
        setattr(cfg_o, k, v)
        setattr(cfg_p, k, v)
    assert pc.depth_tables(cfg_p)[2] == 2
    torch.manual_seed(4)
    uo = ou.UNet2DConditionRef(cfg_o)
    assert len(uo.mid_block.attentions[0].transformer_blocks) == 2
    round_weights_bf16_(uo)
    hu = HipUNet(cfg_p, B, 16, 16, L, needs_grad=False)
    table, sd = hu.weight_table(), uo.state_dict()
    assert set(table) == set(sd), sorted(set(table) ^ set(sd))[:8]
    hu.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, 4, 16, 16, generator=g)
    t = torch.tensor([17, 801])
    ehs = torch.randn(B, L, cfg_o.cross_attention_dim, generator=g).to(torch.bfloat16).float()
    pooled = torch.randn(B, cfg_o.pooled_dim, generator=g).to(torch.bfloat16).float()
    tid = torch.tensor([[128., 128, 0, 0, 128, 128]] * B)
    taps = {}
    ou.cast_hook_ref(uo, taps)
    with torch.no_grad():
        eps_o = uo(x, t, ehs, added_cond_kwargs={"text_embeds": pooled, "time_ids": tid})[0]
    eps_h = hu(x.cuda(), t.cuda(), ehs.cuda(), added_cond_kwargs={"text_embeds": pooled.cuda(), "time_ids": tid.cuda()},
               return_dict=False)[0]
    e = rel_l2(eps_h, eps_o)
    assert e < 1.5e-2, e
    assert "m" in hu.tap_names and list(taps) == hu.tap_names
    for i, k in enumerate(hu.tap_names):
        assert rel_l2(hu.tap(i), taps[k]) < 2e-2, k
    print(f"[nested-last + mid] eps rel_l2={e:.3e}")


def _build_fast(cfg, seed):
    from oracle import unet_ref as ou
    orig = torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_
    torch.nn.init.kaiming_uniform_ = lambda t, *a, **k: t
    torch.nn.init.uniform_ = lambda t, *a, **k: t
    try:
        m = ou.UNet2DConditionRef(cfg)
    finally:
        torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_ = orig
    _fast_fill_(m, seed=seed)
    round_weights_bf16_(m)
    for p in m.parameters():
        p.requires_grad_(False)
    return m


def _ssd1b_under_sdxl_teacher_vs_oracle(B, hw, tag):
    from oracle import unet_ref as ou
    from oracle.step_ref import AdapterRef, synthetic_batch, training_step_ref
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    torch.set_num_threads(min(64, len(os.sched_getaffinity(0))))
    L = 77
    us, ut = _build_fast(ou.ssd1b_config(), 5), _build_fast(ou.sdxl_config(), 7)
    torch.manual_seed(6)
    ad_ref = AdapterRef(1024, 1280, 1024, 2048, False)
    ad = PEAAdapter(1024, 1280, 1024, 2048, False)
    ad.load_state_dict(ad_ref.state_dict())
    ad = ad.cuda()
    round_weights_bf16_(ad_ref)
    hs = HipUNet(pc.ssd1b_config(), B, hw, hw, L, needs_grad=True)
    _ssd1b_weight_table_checks(hs, us)
    hs.load_state_dict(us.state_dict())
    ht = HipUNet(pc.sdxl_config(), B, hw, hw, 77)
    ht.load_state_dict(ut.state_dict())
    batch = synthetic_batch(ou.sdxl_config(), B, L=L, enc_dim=1024, seed=2, latent_hw=hw)
    tr = PEATrainer(ad, hs, ht)
    out = tr.training_step(batch, 0, sync=True)
    assert lib_merge_state(tr) == -1
    bq = dict(batch)
    for k in ("enc", "enc_uncond", "teacher_ehs", "teacher_neg", "teacher_pooled"):
        bq[k] = batch[k].to(torch.bfloat16).float()
    store, handles = layer_grad_hooks(us)
    ref = training_step_ref(ad_ref, us, ut, bq, ou.cast_hook_ref)
    ref["loss"].backward()
    for h in handles:
        h.remove()
    e_s, e_t = rel_l2(tr.export("eps_student"), ref["noise_pred"]), rel_l2(tr.export("eps_teacher"), ref["noise_pred_teacher"])
    print(f"[ssd1b student / sdxl teacher {tag}] eps_student rel_l2={e_s:.3e} eps_teacher rel_l2={e_t:.3e}")
    # every cross-attention K / V projection and time_emb_proj layer of the pruned student on its own (per-position depths
    # [2,2],[4,4] down / [4,4,10],[2,1,1] up, no mid block)
    check_layer_grads(tr, store, B, 2e-2, f"ssd1b {tag} vs fp32 oracle")       # measured worst 8.6e-3 / 5.8e-3 at 512x512
    assert e_s < 2e-2 and e_t < 2e-2
    total = abs(float(ref["loss"]))
    for k in tr.LOG_KEYS:
        h, r = float(out[k]), float(ref[k])
        print(f"   {k}: hip={h:.6f} oracle={r:.6f}")
        assert abs(h - r) <= 1e-2 * abs(r) + 2e-3 * total, k
    g_ref = torch.cat([p.grad.reshape(-1) for p in ad_ref.parameters()])
    eg = rel_l2(ad.flat_grad, g_ref)
    print(f"   adapter grad rel_l2={eg:.3e} |ref|={g_ref.norm():.3e}")
    assert eg < 2e-2


def test_ssd1b_full_width_under_sdxl_teacher_vs_oracle_512(gpu):
    """C4 at full width: the 1.30 B-parameter SSD-1B-layout student under the 2.57 B-parameter SDXL teacher (own weights
    each), 512x512, batch 2: eps of both UNets, the four logged scalars and the flat adapter gradient against the fp32
    CPU oracle (two-stream path: the teacher is a different model)."""
    _ssd1b_under_sdxl_teacher_vs_oracle(2, 64, "512x512")            # measured: eps 6.5e-3 / 5.5e-3, gradient 4.0e-3


def test_ssd1b_full_width_under_sdxl_teacher_vs_oracle_1024(gpu):
    """BASELINE configs[3] at its own resolution (1024x1024, latent 128x128, batch 1) against the fp32 CPU oracle."""
    _ssd1b_under_sdxl_teacher_vs_oracle(1, 128, "1024x1024")


def test_ssd1b_1024_properties(gpu):
    """C4 at the metric's size (1024x1024, batch 4): the oracle cannot finish it in seconds -> properties."""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.train import PEATrainer
    from pea_diffusion_amd.unet import HipUNet
    B, hw, L = 4, 128, 77
    student = HipUNet(pc.ssd1b_config(), B, hw, hw, L, needs_grad=True)
    student.init_random(3)
    teacher = HipUNet(pc.sdxl_config(), B, hw, hw, 77)
    teacher.init_random(4)
    torch.manual_seed(0)
    ad = PEAAdapter(1024, 1280, 1024, 2048, False).cuda()
    tr = PEATrainer(ad, student, teacher)
    _properties(tr, ad, B, hw, L, same_weights=False)
    assert lib_merge_state(tr) == -1


def test_layernorm_fold_opt_in_matches_oracle(gpu):
    """PEA_LN_FOLD=1 (read once per process, so this runs the tiny forward / training-step parity tests in a child
    process): LayerNorm folded into the consuming Linear -- statistics pass + W.gamma / s / t -- must meet the same
    oracle tolerances as the default path."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_model_gpu.py"), "-m", "gpu", "-x", "-q",
                        "-k", "unet_forward_tiny or unet_backward_tiny or training_step_vs_oracle or merged_passes"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PEA_LN_FOLD="1"), cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_upconv_folded_form_opt_out_matches_oracle(gpu):
    """PEA_UPCONV_SUBPIXEL=0 (read once per process -> child process): the upsampler convs as a 3 x 3 gather over the
    nearest-2x virtual image with NHWC taps -- the form the sub-pixel default replaced -- must still meet the oracle, and the
    tap import / export must behave the same through both layouts."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_model_gpu.py"), "-m", "gpu", "-x", "-q",
                        "-k", "unet_forward_tiny or unet_backward_tiny or training_step_vs_oracle or merged_passes"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PEA_UPCONV_SUBPIXEL="0"), cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout

