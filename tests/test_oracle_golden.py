"""Pins the CPU oracle (oracle/*.py) against golden vectors captured from the reference's
own MLP / cast_hook / training_step / rescale_noise_cfg (oracle/make_golden.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle.step_ref import AdapterRef, kd_losses, rescale_noise_cfg_ref, training_step_ref
from oracle.unet_ref import (UNet2DConditionRef, UNetConfig, cast_hook_ref, count_params_analytic, sd15_config,
                             sdxl_config, tap_names, tiny_config)

T = lambda a: torch.from_numpy(np.asarray(a))


def _wsum(sd):
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


def _adapter_from(args):
    args = [int(a) for a in args]
    if len(args) == 3:
        return AdapterRef(args[0], args[1], args[2], None)
    return AdapterRef(args[0], args[1], args[2], args[3], bool(args[4]))


@pytest.mark.parametrize("tag", ["sdxl_small", "sdxl_small_residual", "test_small", "sd_small"])
def test_adapter_small_fwd_bwd(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"mlp_{tag}.npz"))
    m = _adapter_from(g["args"])
    m.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("w.")})   # same state-dict keys
    x = T(g["x"]).clone().requires_grad_(True)
    out = m(x)
    outs = out if isinstance(out, tuple) else (out,)
    for i, o in enumerate(outs):
        torch.testing.assert_close(o, T(g[f"out{i}"]), rtol=1e-6, atol=1e-6)
    torch.autograd.backward(outs, [T(g[f"gout{i}"]) for i in range(len(outs))])
    torch.testing.assert_close(x.grad, T(g["dx"]), rtol=1e-5, atol=1e-6)
    for k, p in m.named_parameters():
        torch.testing.assert_close(p.grad, T(g["g." + k]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["sdxl_6M", "sdxl_11M", "sdxl_in2048", "sdxl_in768", "sd15_full"])
def test_adapter_full_size(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"mlp_{tag}.npz"))
    torch.manual_seed(int(g["seed"]))
    m = _adapter_from(list(g["args"]) if len(g["args"]) == 3 else list(g["args"]))
    assert sum(p.numel() for p in m.parameters()) == int(g["nparam"])
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    assert abs(_wsum(m.state_dict()) - float(g["wsum"])) < 1e-6 * float(g["wsum"]), "seeded weights drifted"
    out = m(T(g["x"]))
    outs = out if isinstance(out, tuple) else (out,)
    for i, o in enumerate(outs):
        torch.testing.assert_close(o.detach(), T(g[f"out{i}"]), rtol=1e-5, atol=1e-6)
    # parameter gradients for the fixture's seeded output gradients: norm + strided sample of every gradient
    torch.autograd.backward(outs, [T(g[f"gout{i}"]) for i in range(len(outs))])
    for k, p_ in m.named_parameters():
        flat = p_.grad.reshape(-1)
        assert abs(float(flat.double().norm()) - float(g["gnorm." + k])) <= 1e-4 * float(g["gnorm." + k]), k
        torch.testing.assert_close(flat[::int(g["gstride." + k])], T(g["gsample." + k]), rtol=2e-4, atol=1e-5 * float(g["gnorm." + k]))


def test_adapter_param_counts():
    n = lambda *a: sum(p.numel() for p in AdapterRef(*a).parameters())
    assert n(1024, 1280, 1024, 2048) == 6_033_408          # README.md:11 "6M"
    assert n(1024, 1280, 2048, 2048) == 11_538_432
    assert n(2048, 1280, 2048, 2048) == 13_637_632
    assert n(768, 1280, 2048, 2048) == 11_013_632
    assert n(1024, 768, 2048, None) == 7_866_368


def _cfg_from(g, which):
    if "hip_dims" in g.files and int(g["hip_dims"]):
        from oracle.unet_ref import tiny15_config
        return tiny_config() if which == "sdxl" else tiny15_config()
    boc = tuple(int(v) for v in g["cfg_boc"])
    heads = tuple(int(v) for v in g["cfg_heads"])
    if which == "sdxl":
        return UNetConfig(sample_size=int(g["cfg_sample"]), block_out_channels=boc,
                          transformer_layers_per_block=(1,) * len(boc), num_attention_heads=heads,
                          cross_attention_dim=int(g["cfg_cross"]), addition_time_embed_dim=int(g["cfg_add_dim"]),
                          projection_class_embeddings_input_dim=int(g["cfg_proj_in"]), name="toy")
    b = sd15_config()
    return UNetConfig(sample_size=int(g["cfg_sample"]), block_out_channels=boc, down_block_types=b.down_block_types,
                      up_block_types=b.up_block_types, transformer_layers_per_block=(1,) * len(boc),
                      num_attention_heads=heads, cross_attention_dim=int(g["cfg_cross"]),
                      use_linear_projection=False, addition_embed_type=None, addition_time_embed_dim=0,
                      projection_class_embeddings_input_dim=0, name="toy15")


def _round_bf16_(module):
    with torch.no_grad():
        for p in module.parameters():
            if p.dim() >= 2:
                p.copy_(p.to(torch.bfloat16).float())


@pytest.mark.parametrize("tag", ["sdxl_mixed", "sdxl_all_en", "sdxl_all_zh", "sd15_mixed", "sdxl_hip_mixed", "sdxl_hip_all_en",
                                 "sdxl_hip_all_zh", "sdxl_hip_shared_teacher", "sd15_hip_mixed"])
def test_training_step_vs_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"step_{tag}.npz"))
    which = "sdxl" if tag.startswith("sdxl") else "sd15"
    cfg = _cfg_from(g, which)
    torch.manual_seed(int(g["seed_model"]))
    us, ut = UNet2DConditionRef(cfg), UNet2DConditionRef(cfg)
    if int(g["shared_teacher"]):
        ut.load_state_dict(us.state_dict())
    if int(g["hip_dims"]):                # the *_hip_* fixtures were generated on bf16-representable weights
        _round_bf16_(us), _round_bf16_(ut)
    assert abs(_wsum(us.state_dict()) + _wsum(ut.state_dict()) - float(g["wsum_unets"])) < 1e-6 * float(g["wsum_unets"])
    ad = _adapter_from(g["mlp_args"])
    ad.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("w.")})
    for p in us.parameters():
        p.requires_grad_(False)         # the build freezes the student UNet (adapter-only backward)
    batch = {k: T(g[k]) for k in ["latents", "noise", "timesteps", "enc", "enc_uncond", "prompt_mask", "zh_or_not",
                                  "teacher_ehs", "teacher_neg", "teacher_pooled", "time_ids"]}
    out = training_step_ref(ad, us, ut, batch, cast_hook_ref, nan_guard=(which == "sd15"))
    assert list(out["taps_s"].keys()) == [str(k) for k in g["tap_keys"]] == tap_names(cfg)
    for k in ["loss", "train_loss", "train_loss_logits", "train_loss_features"]:
        assert abs(float(out[k]) - float(g[k])) <= 2e-5 * max(1.0, abs(float(g[k]))), k
    out["loss"].backward()
    for k, p in ad.named_parameters():
        torch.testing.assert_close(p.grad, T(g["g." + k]), rtol=2e-4, atol=2e-6)
    assert int(g["unet_wgrad_populated"]) == 1   # documented reference quirk
    torch.testing.assert_close(out["noise_pred"], T(g["noise_pred"]), rtol=1e-4, atol=1e-5)


def test_rescale_noise_cfg(golden_dir):
    g = np.load(os.path.join(golden_dir, "rescale_noise_cfg.npz"))
    for gr in (0.0, 0.3, 0.7):
        out = rescale_noise_cfg_ref(T(g["noise_cfg"]), T(g["noise_pred_text"]), gr)
        torch.testing.assert_close(out, T(g[f"out_{gr}"]), rtol=1e-6, atol=1e-6)


def test_unet_known_answers():
    assert count_params_analytic(sdxl_config()) == 2_567_463_684
    assert count_params_analytic(sd15_config()) == 859_520_964
    with torch.device("meta"):
        m = UNet2DConditionRef(sdxl_config())
    pc = lambda mod: sum(p.numel() for p in mod.parameters())
    blocks = [pc(b) for b in m.down_blocks] + [pc(m.mid_block)] + [pc(b) for b in m.up_blocks]
    want = [5.4, 60.1, 757.4, 413.1, 1206.6, 106.3, 11.2]            # SURVEY 8(c), M params
    for got, w in zip(blocks, want):
        assert abs(got / 1e6 - w) < 0.06, (got, w)


def test_tap_shapes_tiny():
    cfg = tiny_config()
    torch.manual_seed(0)
    m = UNet2DConditionRef(cfg)
    st = {}
    cast_hook_ref(m, st)
    B = 2
    with torch.no_grad():
        m(torch.randn(B, 4, 16, 16), torch.tensor([1, 999]), torch.randn(B, 5, cfg.cross_attention_dim),
          added_cond_kwargs={"text_embeds": torch.randn(B, cfg.pooled_dim),
                             "time_ids": torch.tensor([[128, 128, 0, 0, 128, 128]] * B)})
    assert list(st.keys()) == tap_names(cfg)
    assert st["d0"].shape == (B, 64, 8, 8) and st["m"].shape == (B, 128, 4, 4) and st["u2"].shape == (B, 64, 16, 16)


def test_kd_loss_nan_guard_and_masks():
    torch.manual_seed(0)
    a, b, c = torch.randn(3, 4, 4, 4), torch.randn(3, 4, 4, 4), torch.randn(3, 4, 4, 4)
    fs, ft = [torch.randn(3, 8, 2, 2), torch.randn(3, 8, 2, 2)], [torch.randn(3, 8, 2, 2), torch.randn(3, 8, 2, 2)]
    zh = torch.tensor([1, 0, 0])
    tot, l0, l1, l2 = kd_losses(a, b, c, fs, ft, zh)
    # masks are NOT renormalised: mean over all B samples (train_sdxl_zh.py:405,417,429)
    assert abs(float(l0) - float(((a - b) ** 2)[0].mean() / 3)) < 1e-6
    assert abs(float(l1) - float((((a - c) ** 2)[1:].mean([1, 2, 3])).sum() / 3)) < 1e-6
    assert abs(float(tot) - float(l0 + l1 + 0.1 * l2)) < 1e-6
    ft[1][1, 0, 0, 0] = float("nan")
    _, _, _, l2g = kd_losses(a, b, c, fs, ft, zh, nan_guard=True)
    _, _, _, l2_first = kd_losses(a, b, c, fs[:1], ft[:1], zh)
    assert abs(float(l2g) - float(l2_first)) < 1e-7


def test_ssd1b_layout_known_answers_and_config_json():
    """BASELINE config 4 / tests/test_sdxl_zh.py:449-454 (SSD-1B as the downstream UNet): per-position transformer depths,
    a reverse (up-path) list and no mid block.  Known answers: the model card's 1.3 B parameters; the diffusers rule for
    the depth tables; the product's config reader agrees with the oracle's on the same `config.json` dictionary."""
    from oracle.unet_ref import ssd1b_config, tap_names
    from pea_diffusion_amd import config as pc
    assert count_params_analytic(ssd1b_config()) == 1_300_195_844
    assert tap_names(ssd1b_config()) == ["d0", "d1", "d2", "u0", "u1", "u2"]
    js = {"block_out_channels": [320, 640, 1280], "down_block_types": ["DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"],
          "up_block_types": ["CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"], "layers_per_block": 2,
          "transformer_layers_per_block": [1, [2, 2], [4, 4]], "reverse_transformer_layers_per_block": [[4, 4, 10], [2, 1, 1], 1],
          "mid_block_type": None, "attention_head_dim": [5, 10, 20], "cross_attention_dim": 2048, "use_linear_projection": True,
          "addition_embed_type": "text_time", "addition_time_embed_dim": 256, "projection_class_embeddings_input_dim": 2816,
          "sample_size": 128, "in_channels": 4, "out_channels": 4, "norm_num_groups": 32, "norm_eps": 1e-5}
    c = pc.unet_config_from_diffusers(js)
    assert pc.depth_tables(c) == ([[1, 1], [2, 2], [4, 4]], [[4, 4, 10], [2, 1, 1], [1, 1, 1]], -1)
    assert pc.depth_tables(pc.ssd1b_config()) == pc.depth_tables(c)
    cc = pc.to_c(c)
    assert cc.per_layer_depth == 1 and cc.depth_mid == -1 and [cc.depth_up[0][j] for j in range(3)] == [4, 4, 10]
    # uniform configs: ints broadcast, the up path mirrors the down path, the mid block takes the last entry
    assert pc.depth_tables(pc.sdxl_config()) == ([[1, 1], [2, 2], [10, 10]], [[10, 10, 10], [2, 2, 2], [1, 1, 1]], 10)
    # a nested last entry WITH a mid block: diffusers hands transformer_layers_per_block[-1] to UNetMidBlock2DCrossAttn,
    # which reads element [0] for its single layer
    import dataclasses
    nested = dataclasses.replace(pc.sdxl_config(), transformer_layers_per_block=(1, 2, (4, 10)),
                                 reverse_transformer_layers_per_block=((10, 4, 4), 2, 1))
    assert pc.depth_tables(nested)[2] == 4 and pc.depth_tables(nested)[0][2] == [4, 10]
    # ... and the oracle built from that SAME nested config agrees with the product's table at every position, mid included
    from oracle.unet_ref import sdxl_config as o_sdxl
    ocfg = o_sdxl()
    ocfg.transformer_layers_per_block, ocfg.reverse_transformer_layers_per_block = (1, 2, (4, 10)), ((10, 4, 4), 2, 1)
    with torch.device("meta"):
        mo = UNet2DConditionRef(ocfg)
    d_, u_, mid_ = pc.depth_tables(nested)
    assert len(mo.mid_block.attentions[0].transformer_blocks) == mid_ == 4
    assert [len(a.transformer_blocks) for a in mo.down_blocks[2].attentions] == d_[2]
    assert [len(a.transformer_blocks) for a in mo.up_blocks[0].attentions] == u_[0]
    sd = dict(js, transformer_layers_per_block=[1, 2, 10], reverse_transformer_layers_per_block=None,
              mid_block_type="UNetMidBlock2DCrossAttn")
    assert pc.depth_tables(pc.unet_config_from_diffusers(sd)) == pc.depth_tables(pc.sdxl_config())
    with pytest.raises(ValueError):
        pc.depth_tables(pc.unet_config_from_diffusers(dict(js, reverse_transformer_layers_per_block=None)))
    # the oracle built from the same nested config has exactly the per-position stacks
    with torch.device("meta"):
        m = UNet2DConditionRef(ssd1b_config())
    assert m.mid_block is None
    assert [len(a.transformer_blocks) for a in m.up_blocks[0].attentions] == [4, 4, 10]
    assert [len(a.transformer_blocks) for a in m.up_blocks[1].attentions] == [2, 1, 1]
    assert [len(a.transformer_blocks) for a in m.down_blocks[2].attentions] == [4, 4]


def test_bf16_storage_mode_is_identity_when_off_and_rounds_values_and_gradients_when_on():
    """oracle/bf16_store.py: outside the context `st` must not change a single bit (every golden fixture was generated with the fp32
    oracle); inside, the forward value and the gradient flowing back through the same point are bf16-rounded."""
    import torch
    from oracle.bf16_store import bf16_storage, enabled, st
    x = (torch.randn(64, generator=torch.Generator().manual_seed(0)) * 3).requires_grad_(True)
    assert not enabled() and st(x) is x
    with bf16_storage():
        assert enabled()
        y = st(x)
        assert torch.equal(y.detach(), x.detach().to(torch.bfloat16).float())
        g = torch.randn(64, generator=torch.Generator().manual_seed(1))
        y.backward(g)
        assert torch.equal(x.grad, g.to(torch.bfloat16).float())
    assert not enabled()


def test_oracle_subpixel_upsampler_restatement_is_exact():
    """The bf16-storage mode of the oracle restates Upsample2D's conv as four 2 x 2 kernels of summed taps (what the HIP path
    stores and executes: elementwise.hip pack_conv_subpix_kernel, gemm.hip conv_tap).  In float64 that restatement must equal
    conv3x3(interpolate(x, 2x nearest)) to rounding, forward and input gradient, on ragged sizes."""
    import torch
    from oracle.unet_ref import Upsample2D
    torch.manual_seed(0)
    for (B, C, H, W) in [(2, 8, 5, 7), (1, 4, 1, 1), (3, 6, 2, 9)]:
        m = Upsample2D(C).double()
        x = torch.randn(B, C, H, W, dtype=torch.double, requires_grad=True)
        y0, y1 = m(x), m._subpixel(x)
        assert y0.shape == y1.shape == (B, C, 2 * H, 2 * W)
        assert (y0 - y1).abs().max().item() < 1e-12
        s = torch.randn_like(y0)
        g0, = torch.autograd.grad((y0 * s).sum(), x)
        g1, = torch.autograd.grad((y1 * s).sum(), x)
        assert (g0 - g1).abs().max().item() < 1e-12

