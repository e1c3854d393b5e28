"""-m gpu: the inference denoise-loop pieces through the C ABI against oracle/sampler_ref.py -- CFG combine and
`rescale_noise_cfg` (the latter also against the golden vectors generated from the reference), the DPM-Solver++
update, ControlNet residual inputs of the UNet call, and the whole loop on the tiny UNet."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from test_model_gpu import cond_inputs, gpu, make_pair, rel_l2  # noqa: E402,F401


def test_cfg_combine_vs_oracle(gpu):
    from oracle.sampler_ref import cfg_combine_ref, rescale_noise_cfg_ref
    from pea_diffusion_amd import ops
    g = torch.Generator().manual_seed(0)
    for shape in [(2, 4, 8, 8), (8, 4, 128, 128), (6, 4, 33, 17)]:
        e = torch.randn(shape, generator=g) * 1.7 + 0.1
        for gs, phi in [(7.5, 0.0), (5.0, 0.7), (1.5, 1.0)]:
            ref, t = cfg_combine_ref(e, gs)
            if phi > 0:
                ref = rescale_noise_cfg_ref(ref, t, phi)
            got = ops.cfg_combine(e.cuda(), gs, phi).cpu()
            assert got.shape == ref.shape
            assert torch.allclose(got, ref, rtol=2e-5, atol=2e-5), (shape, gs, phi, (got - ref).abs().max())


def test_rescale_noise_cfg_golden(gpu, golden_dir):
    """tests/golden/rescale_noise_cfg.npz comes from the reference's own function (oracle/make_golden.py).  The HIP
    kernel forms noise_cfg itself, so feed it u = 2 t - noise_cfg with guidance 2: u + 2 (t - u) = noise_cfg."""
    from pea_diffusion_amd import ops
    z = np.load(os.path.join(golden_dir, "rescale_noise_cfg.npz"))
    cfg, t = torch.from_numpy(z["noise_cfg"]).float(), torch.from_numpy(z["noise_pred_text"]).float()
    e = torch.cat([2 * t - cfg, t]).cuda()
    for phi in (0.0, 0.3, 0.7):
        got = ops.cfg_combine(e, 2.0, phi).cpu()
        assert torch.allclose(got, torch.from_numpy(z[f"out_{phi}"]).float(), rtol=1e-4, atol=1e-5), phi


def test_dpm_update_vs_oracle(gpu):
    from oracle.sampler_ref import DPMSolverMultistepRef
    from pea_diffusion_amd.sampler import DPMSolverMultistep
    for n, order in [(30, 2), (8, 2), (5, 1)]:
        ref, hip = DPMSolverMultistepRef(solver_order=order), DPMSolverMultistep(solver_order=order)
        ref.set_timesteps(n)
        ts = hip.set_timesteps(n)
        g = torch.Generator().manual_seed(n)
        x_ref = torch.randn(2, 4, 16, 16, generator=g, dtype=torch.float64)
        x_hip = x_ref.float().cuda()
        for t in ts:
            eps = torch.randn(2, 4, 16, 16, generator=g)
            x_ref = ref.step(eps.double(), t, x_ref)[0]
            x_hip = hip.step(eps.cuda(), t, x_hip)[0]
            assert torch.allclose(x_hip.cpu().double(), x_ref, rtol=1e-4, atol=1e-4), (n, int(t))


def test_unet_residual_inputs_vs_oracle(gpu):
    from oracle.unet_ref import tiny_config
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.unet import HipUNet
    B, L = 2, 77
    cfg, ref, plain = make_pair(tiny_config, B, L, needs_grad=False)
    hip = HipUNet(pc.tiny_config(), B, cfg.sample_size, cfg.sample_size, L, residual_inputs=True, share_weights_from=plain)
    x, t, ehs, added = cond_inputs(cfg, B, L, cfg.sample_size)
    cadd = {k: v.cuda() for k, v in added.items()}
    shapes = hip.residual_shapes()
    g = torch.Generator().manual_seed(5)
    res = [(0.5 * torch.randn(B, *s, generator=g)).to(torch.bfloat16).float() for s in shapes]
    with torch.no_grad():
        e_plain = ref(x, t, ehs.to(torch.bfloat16).float(), added_cond_kwargs=added)[0]
        e_res = ref(x, t, ehs.to(torch.bfloat16).float(), added_cond_kwargs=added,
                    down_block_additional_residuals=res[:-1], mid_block_additional_residual=res[-1])[0]
    assert rel_l2(e_res, e_plain) > 0.05                                  # the residuals matter
    got0 = hip(x.cuda(), t.cuda(), ehs.cuda(), added_cond_kwargs=cadd)[0]
    got_plain = plain(x.cuda(), t.cuda(), ehs.cuda(), added_cond_kwargs=cadd)[0]
    assert torch.equal(got0, got_plain)                                   # unset residuals are zeros: bit-identical
    got = hip(x.cuda(), t.cuda(), ehs.cuda(), added_cond_kwargs=cadd,
              down_block_additional_residuals=[r.cuda() for r in res[:-1]], mid_block_additional_residual=res[-1].cuda())[0]
    e = rel_l2(got, e_res)
    print(f"[unet residual inputs] {len(shapes)} residuals, eps rel_l2={e:.3e}")
    assert e < 2e-2
    again = hip(x.cuda(), t.cuda(), ehs.cuda(), added_cond_kwargs=cadd)[0]  # a call without the kwargs is plain again
    assert torch.equal(again, got_plain)
    with pytest.raises(Exception):
        plain(x.cuda(), t.cuda(), ehs.cuda(), added_cond_kwargs=cadd, mid_block_additional_residual=res[-1].cuda())


def test_denoise_loop_tiny_vs_oracle(gpu):
    from oracle.sampler_ref import DPMSolverMultistepRef, denoise_ref
    from oracle.unet_ref import tiny_config
    from pea_diffusion_amd.sampler import DPMSolverMultistep, denoise
    B, L = 2, 77
    cfg, ref, hip = make_pair(tiny_config, 2 * B, L, needs_grad=False)
    x, _, ehs, added = cond_inputs(cfg, 2 * B, L, cfg.sample_size)
    lat = x[:B]
    ehs = ehs.to(torch.bfloat16).float()
    with torch.no_grad():
        want = denoise_ref(lambda *a, **k: ref(*a, **k), DPMSolverMultistepRef(), lat.clone(), ehs, added,
                           num_inference_steps=6, guidance_scale=5.0, guidance_rescale=0.7)
    got = denoise(hip, DPMSolverMultistep(), lat.cuda(), ehs.cuda(), {k: v.cuda() for k, v in added.items()},
                  num_inference_steps=6, guidance_scale=5.0, guidance_rescale=0.7)
    e = rel_l2(got, want)
    print(f"[denoise loop tiny, 6 steps, cfg 5.0, rescale 0.7] latents rel_l2={e:.3e}")
    assert torch.isfinite(got).all() and e < 3e-2


def test_end_to_end_generation_tiny_vs_oracle(gpu):
    """The reference's generation program end to end on tiny models (tests/test_sdxl_zh.py `StableDiffusionTest.__call__`,
    :153-290 encode_prompt through the Chinese-CLIP tower and the adapter, :350-406 denoise loop, :430 VAE decode): every
    stage on the HIP path against the same chain of CPU oracles."""
    from oracle.sampler_ref import DPMSolverMultistepRef, denoise_ref
    from oracle.step_ref import AdapterRef
    from oracle.text_ref import BertTextRef
    from oracle.unet_ref import tiny_config
    from oracle.vae_ref import VAEDecoderRef, tiny_vae_config
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.adapter import PEAAdapter
    from pea_diffusion_amd.sampler import DPMSolverMultistep, denoise
    from pea_diffusion_amd.text import HipTextEncoder
    from pea_diffusion_amd.vae import HipVAEDecoder
    from test_model_gpu import round_weights_bf16_
    B, L = 2, 52
    torch.manual_seed(0)
    # --- text tower (BERT flavour, width 128) -> adapter (128 -> pooled / 128-wide tokens = the tiny UNet's cross dim)
    tcfg = pc.tiny_bert_config()
    t_ref = BertTextRef(tcfg)
    with torch.no_grad():
        for p in t_ref.parameters():
            if p.dim() >= 2:
                p.mul_(3.0)
    round_weights_bf16_(t_ref)
    t_hip = HipTextEncoder(tcfg, 2 * B, L)
    t_hip.load_state_dict(t_ref.state_dict())
    cfg, u_ref, u_hip = make_pair(tiny_config, 2 * B, L, needs_grad=False)
    a_ref = AdapterRef(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    a_hip = PEAAdapter(128, cfg.pooled_dim, 192, cfg.cross_attention_dim, False)
    a_hip.load_state_dict(a_ref.state_dict())
    a_hip = a_hip.cuda()
    round_weights_bf16_(a_ref)
    vcfg = tiny_vae_config()
    v_ref = VAEDecoderRef(vcfg)
    round_weights_bf16_(v_ref)
    v_hip = HipVAEDecoder(pc.tiny_vae_config(), B, cfg.sample_size, cfg.sample_size)
    v_hip.load_state_dict(v_ref.state_dict())
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 1000, (2 * B, L), generator=g)          # [negative prompts | prompts]
    ids[:, 30:] = 0
    lat = torch.randn(B, 4, cfg.sample_size, cfg.sample_size, generator=g)
    time_ids = torch.tensor([[128, 128, 0, 0, 128, 128]] * (2 * B))
    with torch.no_grad():
        tok = t_ref(ids)["last_hidden_state"].to(torch.bfloat16).float()
        pooled, tokens = a_ref(tok)
        want_lat = denoise_ref(lambda *a, **k: u_ref(*a, **k), DPMSolverMultistepRef(), lat.clone(),
                               tokens.to(torch.bfloat16).float(), {"text_embeds": pooled, "time_ids": time_ids},
                               num_inference_steps=4, guidance_scale=5.0)
        want_img = v_ref.decode(want_lat / vcfg.scaling_factor)[0]
    tok_h, _ = t_hip.encode_text(ids.cuda())
    pooled_h, tokens_h = a_hip(tok_h)
    got_lat = denoise(u_hip, DPMSolverMultistep(), lat.cuda(), tokens_h, {"text_embeds": pooled_h, "time_ids": time_ids.cuda()},
                      num_inference_steps=4, guidance_scale=5.0)
    got_img = v_hip.decode(got_lat, inv_scaling=1.0 / vcfg.scaling_factor)[0]
    e_tok, e_lat, e_img = rel_l2(tokens_h, tokens), rel_l2(got_lat, want_lat), rel_l2(got_img, want_img)
    print(f"[end to end tiny] adapter tokens rel_l2={e_tok:.3e} latents rel_l2={e_lat:.3e} image rel_l2={e_img:.3e}")
    assert e_tok < 2e-2 and e_lat < 3e-2 and e_img < 4e-2 and torch.isfinite(got_img).all()
