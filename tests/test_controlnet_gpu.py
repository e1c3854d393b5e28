"""-m gpu: ControlNet on the HIP tape (C ABI pea_controlnet_*) against oracle/controlnet_ref.py, and the ControlNet
denoise loop of tests/test_sdxl_zh_controlnet.py:478-553 (BASELINE config 5 shape case) on the tiny models."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from test_model_gpu import cond_inputs, gpu, make_pair, rel_l2, round_weights_bf16_  # noqa: E402,F401


def _cn_pair(B, L, seed=3):
    from oracle.controlnet_ref import ControlNetRef
    from oracle.unet_ref import tiny_config
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.controlnet import HipControlNet
    cfg = tiny_config()
    torch.manual_seed(seed)
    ref = ControlNetRef(cfg)
    with torch.no_grad():                       # upstream zero-initialises these; random here so they matter
        for p in ref.parameters():
            if p.dim() == 1:
                p.mul_(0.5)
    round_weights_bf16_(ref)
    hip = HipControlNet(pc.tiny_config(), B, cfg.sample_size, cfg.sample_size, L)
    assert set(hip.weight_table()) == set(ref.state_dict())
    missing, unexpected = hip.load_state_dict(ref.state_dict())
    assert not missing and not unexpected
    return cfg, ref, hip


def test_controlnet_forward_vs_oracle(gpu):
    B, L = 2, 77
    cfg, ref, hip = _cn_pair(B, L)
    x, t, ehs, added = cond_inputs(cfg, B, L, cfg.sample_size)
    g = torch.Generator().manual_seed(9)
    img = torch.rand(B, 3, 8 * cfg.sample_size, 8 * cfg.sample_size, generator=g)
    ehs = ehs.to(torch.bfloat16).float()
    with torch.no_grad():
        dref, mref = ref(x, t, ehs, img, conditioning_scale=0.8, added_cond_kwargs=added)
    cadd = {k: v.cuda() for k, v in added.items()}
    down, mid = hip(x.cuda(), t.cuda(), encoder_hidden_states=ehs.cuda(), controlnet_cond=img.cuda(),
                    conditioning_scale=0.8, guess_mode=False, added_cond_kwargs=cadd, return_dict=False)
    assert len(down) == len(dref) == 9
    for i, (a, b) in enumerate(zip(down + [mid], dref + [mref])):
        e = rel_l2(a, b)
        print(f"   controlnet residual {i}: {tuple(a.shape)} rel_l2={e:.3e}")
        assert a.shape == b.shape and e < 2e-2, i
    # guess_mode: log-spaced residual weights (diffusers 0.23; the reference passes the flag through, :516)
    with torch.no_grad():
        dg, mg = ref(x, t, ehs, img, conditioning_scale=0.8, guess_mode=True, added_cond_kwargs=added)
    down_g, mid_g = hip(x.cuda(), t.cuda(), encoder_hidden_states=ehs.cuda(), controlnet_cond=img.cuda(),
                        conditioning_scale=0.8, guess_mode=True, added_cond_kwargs=cadd, return_dict=False)
    for a, b in zip(down_g + [mid_g], dg + [mg]):
        assert rel_l2(a, b) < 2e-2
    assert abs(float(down_g[0].abs().mean() / down[0].abs().mean()) - 0.1) < 1e-3
    # the cached conditioning embedding is reused for the same image tensor and recomputed for a new one
    down2, _ = hip(x.cuda(), t.cuda(), ehs.cuda(), img.cuda(), 0.8, added_cond_kwargs=cadd)
    img_c = img.cuda()
    a1, _ = hip(x.cuda(), t.cuda(), ehs.cuda(), img_c, 1.0, added_cond_kwargs=cadd)
    a2, _ = hip(x.cuda(), t.cuda(), ehs.cuda(), img_c, 1.0, added_cond_kwargs=cadd)
    assert all(torch.equal(p, q) for p, q in zip(a1, a2))
    b1, _ = hip(x.cuda(), t.cuda(), ehs.cuda(), (1.0 - img).cuda(), 1.0, added_cond_kwargs=cadd)
    with torch.no_grad():
        d_inv, _ = ref(x, t, ehs, 1.0 - img, conditioning_scale=1.0, added_cond_kwargs=added)
    assert not torch.equal(b1[0], a1[0]) and rel_l2(b1[0], d_inv[0]) < 2e-2 and rel_l2(b1[-1], d_inv[-1]) < 2e-2
    # a fresh conditioning tensor per generation (prepare_image() in the reference pipeline): the caching allocator
    # hands the next image the address of the freed one at in-place version 0 -- the cache must not mistake it for the old
    prev = None
    for k in range(3):
        cimg = ((img + 0.37 * k) % 1.0).cuda()
        addr = cimg.data_ptr()
        r, _ = hip(x.cuda(), t.cuda(), ehs.cuda(), cimg, 1.0, added_cond_kwargs=cadd)
        with torch.no_grad():
            d_k, _ = ref(x, t, ehs, (img + 0.37 * k) % 1.0, conditioning_scale=1.0, added_cond_kwargs=added)
        assert rel_l2(r[0], d_k[0]) < 2e-2, (k, addr)
        if prev is not None:
            assert not torch.equal(prev, r[0])
        prev = r[0].clone()
        del cimg, r


def test_controlnet_denoise_loop_vs_oracle(gpu):
    from oracle.sampler_ref import DPMSolverMultistepRef, denoise_ref
    from oracle.unet_ref import tiny_config
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.sampler import DPMSolverMultistep, denoise
    from pea_diffusion_amd.unet import HipUNet
    B, L = 1, 77
    cfg, uref, plain = make_pair(tiny_config, 2 * B, L, needs_grad=False)
    unet = HipUNet(pc.tiny_config(), 2 * B, cfg.sample_size, cfg.sample_size, L, residual_inputs=True, share_weights_from=plain)
    _, cref, cn = _cn_pair(2 * B, L)
    x, _, ehs, added = cond_inputs(cfg, 2 * B, L, cfg.sample_size)
    ehs = ehs.to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(11)
    img = torch.rand(B, 3, 8 * cfg.sample_size, 8 * cfg.sample_size, generator=g)
    img2 = torch.cat([img] * 2)
    scale = 0.5
    with torch.no_grad():
        want = denoise_ref(lambda *a, **k: uref(*a, **k), DPMSolverMultistepRef(), x[:B].clone(), ehs, added,
                           num_inference_steps=5, guidance_scale=5.0,
                           residual_fn=lambda xi, t: cref(xi, int(t), ehs, img2, scale, added_cond_kwargs=added))
    cadd = {k: v.cuda() for k, v in added.items()}
    img2c = img2.cuda()

    def via_tensors(xi, t):                    # the reference's data flow: NCHW tensors from controlnet(...) into unet(...)
        return cn(xi, t, encoder_hidden_states=ehs.cuda(), controlnet_cond=img2c, conditioning_scale=scale,
                  guess_mode=False, added_cond_kwargs=cadd, return_dict=False)
    got = denoise(unet, DPMSolverMultistep(), x[:B].cuda(), ehs.cuda(), cadd, num_inference_steps=5, guidance_scale=5.0,
                  residual_fn=via_tensors)
    e = rel_l2(got, want)
    print(f"[controlnet denoise loop tiny, 5 steps] latents rel_l2={e:.3e}")
    assert torch.isfinite(got).all() and e < 3e-2

    class Fed:                                 # device-to-device hand-over instead of NCHW tensors
        def __call__(self, xi, t, encoder_hidden_states=None, added_cond_kwargs=None, return_dict=False):
            cn.run(xi, t, encoder_hidden_states, img2c, added_cond_kwargs)
            cn.feed(unet, scale)
            return unet(xi, t, encoder_hidden_states=encoder_hidden_states, added_cond_kwargs=added_cond_kwargs)
    got2 = denoise(Fed(), DPMSolverMultistep(), x[:B].cuda(), ehs.cuda(), cadd, num_inference_steps=5, guidance_scale=5.0)
    unet.clear_residuals()
    e2 = rel_l2(got2, want)
    print(f"[controlnet denoise loop tiny, fed device-to-device] latents rel_l2={e2:.3e}")
    assert e2 < 3e-2


def test_sdxl_controlnet_full_size_steps(gpu):
    """BASELINE config 5 at full size: SDXL UNet + ControlNet, 1024x1024, one image with CFG (batch 2), 3 DPM-Solver++
    steps on random-init weights: finite latents, bit-reproducible, and the ControlNet changes the result."""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.controlnet import HipControlNet
    from pea_diffusion_amd.sampler import DPMSolverMultistep, denoise
    from pea_diffusion_amd.unet import HipUNet
    cfg = pc.sdxl_config()
    unet = HipUNet(cfg, 2, 128, 128, 77, residual_inputs=True)
    unet.init_random(1)
    cn = HipControlNet(cfg, 2, 128, 128, 77)
    cn.init_random(2)
    assert cn.output_shapes() == [(320, 128, 128)] * 3 + [(320, 64, 64)] + [(640, 64, 64)] * 2 + [(640, 32, 32)] + [(1280, 32, 32)] * 3
    assert cn.output_shapes() == unet.residual_shapes()
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(1, 4, 128, 128, generator=g).cuda()
    ehs = torch.randn(2, 77, 2048, generator=g).cuda().to(torch.bfloat16)
    added = {"text_embeds": torch.randn(2, 1280, generator=g).cuda().to(torch.bfloat16),
             "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * 2).cuda()}
    img = torch.cat([(torch.rand(1, 3, 1024, 1024, generator=g) > 0.9).float()] * 2).cuda()

    class Pipe:
        use_cn = True
        def __call__(self, x, t, encoder_hidden_states=None, added_cond_kwargs=None, return_dict=False):
            if self.use_cn:
                cn.run(x, t, encoder_hidden_states, img, added_cond_kwargs)
                cn.feed(unet, 1.0)
            return unet(x, t, encoder_hidden_states=encoder_hidden_states, added_cond_kwargs=added_cond_kwargs)
    p = Pipe()
    a = denoise(p, DPMSolverMultistep(), lat.clone(), ehs, added, num_inference_steps=3, guidance_scale=5.0)
    b = denoise(p, DPMSolverMultistep(), lat.clone(), ehs, added, num_inference_steps=3, guidance_scale=5.0)
    assert a.shape == (1, 4, 128, 128) and torch.isfinite(a).all() and torch.equal(a, b)
    unet.clear_residuals()
    p.use_cn = False
    c = denoise(p, DPMSolverMultistep(), lat.clone(), ehs, added, num_inference_steps=3, guidance_scale=5.0)
    assert torch.isfinite(c).all() and not torch.equal(a, c)


def test_sdxl_controlnet_full_size_step_vs_oracle(gpu):
    """BASELINE config 5 at full size AGAINST THE ORACLE: one ControlNet + UNet evaluation of the reference's denoise loop
    (tests/test_sdxl_zh_controlnet.py:510-538: `controlnet(...)` -> `unet(..., down_block_additional_residuals=...,
    mid_block_additional_residual=...)`) at 1024x1024, batch 1, full 2.57 B UNet + 1.25 B ControlNet: the ten residuals and
    eps vs oracle/controlnet_ref.py + oracle/unet_ref.py in fp32 on the host cores (about a minute)."""
    import time
    from oracle import unet_ref as ou
    from oracle.controlnet_ref import ControlNetRef
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.controlnet import HipControlNet
    from pea_diffusion_amd.unet import HipUNet
    from test_model_gpu import _fast_fill_
    torch.set_num_threads(min(64, len(__import__("os").sched_getaffinity(0))))
    cfg = ou.sdxl_config()
    B, L, hw = 1, 77, 128
    orig = torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_
    torch.nn.init.kaiming_uniform_ = lambda t, *a, **k: t
    torch.nn.init.uniform_ = lambda t, *a, **k: t
    try:
        uref, cref = ou.UNet2DConditionRef(cfg), ControlNetRef(cfg)
    finally:
        torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_ = orig
    _fast_fill_(uref, seed=7)
    _fast_fill_(cref, seed=8)
    for m in (uref, cref):
        round_weights_bf16_(m)
        for p in m.parameters():
            p.requires_grad_(False)
    unet = HipUNet(pc.sdxl_config(), B, hw, hw, L, residual_inputs=True)
    missing, unexpected = unet.load_state_dict(uref.state_dict())
    assert not missing and not unexpected
    cn = HipControlNet(pc.sdxl_config(), B, hw, hw, L)
    assert set(cn.weight_table()) == set(cref.state_dict())
    missing, unexpected = cn.load_state_dict(cref.state_dict())
    assert not missing and not unexpected
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 4, hw, hw, generator=g)
    t = torch.tensor([601])
    ehs = torch.randn(B, L, 2048, generator=g).to(torch.bfloat16).float()
    added = {"text_embeds": torch.randn(B, 1280, generator=g).to(torch.bfloat16).float(),
             "time_ids": torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * B)}
    img = (torch.rand(B, 3, 8 * hw, 8 * hw, generator=g) > 0.9).float()        # canny-like sparse edges
    scale = 0.5
    t0 = time.time()
    with torch.no_grad():
        dref, mref = cref(x, t, ehs, img, conditioning_scale=scale, added_cond_kwargs=added)
        eps_ref = uref(x, t, ehs, added_cond_kwargs=added, down_block_additional_residuals=dref,
                       mid_block_additional_residual=mref)[0]
        eps_plain = uref(x, t, ehs, added_cond_kwargs=added)[0]
    t_or = time.time() - t0
    cadd = {k: v.cuda() for k, v in added.items()}
    down, mid = cn(x.cuda(), t.cuda(), encoder_hidden_states=ehs.cuda(), controlnet_cond=img.cuda(),
                   conditioning_scale=scale, guess_mode=False, added_cond_kwargs=cadd, return_dict=False)
    worst = 0.0
    for i, (a, b) in enumerate(zip(down + [mid], dref + [mref])):
        e = rel_l2(a, b)
        worst = max(worst, e)
        assert a.shape == b.shape and e < 1.5e-2, (i, e)
    eps = unet(x.cuda(), t.cuda(), ehs.cuda(), added_cond_kwargs=cadd, down_block_additional_residuals=down,
               mid_block_additional_residual=mid, return_dict=False)[0]
    e = rel_l2(eps, eps_ref)
    # device-to-device hand-over (cn.run + cn.feed) gives the same eps as the tensor route
    cn.run(x.cuda(), t.cuda(), ehs.cuda(), img.cuda(), cadd)
    cn.feed(unet, scale)
    eps_fed = unet(x.cuda(), t.cuda(), encoder_hidden_states=ehs.cuda(), added_cond_kwargs=cadd)[0]
    e_fed = rel_l2(eps_fed, eps_ref)
    moved = rel_l2(eps_ref, eps_plain)
    print(f"[sdxl + controlnet 1024x1024 B=1 vs oracle] worst residual rel_l2={worst:.3e} eps rel_l2={e:.3e} "
          f"(fed device-to-device {e_fed:.3e}); the ControlNet moves eps by {moved:.3e}; oracle {t_or:.0f} s")
    assert e < 1.5e-2 and e_fed < 1.5e-2
    assert moved > 10 * e, "residual injection too weak to be tested by this comparison"
