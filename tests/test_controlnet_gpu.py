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
