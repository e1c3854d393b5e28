"""-m gpu: the VAE encoder on the HIP tape (C ABI pea_vae_*) against oracle/vae_ref.py on the same bf16-rounded weights
and seeded pixels; tolerance as for the UNet forward (relative L2 <= 2e-2 through ~40 chained bf16 ops)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
from test_model_gpu import gpu, rel_l2, round_weights_bf16_  # noqa: E402,F401


def _pair(cfg_name, B, H, W, seed=0):
    import oracle.vae_ref as ov
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.vae import HipVAEEncoder
    torch.manual_seed(seed)
    ref = ov.VAEEncoderRef(getattr(ov, cfg_name)())
    round_weights_bf16_(ref)
    hip = HipVAEEncoder(getattr(pc, cfg_name)(), B, H, W)
    assert {k: tuple(v.shape) for k, v in ref.state_dict().items()}.keys() == hip.weight_table().keys()
    missing, unexpected = hip.load_state_dict(ref.state_dict())
    assert not missing and not unexpected
    return ref, hip


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 128, 64)])
def test_vae_encode_tiny_vs_oracle(gpu, B, H, W):
    ref, hip = _pair("tiny_vae_config", B, H, W)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, H, W, generator=g).clamp(-1, 1)
    with torch.no_grad():
        mom = ref.moments(x)
    d = hip.encode(x.cuda()).latent_dist
    e = rel_l2(d.moments, mom)
    print(f"[vae tiny {B}x3x{H}x{W}] moments {tuple(d.moments.shape)} rel_l2={e:.3e}")
    assert d.moments.shape == mom.shape and e < 2e-2
    nz = torch.randn(hip.latent_shape, generator=g)
    want = (mom[:, :4] + torch.exp(0.5 * mom[:, 4:].clamp(-30, 20)) * nz) * ref.config.scaling_factor
    got = hip.encode_latents(x.cuda(), noise=nz.cuda())
    assert rel_l2(got, want) < 2e-2
    # the fused sample equals the unfused pieces of the same run bit for bit in the mean/std arithmetic
    assert torch.allclose(got, d.sample(noise=nz.cuda()) * ref.config.scaling_factor, rtol=1e-5, atol=1e-6)
    assert torch.allclose(hip.encode_latents(x.cuda(), sample=False), d.mode() * ref.config.scaling_factor, rtol=1e-6, atol=1e-7)


def test_vae_full_checkpoint_keys_and_shape_errors(gpu):
    ref, hip = _pair("tiny_vae_config", 1, 64, 64)
    sd = dict(ref.state_dict())
    sd["decoder.conv_in.weight"] = torch.zeros(1)
    sd["post_quant_conv.bias"] = torch.zeros(4)
    assert hip.load_state_dict(sd) == ([], [])
    with pytest.raises(Exception):
        hip.encode(torch.zeros(1, 3, 32, 64).cuda())


def test_vae_sdxl_size_vs_oracle_256(gpu):
    """SDXL VAE widths (128/256/512/512, 34.2 M parameters) on a 256x256 image (1024 latent tokens through the
    materialised single-head attention), against the fp32 oracle."""
    ref, hip = _pair("sdxl_vae_config", 1, 256, 256)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 3, 256, 256, generator=g).clamp(-1, 1)
    with torch.no_grad():
        mom = ref.moments(x)
    got = hip.encode(x.cuda()).latent_dist.moments
    e = rel_l2(got, mom)
    print(f"[vae sdxl widths 256x256] rel_l2={e:.3e}")
    assert e < 2e-2


def test_vae_full_size_properties(gpu):
    """1024x1024 (16 384 latent tokens), batch 2: finite, deterministic run to run, batch elements independent."""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.vae import HipVAEEncoder
    hip = HipVAEEncoder(pc.sdxl_vae_config(), 2)
    hip.init_random(3)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 1024, 1024, generator=g).clamp(-1, 1).cuda()
    a = hip.encode(x).latent_dist.moments.clone()
    b = hip.encode(x).latent_dist.moments
    assert a.shape == (2, 8, 128, 128) and torch.isfinite(a).all() and torch.equal(a, b)
    c = hip.encode(torch.stack([x[1], x[0]])).latent_dist.moments
    assert torch.equal(c[0], a[1]) and torch.equal(c[1], a[0])


def _dec_pair(cfg_name, B, h, w, seed=0):
    import oracle.vae_ref as ov
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.vae import HipVAEDecoder
    torch.manual_seed(seed)
    ref = ov.VAEDecoderRef(getattr(ov, cfg_name)())
    round_weights_bf16_(ref)
    hip = HipVAEDecoder(getattr(pc, cfg_name)(), B, h, w)
    assert set(ref.state_dict()) == set(hip.weight_table())
    missing, unexpected = hip.load_state_dict(ref.state_dict())
    assert not missing and not unexpected
    return ref, hip


@pytest.mark.parametrize("B,h,w", [(2, 16, 16), (1, 8, 16)])
def test_vae_decode_tiny_vs_oracle(gpu, B, h, w):
    ref, hip = _dec_pair("tiny_vae_config", B, h, w)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, 4, h, w, generator=g)
    with torch.no_grad():
        want = ref.decode(z / ref.config.scaling_factor)[0]
    got = hip.decode(z.cuda() / ref.config.scaling_factor, return_dict=False)[0]
    e = rel_l2(got, want)
    print(f"[vae decode tiny {tuple(z.shape)} -> {tuple(got.shape)}] rel_l2={e:.3e}")
    assert got.shape == want.shape and e < 2e-2
    got2 = hip.decode(z.cuda(), inv_scaling=1.0 / ref.config.scaling_factor)[0]       # division inside the kernel
    assert rel_l2(got2, want) < 2e-2


def test_vae_decode_sdxl_widths_vs_oracle(gpu):
    """SDXL VAE decoder widths (512/512/256/128, 49.5 M parameters) from a 32x32 latent (256x256 image) vs the oracle."""
    ref, hip = _dec_pair("sdxl_vae_config", 1, 32, 32)
    z = torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        want = ref.decode(z)[0]
    got = hip.decode(z.cuda())[0]
    e = rel_l2(got, want)
    print(f"[vae decode sdxl widths 32x32 -> 256x256] rel_l2={e:.3e}")
    assert e < 3e-2          # 16 ResNet blocks + attention chained in bf16 on kaiming-random weights (measured 1.8e-2)


def test_vae_decode_full_size_properties(gpu):
    """128x128 latents -> 1024x1024 image, batch 2: finite, deterministic, batch elements independent; and the
    encode -> decode chain of the two contexts runs end to end."""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.vae import HipVAEDecoder
    hip = HipVAEDecoder(pc.sdxl_vae_config(), 2)
    hip.init_random(5)
    z = torch.randn(2, 4, 128, 128, generator=torch.Generator().manual_seed(6)).cuda()
    a = hip.decode(z)[0].clone()
    b = hip.decode(z)[0]
    assert a.shape == (2, 3, 1024, 1024) and torch.isfinite(a).all() and torch.equal(a, b)
    c = hip.decode(torch.stack([z[1], z[0]]))[0]
    assert torch.equal(c[0], a[1]) and torch.equal(c[1], a[0])


def test_vae_decode_batch4_full_size_matches_single(gpu):
    """Batch 4 at 1024x1024: the decoder's [4][1024][1024][256] activation is exactly 2^31 bytes, past what a 32-bit
    buffer offset from the tensor base reaches -- the conv gather must address it per row tile.  Every sample of the
    batch equals the same latent decoded alone."""
    from pea_diffusion_amd import config as pc
    from pea_diffusion_amd.vae import HipVAEDecoder
    z = torch.randn(4, 4, 128, 128, generator=torch.Generator().manual_seed(8)).cuda()
    hip4 = HipVAEDecoder(pc.sdxl_vae_config(), 4)
    hip4.init_random(5)
    a = hip4.decode(z)[0].clone()
    del hip4
    hip1 = HipVAEDecoder(pc.sdxl_vae_config(), 1)
    hip1.init_random(5)
    assert a.shape == (4, 3, 1024, 1024) and torch.isfinite(a).all()
    for i in (3, 0):
        one = hip1.decode(z[i:i + 1])[0]
        assert torch.equal(one[0], a[i]), f"sample {i} of the batch differs from its single decode"
