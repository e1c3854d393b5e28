"""CPU checks of the VAE-encoder restatement (oracle/vae_ref.py): structural known-answers of the published SDXL VAE
(parameter totals, latent geometry, FLOPs of SURVEY 8f row 2) and the posterior semantics."""
import torch

from oracle.vae_ref import (DiagonalGaussianRef, VAEEncoderRef, sdxl_vae_config, tiny_vae_config, vae_encoder_flops)


def test_sdxl_vae_encoder_parameter_total():
    with torch.device("meta"):
        m = VAEEncoderRef(sdxl_vae_config())
    n_enc = sum(p.numel() for k, p in m.named_parameters() if k.startswith("encoder."))
    n_q = sum(p.numel() for k, p in m.named_parameters() if k.startswith("quant_conv."))
    # published AutoencoderKL (sdxl-vae) total 83 653 863 = encoder 34 163 592 + quant 72 + decoder 49 490 179 + post_quant 20
    assert (n_enc, n_q) == (34_163_592, 72)
    assert abs(vae_encoder_flops(sdxl_vae_config(), 1024, 1024) / 1e12 - 4.879) < 5e-3      # 4.89 TFLOP/img


def test_sdxl_vae_decoder_parameter_total():
    from oracle.vae_ref import VAEDecoderRef
    with torch.device("meta"):
        m = VAEDecoderRef(sdxl_vae_config())
    n_dec = sum(p.numel() for k, p in m.named_parameters() if k.startswith("decoder."))
    n_pq = sum(p.numel() for k, p in m.named_parameters() if k.startswith("post_quant_conv."))
    assert (n_dec, n_pq) == (49_490_179, 20)          # encoder 34 163 592 + 72 + these = 83 653 863, the published total


def test_encode_geometry_and_posterior():
    torch.manual_seed(0)
    cfg = tiny_vae_config()
    m = VAEEncoderRef(cfg)
    x = torch.randn(2, 3, 64, 48)
    with torch.no_grad():
        mom = m.moments(x)
        d = m.encode(x).latent_dist
    f = 2 ** (len(cfg.block_out_channels) - 1)
    assert mom.shape == (2, 8, 64 // f, 48 // f)
    assert torch.equal(d.mean, mom[:, :4]) and torch.equal(d.mode(), d.mean)
    nz = torch.randn(d.mean.shape)
    assert torch.allclose(d.sample(noise=nz), mom[:, :4] + torch.exp(0.5 * mom[:, 4:].clamp(-30, 20)) * nz)
    big = DiagonalGaussianRef(torch.cat([torch.zeros(1, 4, 2, 2), torch.full((1, 4, 2, 2), 1e4)], 1))
    assert float(big.logvar.max()) == 20.0


def test_downsample_uses_bottom_right_padding():
    """Downsample2D(padding=0) pads (0,1,0,1): output pixel (i,j) sees input rows 2i..2i+2 -- an impulse in the last
    row/column must reach the last output pixel only through the zero padding's neighbours."""
    from oracle.vae_ref import Downsample
    d = Downsample(1)
    with torch.no_grad():
        d.conv.weight.fill_(1.0); d.conv.bias.zero_()
        x = torch.zeros(1, 1, 4, 4); x[0, 0, 0, 0] = 1.0
        y = d(x)
    assert y.shape == (1, 1, 2, 2) and y[0, 0, 0, 0] == 1.0 and y[0, 0].sum() == 1.0   # symmetric padding would give 4
