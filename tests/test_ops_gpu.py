"""Kernel-level parity (-m gpu): every HIP kernel family, called through the C ABI, against a
plain torch fp32 CPU reference of the same op on the same bf16-rounded inputs.

Tolerances (written here, used everywhere below):
  * fp32-stored outputs (losses, fp32 GEMM output, conv_out, add_noise):   rtol 1e-3, atol 1e-4
  * bf16-stored outputs: the reference is compared after the same final rounding; allowed
    deviation is 1 bf16 ulp (rtol 2^-7) plus atol scaled to the tensor (2^-7 * rms), because a
    differently-ordered fp32 accumulation can flip the final rounding of an element.
"""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X (torch.cuda.is_available() is False)")
    from pea_diffusion_amd import ops as o
    return o


def bfr(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def rel_l2(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).pow(2).sum().sqrt() / (ref.pow(2).sum().sqrt() + 1e-30)).item()


def close_bf16(name, got, ref, ulps=1.0):
    got = got.detach().float().cpu()
    ref = ref.detach().float()
    rms = ref.pow(2).mean().sqrt().item() + 1e-30
    err = (got - ref).abs()
    tol = ulps * (2.0 ** -7) * (ref.abs() + rms)
    bad = (err > tol).float().mean().item()
    rel_l2 = ((got - ref).pow(2).sum().sqrt() / (ref.pow(2).sum().sqrt() + 1e-30)).item()
    print(f"[{name}] max_abs={err.max().item():.3e} rel_l2={rel_l2:.3e} rms={rms:.3e} frac_bad={bad:.2e}")
    assert torch.isfinite(got).all(), name
    assert bad == 0.0 and rel_l2 < 6e-3, f"{name}: frac_bad={bad} rel_l2={rel_l2}"


def close_f32(name, got, ref, rtol=1e-3, atol=1e-4):
    got = got.detach().float().cpu()
    ref = ref.detach().float()
    err = (got - ref).abs()
    print(f"[{name}] max_abs={err.max().item():.3e} ref_max={ref.abs().max().item():.3e}")
    torch.testing.assert_close(got, ref, rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (308, 640, 2048), (4, 1280, 320),
                                   (1000, 100, 192), (2048, 5120, 640)])
def test_gemm_plain(ops, M, N, K):
    a, w = bfr(M, K, seed=1), bfr(N, K, seed=2, scale=K ** -0.5)
    out = ops.gemm(a.cuda(), w.cuda())
    close_bf16(f"gemm {M}x{N}x{K}", out, a.float() @ w.float().T)


@pytest.mark.parametrize("M,N,K", [(6144, 1280, 1280), (6144, 3840, 640), (3072, 2560, 320), (6100, 1280, 640)])
def test_gemm_192_row_tiles_by_rule(ops, M, N, K):
    """row counts that leave 128- and 256-row tiles a partial last round (6144 = a merged pass with dead teacher rows, batch 3 / 6):
    the launcher takes the 192-row tile (one-tile and persistent forms); plain, bias + residual, and a ragged M"""
    a, w = bfr(M, K, seed=1), bfr(N, K, seed=2, scale=K ** -0.5)
    bias, res = torch.randn(N), bfr(M, N, seed=3)
    ref = a.float() @ w.float().T
    close_bf16(f"gemm192 {M}x{N}x{K}", ops.gemm(a.cuda(), w.cuda()), ref)
    close_bf16(f"gemm192 bias+res {M}x{N}x{K}", ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), res=res.cuda()), ref + bias + res.float())


@pytest.mark.parametrize("M,N,K,shift", [(8192, 3840, 1280, 0.0), (4096, 1280, 1280, 4.0), (1000, 640, 320, -2.5),
                                          (77, 160, 64, 1.0)])
def test_ln_linear_folded(ops, M, N, K, shift):
    """LayerNorm folded into the consuming Linear (W' = W . gamma, s, t; statistics applied in the GEMM epilogue) against
    LayerNorm -> Linear in fp32 on the same bf16 inputs; `shift` moves the row mean away from 0 (the folded form subtracts
    mean * s[n] from the accumulator: cancellation must stay harmless).  Includes the fused GEGLU epilogue."""
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(M, K, generator=g) * 1.3 + shift).to(BF)
    w = bfr(N, K, seed=2, scale=K ** -0.5)
    gamma = 1.0 + 0.2 * torch.randn(K, generator=g)
    beta = 0.3 * torch.randn(K, generator=g)
    bias = torch.randn(N, generator=g)
    ln = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5)
    ref = ln @ w.float().T + bias
    out = ops.ln_linear(x.cuda(), gamma.cuda(), beta.cuda(), w.cuda(), bias.cuda())
    close_bf16(f"ln_linear {M}x{N}x{K} shift {shift}", out, ref, ulps=2.0)
    out_nb = ops.ln_linear(x.cuda(), gamma.cuda(), beta.cuda(), w.cuda(), None)
    close_bf16(f"ln_linear no-bias {M}x{N}x{K}", out_nb, ln @ w.float().T, ulps=2.0)
    gy, pre = ops.ln_linear(x.cuda(), gamma.cuda(), beta.cuda(), w.cuda(), bias.cuda(), geglu=True)
    close_bf16(f"ln_linear geglu preact {M}x{N}x{K}", pre, ref, ulps=2.0)
    # y = h * gelu(gate): the rounding of BOTH factors enters (W. gamma is rounded to bf16 once more than in the unfolded form)
    close_bf16(f"ln_linear geglu y {M}x{N}x{K}", gy, ref[:, 0::2] * F.gelu(ref[:, 1::2]), ulps=3.0)


def test_gemm_epilogues(ops):
    M, N, K, rpb = 192, 256, 128, 48
    a, w = bfr(M, K, seed=1), bfr(N, K, seed=2, scale=K ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    rv, res = bfr(M // rpb, N, seed=4), bfr(M, N, seed=5)
    base = a.float() @ w.float().T * 0.5 + bias + rv.float().repeat_interleave(rpb, 0)
    for act, fn in [(0, lambda t: t), (1, F.gelu), (2, F.silu)]:
        out, pre = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), rowvec=rv.cuda(), rows_per_batch=rpb, act=act,
                            res=res.cuda(), want_preact=True, alpha=0.5)
        close_bf16(f"gemm epi act={act}", out, fn(base) + res.float())
        close_bf16(f"gemm preact act={act}", pre, base)
    o32 = ops.gemm(a.cuda(), w.cuda(), out_f32=True)
    close_f32("gemm fp32 out", o32, a.float() @ w.float().T, rtol=1e-3, atol=1e-4)
    acc = torch.ones(M, N, device="cuda")
    ops.gemm(a.cuda(), w.cuda(), out=acc, accum_f32=True)
    close_f32("gemm fp32 accum", acc, a.float() @ w.float().T + 1.0)
    # accumulate into a bf16 gradient buffer: res aliases the output
    gbuf = res.clone().cuda()
    ops.gemm(a.cuda(), w.cuda(), res=gbuf, out=gbuf)
    close_bf16("gemm bf16 accumulate", gbuf, a.float() @ w.float().T + res.float())


@pytest.mark.parametrize("variant", [18, 19, 22, 23, 24, 25, 27, 28, 29, 30, 31, 33, 34, 35, 39, 40])
def test_gemm_every_variant_ragged_shapes(ops, variant):
    """Every kernel form the launcher can pick (or that an experiment switch selects), forced on shapes that do not fit its
    tiles: ragged M, N that ends inside a tile / inside a 32-column store pair, fewer and more tiles than CUs (the
    persistent kernels' tile walk), with the epilogue terms each form supports."""
    import ctypes
    from pea_diffusion_amd._lib import lib
    L = lib()
    g = torch.Generator().manual_seed(variant)
    shapes = [(308, 640, 128), (1000, 104, 192), (4100, 1288, 256), (130, 3840, 64), (33000, 336, 128), (64, 160, 640)]
    try:
        for (M, N, K) in shapes:
            a = (torch.randn(M, K, generator=g)).to(BF)
            w = (torch.randn(N, K, generator=g) * K ** -0.5).to(BF)
            bias = torch.randn(N, generator=g)
            res = torch.randn(M, N, generator=g).to(BF)
            ref = a.float() @ w.float().T
            L.pea_debug_set_gemm_variant(variant)
            out = ops.gemm(a.cuda(), w.cuda())
            close_bf16(f"v{variant} plain {M}x{N}x{K}", out, ref)
            out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda())
            close_bf16(f"v{variant} bias {M}x{N}x{K}", out, ref + bias)
            if variant not in (34, 35):            # staged / deferred epilogues: bf16 output, bias only (the launcher's rule)
                out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), act=2, res=res.cuda())
                close_bf16(f"v{variant} bias+silu+res {M}x{N}x{K}", out, F.silu(ref + bias) + res.float())
                o32 = ops.gemm(a.cuda(), w.cuda(), out_f32=True)
                close_f32(f"v{variant} fp32 {M}x{N}x{K}", o32, ref, rtol=2e-3, atol=2e-3)
    finally:
        L.pea_debug_set_gemm_variant(-1)


def test_gemm_intra_workgroup_k_split_variant(ops):
    """Variant 41 (opt-in, PEA_GEMM_KSW_MINK): eight consumer waves of 64 x 80 over half a K-step each on the 128 x 160 one-tile
    kernel, accumulators exchanged through the ring behind the loop.  Batched-load epilogue only (bf16 output; bias / residual),
    an even number of K-steps; ragged M and N, fewer and more tiles than CUs, a deep K."""
    from pea_diffusion_amd._lib import lib
    L = lib()
    g = torch.Generator().manual_seed(41)
    try:
        for (M, N, K) in [(4096, 1280, 1280), (1000, 336, 256), (300, 160, 128), (4100, 1296, 384), (520, 1280, 10240)]:
            a = torch.randn(M, K, generator=g).to(BF)
            w = (torch.randn(N, K, generator=g) * K ** -0.5).to(BF)
            bias = torch.randn(N, generator=g)
            res = torch.randn(M, N, generator=g).to(BF)
            ref = a.float() @ w.float().T
            L.pea_debug_set_gemm_variant(41)
            close_bf16(f"v41 plain {M}x{N}x{K}", ops.gemm(a.cuda(), w.cuda()), ref)
            close_bf16(f"v41 bias {M}x{N}x{K}", ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda()), ref + bias)
            out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), res=res.cuda())
            close_bf16(f"v41 bias+res {M}x{N}x{K}", out, ref + bias + res.float())
            again = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), res=res.cuda())
            assert torch.equal(out, again), "K-split GEMM is not bit-reproducible"
    finally:
        L.pea_debug_set_gemm_variant(-1)


# ------------------------------------------------------------------------------------ conv
def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,Cin,Cout,stride,ups", [(2, 16, 64, 128, 1, False), (1, 32, 320, 320, 1, False),
                                                     (2, 16, 128, 128, 2, False), (2, 8, 64, 64, 1, True),
                                                     (1, 12, 192, 64, 1, False)])
def test_conv3x3_fwd(ops, B, H, Cin, Cout, stride, ups):
    x = bfr(B, Cin, H, H, seed=1)
    w = torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(2)) * (9 * Cin) ** -0.5
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(3))
    wq = w.to(BF).float()
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, wq, bias, stride=stride, padding=1)
    wp = ops.pack_conv(wq.cuda())
    y = ops.conv3x3(_nhwc(x).cuda(), wp, bias=bias.cuda(), stride=stride, upsample2x=ups)
    close_bf16(f"conv3x3 B{B} H{H} {Cin}->{Cout} s{stride} ups{ups}", y, _nhwc(ref))


@pytest.mark.parametrize("variant", [22, 23, 24, 25, 27, 28, 29, 30, 31, 33])
def test_conv3x3_every_variant(ops, variant):
    """the implicit-GEMM gather (padding, stride 2, folded nearest-2x upsample) under every tile shape / kernel form"""
    from pea_diffusion_amd._lib import lib
    L = lib()
    try:
        for (B, H, Cin, Cout, stride, ups) in [(2, 20, 64, 192, 1, False), (3, 18, 128, 128, 2, False), (1, 12, 64, 320, 1, True)]:
            x = bfr(B, Cin, H, H, seed=variant)
            wq = (torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(2)) * (9 * Cin) ** -0.5).to(BF).float()
            bias = torch.randn(Cout, generator=torch.Generator().manual_seed(3))
            xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
            ref = F.conv2d(xin, wq, bias, stride=stride, padding=1)
            wp = ops.pack_conv(wq.cuda())
            L.pea_debug_set_gemm_variant(variant)
            y = ops.conv3x3(_nhwc(x).cuda(), wp, bias=bias.cuda(), stride=stride, upsample2x=ups)
            close_bf16(f"v{variant} conv B{B} H{H} {Cin}->{Cout} s{stride} ups{ups}", y, _nhwc(ref))
    finally:
        L.pea_debug_set_gemm_variant(-1)


def test_conv3x3_epilogue_rowvec_res(ops):
    B, H, Cin, Cout = 2, 8, 64, 128
    x = bfr(B, Cin, H, H, seed=1)
    wq = (torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(2)) * (9 * Cin) ** -0.5).to(BF).float()
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(3))
    temb, res = bfr(B, Cout, seed=4), bfr(B, H, H, Cout, seed=5)
    ref = F.conv2d(x.float(), wq, bias, padding=1) + temb.float()[:, :, None, None] + res.float().permute(0, 3, 1, 2)
    y = ops.conv3x3(_nhwc(x).cuda(), ops.pack_conv(wq.cuda()), bias=bias.cuda(), rowvec=temb.cuda(), res=res.cuda())
    close_bf16("conv3x3 + temb + res", y, _nhwc(ref))


@pytest.mark.parametrize("stride,ups", [(1, False), (2, False), (1, True)])
def test_conv3x3_dgrad(ops, stride, ups):
    B, H, Cin, Cout = 2, 8, 64, 128
    x = bfr(B, Cin, H, H, seed=1).float().requires_grad_(True)
    wq = (torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(2)) * (9 * Cin) ** -0.5).to(BF).float()
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    y = F.conv2d(xin, wq, None, stride=stride, padding=1)
    dy = bfr(*y.shape, seed=7)
    y.backward(dy.float())
    wd = ops.pack_conv(wq.cuda(), dgrad=True)            # [Cin, 9*Cout]
    dyn = _nhwc(dy).cuda()
    if stride == 2:
        dx = ops.conv3x3(dyn, wd, transposed2=True)      # zero-stuffed transposed conv at the input resolution
    elif ups:
        dx = ops.sumpool2(ops.conv3x3(dyn, wd))          # dgrad at the upsampled resolution, then 2x2 sum
    else:
        dx = ops.conv3x3(dyn, wd)
    close_bf16(f"conv3x3 dgrad s{stride} ups{ups}", dx, _nhwc(x.grad), ulps=2.0)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 8, 8, 64, 128), (1, 16, 12, 128, 64), (3, 10, 14, 64, 64), (2, 32, 32, 320, 320)])
def test_upconv_subpixel_fwd_dgrad(ops, B, H, W, Cin, Cout):
    """conv3x3(interpolate(x, 2x nearest)) in its sub-pixel form (four 2 x 2 kernels of summed taps, depth-to-space output)
    against F.conv2d on the upsampled image, forward and data gradient; the merged taps are bf16 roundings of fp32 sums, so the
    comparison carries the weight-rounding term (2^-9 relative per merged tap) on top of the output rounding"""
    x = bfr(B, Cin, H, W, seed=1).float().requires_grad_(True)
    wq = (torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(2)) * (9 * Cin) ** -0.5).to(BF).float()
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(3))
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wq, bias, padding=1)
    dy = bfr(*ref.shape, seed=7)
    ref.backward(dy.float())
    wp = ops.pack_conv_subpixel(wq.cuda())
    y = ops.upconv_subpixel(_nhwc(x.detach().to(BF)).cuda(), wp, bias=bias.cuda())
    e = rel_l2(ops.d2s_to_nhwc(y).float(), _nhwc(ref.detach()).cuda())
    wd = ops.pack_conv_subpixel(wq.cuda(), dgrad=True)
    res = bfr(B, H, W, Cin, seed=9)
    dx = ops.upconv_subpixel_dgrad(ops.nhwc_to_d2s(_nhwc(dy).cuda()), wd, res=res.cuda())
    e2 = rel_l2(dx.float(), (_nhwc(x.grad) + res.float()).cuda())
    # against the folded 3 x 3 form of the same operator (the two differ by the weight roundings of the merged taps only)
    y3 = ops.conv3x3(_nhwc(x.detach().to(BF)).cuda(), ops.pack_conv(wq.cuda()), bias=bias.cuda(), upsample2x=True)
    e3 = rel_l2(ops.d2s_to_nhwc(y).float(), y3.float())
    print(f"[upconv subpixel B{B} {H}x{W} {Cin}->{Cout}] fwd rel_l2={e:.2e} dgrad rel_l2={e2:.2e} vs 3x3 form {e3:.2e}")
    assert e < 4e-3 and e2 < 5e-3 and e3 < 5e-3


def test_concat_split_depth_to_space_operand(ops):
    B, H, W, C1, C2 = 2, 6, 10, 64, 32
    a, b = bfr(B, H, W, C1, seed=1).cuda(), bfr(B, H, W, C2, seed=2).cuda()
    y = ops.concat2(ops.nhwc_to_d2s(a), b, d2s_hw=(H, W))
    assert torch.equal(y.view(B, H, W, C1 + C2), torch.cat([a, b], dim=-1))
    dy = bfr(B * H * W, C1 + C2, seed=3).cuda()
    da0 = bfr(B, H // 2, W // 2, 4, C1, seed=4).cuda()
    da, db = da0.clone(), torch.empty(B * H * W, C2, device="cuda", dtype=BF)
    ops.split2(dy, C1, C2, da=da, db=db, accum_a=True, d2s_hw=(H, W))
    want = (ops.d2s_to_nhwc(da0).float() + dy[:, :C1].view(B, H, W, C1).float()).to(BF)
    assert torch.equal(ops.d2s_to_nhwc(da), want)
    assert torch.equal(db, dy[:, C1:].contiguous())


def test_conv_in_out(ops):
    B, H, C = 2, 16, 64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 4, H, H, generator=g)
    w_in, b_in = torch.randn(C, 4, 3, 3, generator=g) * 0.2, torch.randn(C, generator=g)
    y = ops.conv_in(x.cuda(), w_in.cuda(), b_in.cuda())
    close_bf16("conv_in", y, _nhwc(F.conv2d(x, w_in, b_in, padding=1)))
    h = bfr(B, C, H, H, seed=3)
    w_out, b_out = torch.randn(4, C, 3, 3, generator=g) * 0.05, torch.randn(4, generator=g)
    wp = ops.pack_conv_out(w_out.cuda())
    close_f32("conv_out", ops.conv_out(_nhwc(h).cuda(), wp, b_out.cuda()), F.conv2d(h.float(), w_out, b_out, padding=1))
    hh = h.float().requires_grad_(True)
    dy = torch.randn(B, 4, H, H, generator=g)
    F.conv2d(hh, w_out, b_out, padding=1).backward(dy)
    close_bf16("conv_out dgrad", ops.conv_out_dgrad(dy.cuda(), wp, C), _nhwc(hh.grad))


def _geglu_factors(pre):
    """(h, gate) interleaved fp32 -> gelu(gate), h * gelu'(gate)"""
    h, gate = pre[:, 0::2], pre[:, 1::2]
    phi = 0.5 * (1.0 + torch.erf(gate / 2 ** 0.5))
    return gate * phi, h * (phi + gate * torch.exp(-0.5 * gate * gate) / (2 * torch.pi) ** 0.5)


@pytest.mark.parametrize("form", [0, 1])
@pytest.mark.parametrize("M,N,K", [(4096, 5120, 1280), (2048, 2560, 640), (384, 1280, 320), (1000, 5120, 1280)])
def test_gemm_with_geglu_backward_epilogue(ops, M, N, K, form):
    """The FF output projection's dgrad with the GEGLU backward in its epilogue (d y never stored) against fp32 torch:
    the three tile instantiations (256x160 persistent at 512 tiles, 128x160, 64x160) and a ragged row count; both stash
    forms -- 0: the pre-activation (h, gate), 1: the backward's own factors (gelu(gate), h gelu'(gate))."""
    g = torch.Generator().manual_seed(M + N)
    a = (torch.randn(M, K, generator=g) * 0.5).to(BF)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    pre = (torch.randn(M, 2 * N, generator=g) * 1.2).to(BF)
    dy = a.float() @ w.float().t()
    fa, fb = _geglu_factors(pre.float())
    if form == 1:                                # the forward stored the two factors, rounded to bf16
        stash = torch.stack([fa, fb], -1).reshape(M, 2 * N).to(BF)
        fa, fb = stash.float()[:, 0::2], stash.float()[:, 1::2]
        got = ops.gemm_geglu_bwd(a.cuda(), w.cuda(), stash.cuda(), form=1)
    else:
        got = ops.gemm_geglu_bwd(a.cuda(), w.cuda(), pre.cuda())
    ref = torch.stack([dy * fa, dy * fb], -1).reshape(M, 2 * N)
    close_bf16(f"gemm + geglu bwd form {form} {M}x{N}x{K}", got, ref)


@pytest.mark.parametrize("stash_grad", [1, 0])
@pytest.mark.parametrize("M,N,K", [(8192, 10240, 1280), (1000, 2560, 320), (256, 1280, 640)])
def test_gemm_geglu_forward_and_stash_forms(ops, M, N, K, stash_grad):
    """GEGLU in the FF projection's epilogue (diffusers GEGLU.forward) and what it leaves for the backward: the
    pre-activation (h, gate), or -- stash_grad -- the backward's factors (gelu(gate), h gelu'(gate)); rows >= stash_rows
    (the teacher half of a merged pass) are not stashed."""
    g = torch.Generator().manual_seed(M + K)
    a = (torch.randn(M, K, generator=g)).to(BF)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    bias = torch.randn(N, generator=g) * 0.1
    rows = M // 2 if M % 128 == 0 else 0
    y, st = ops.gemm_geglu(a.cuda(), w.cuda(), bias.cuda(), stash_grad=bool(stash_grad), stash_rows=rows)
    pre = a.float() @ w.float().t() + bias
    close_bf16(f"gemm geglu y {M}x{N}x{K}", y, pre[:, 0::2] * F.gelu(pre[:, 1::2]), ulps=2.0)
    if stash_grad:
        fa, fb = _geglu_factors(pre)
        ref = torch.stack([fa, fb], -1).reshape(M, N)
    else:
        ref = pre
    n = rows if rows else M
    close_bf16(f"gemm geglu stash (grad form {stash_grad}) {M}x{N}x{K}", st[:n], ref[:n], ulps=2.0)
    if rows:
        assert float(st[rows:].float().abs().max()) == 0.0          # teacher rows: never written


@pytest.mark.parametrize("B,H,W,C,Cout", [(2, 32, 32, 320, 4), (1, 12, 20, 128, 3), (3, 8, 24, 512, 8), (1, 40, 40, 64, 4),
                                          (2, 16, 16, 72, 4)])
def test_conv_out_shapes(ops, B, H, W, C, Cout):
    """conv_out on the matrix cores (weights as hi + lo bf16 halves): UNet eps head (320 -> 4), VAE decoder head (128 -> 3),
    VAE encoder moments (512 -> 8), pixel counts that are not multiples of the 512-pixel workgroup or the 16-pixel strip;
    Cin = 72 takes the direct kernel (Cin % 32 != 0)."""
    g = torch.Generator().manual_seed(B * 100 + C)
    h = (torch.randn(B, C, H, W, generator=g) * 1.5).to(BF)
    w_out = torch.randn(Cout, C, 3, 3, generator=g) * (1.0 / (3.0 * C ** 0.5))
    b_out = torch.randn(Cout, generator=g)
    wp = ops.pack_conv_out(w_out.cuda())
    got = ops.conv_out(_nhwc(h).cuda(), wp, b_out.cuda())
    close_f32(f"conv_out {C}->{Cout} {B}x{H}x{W}", got, F.conv2d(h.float(), w_out, b_out, padding=1))


@pytest.mark.parametrize("B,H,W,Cs,N", [(2, 32, 32, 4, 320), (1, 12, 20, 3, 128), (1, 16, 24, 4, 512), (2, 9, 10, 4, 64)])
def test_conv_in_and_conv_out_dgrad_shapes(ops, B, H, W, Cs, N):
    """The 4-pixels-per-thread kernel behind conv_in and the conv_out dgrad (W % 4 == 0) and the one-pixel fallback (W = 10):
    UNet conv_in (4 -> 320), VAE encoder conv_in (3 -> 128), VAE decoder conv_in (4 -> 512, 72 KB of weights in LDS)."""
    g = torch.Generator().manual_seed(B * 1000 + N + W)
    x = torch.randn(B, Cs, H, W, generator=g)
    w_in, b_in = torch.randn(N, Cs, 3, 3, generator=g) * 0.2, torch.randn(N, generator=g)
    y = ops.conv_in(x.cuda(), w_in.cuda(), b_in.cuda())
    close_bf16(f"conv_in {Cs}->{N} {B}x{H}x{W}", y, _nhwc(F.conv2d(x, w_in, b_in, padding=1)))
    w_out = torch.randn(Cs, N, 3, 3, generator=g) * 0.05
    wp = ops.pack_conv_out(w_out.cuda())
    hh = torch.zeros(B, N, H, W, requires_grad=True)
    dy = torch.randn(B, Cs, H, W, generator=g)
    F.conv2d(hh, w_out, None, padding=1).backward(dy)
    close_bf16(f"conv_out dgrad {Cs}->{N} {B}x{H}x{W}", ops.conv_out_dgrad(dy.cuda(), wp, N), _nhwc(hh.grad))


# ------------------------------------------------------------------------------------ norms
# one-kernel (slab in registers) forms: 8-element vectors (40 / 80 channels per group), 4-element vectors (20 / 60), group
# pairs (10 / 30), a ragged last wave (HW = 120 * 8), every pixels-per-thread count; three-launch general path: 16384 x 320,
# 1024 x 960 (slab too large), HW = 16 / 36 (not a multiple of 8)
@pytest.mark.parametrize("B,HW,C,silu", [(2, 256, 320, True), (2, 64, 64, False), (1, 1024, 960, True),
                                         (3, 16, 2560, True), (2, 4096, 640, False), (2, 1024, 1280, True),
                                         (2, 1024, 2560, True), (1, 1024, 1920, False), (1, 4096, 1280, True),
                                         (2, 960, 1280, True), (1, 16384, 320, True), (2, 36, 640, True),
                                         (2, 256, 960, False), (1, 4096, 640, True)])
def test_groupnorm(ops, B, HW, C, silu):
    x = (bfr(B, HW, C, seed=1).float() * 1.5 + 0.7).to(BF)
    g = torch.Generator().manual_seed(2)
    gamma, beta = 1 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    xr = x.float().requires_grad_(True)
    z = F.group_norm(xr.permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1)
    yr = F.silu(z) if silu else z
    y, stats = ops.groupnorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 32, 1e-5, silu)
    close_bf16(f"groupnorm fwd C{C} silu{silu}", y, yr)
    zr = z.detach().reshape(B, HW, 32, C // 32)
    mean_ref = x.float().reshape(B, HW, 32, C // 32).mean(dim=(1, 3))
    var_ref = x.float().reshape(B, HW, 32, C // 32).var(dim=(1, 3), unbiased=False)
    st = stats.cpu().reshape(B, 32, 2)
    assert torch.allclose(st[..., 0], mean_ref, rtol=1e-4, atol=1e-5), "groupnorm: saved mean"
    assert torch.allclose(st[..., 1], (var_ref + 1e-5).rsqrt(), rtol=1e-4, atol=1e-5), "groupnorm: saved rstd"
    dy = bfr(B, HW, C, seed=3)
    yr.backward(dy.float())
    dx = ops.groupnorm_bwd(x.cuda(), dy.cuda(), gamma.cuda(), beta.cuda(), stats, 32, silu)
    close_bf16(f"groupnorm bwd C{C} silu{silu}", dx, xr.grad, ulps=2.0)
    # accumulate form (the resnet input's gradient already holds the shortcut's share)
    prev = bfr(B, HW, C, seed=4)
    acc = prev.cuda().clone()
    ops.groupnorm_bwd(x.cuda(), dy.cuda(), gamma.cuda(), beta.cuda(), stats, 32, silu, accum_into=acc)
    close_bf16(f"groupnorm bwd accumulate C{C}", acc, xr.grad + prev.float(), ulps=2.0)


@pytest.mark.parametrize("R,C", [(64, 640), (308, 1024), (100, 1280), (16, 128), (7, 2048)])
def test_layernorm(ops, R, C):
    x = (bfr(R, C, seed=1).float() * 2 - 0.5).to(BF)
    g = torch.Generator().manual_seed(2)
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_(True), (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    xr = x.float().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), gamma, beta, 1e-5)
    y, stats = ops.layernorm_fwd(x.cuda(), gamma.detach().cuda(), beta.detach().cuda())
    close_bf16(f"layernorm fwd {R}x{C}", y, yr)
    dy = bfr(R, C, seed=3)
    yr.backward(dy.float())
    dx, dg, db = ops.layernorm_bwd(x.cuda(), dy.cuda(), gamma.detach().cuda(), stats, want_param_grads=True)
    close_bf16(f"layernorm bwd {R}x{C}", dx, xr.grad, ulps=2.0)
    close_f32("layernorm dgamma", dg, gamma.grad, rtol=2e-3, atol=2e-3)
    close_f32("layernorm dbeta", db, beta.grad, rtol=2e-3, atol=2e-3)


# ------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, H):
    B, Sq, C = q.shape
    qh = q.view(B, Sq, H, 64).transpose(1, 2)
    kh = k.view(B, -1, H, 64).transpose(1, 2)
    vh = v.view(B, -1, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * 0.125
    o = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Sq, C)
    return o, torch.logsumexp(s, -1)


ALPHA = 0.125 * math.log2(math.e)        # softmax scale x log2(e): what a prescaled Q carries (head_dim 64)


@pytest.mark.parametrize("use_tr,prescaled", [(1, True), (1, False), (0, False)])     # (the scalar-read debug variant: plain Q only)
@pytest.mark.parametrize("B,H,Sq,Skv", [(2, 2, 256, 256), (1, 3, 128, 77), (2, 2, 16, 16), (1, 2, 1024, 200),
                                        (1, 1, 64, 7), (2, 2, 1024, 77), (1, 2, 576, 77),
                                        # one-pass cross-attention backward (<= 128 keys): ragged query counts of the aspect-ratio
                                        # buckets (14x26, 28x52 tokens), 1..4 key blocks, several query splits
                                        (1, 2, 364, 77), (2, 3, 1456, 77), (1, 2, 640, 128), (1, 2, 512, 100), (1, 2, 260, 33)])
def test_attention_fwd_bwd(ops, use_tr, B, H, Sq, Skv, prescaled):
    """prescaled (the product path): the Q handed over already holds q * scale * log2(e) -- exact inputs for the fp32
    reference, whose q is Q' / (scale log2 e); dQ comes back as the gradient w.r.t. that unscaled q.  Not prescaled: the
    kernels scale their resident operand themselves (one more bf16 rounding, same tolerances at these logit sizes)."""
    from pea_diffusion_amd._lib import lib
    lib().pea_debug_set_attn_tr(use_tr)
    try:
        q, k, v = bfr(B, Sq, H * 64, seed=1), bfr(B, Skv, H * 64, seed=2), bfr(B, Skv, H * 64, seed=3)
        if prescaled:
            q = (q.float() * ALPHA).to(BF)                 # any bf16 tensor is a valid Q'; this one has the usual score sizes
        qr = (q.float() / ALPHA if prescaled else q.float()).requires_grad_(True)
        kr, vr = [t.float().requires_grad_(True) for t in (k, v)]
        oref, lref = _attn_ref(qr, kr, vr, H)
        o, lse = ops.attention_fwd(q.cuda(), k.cuda(), v.cuda(), H, q_prescaled=prescaled)
        tag = f"attn tr{use_tr} pre{int(prescaled)} B{B} H{H} Sq{Sq} Skv{Skv}"
        close_bf16(tag + " O", o, oref, ulps=2.0)
        close_f32(tag + " lse", lse, lref, rtol=1e-3, atol=2e-3)
        do = bfr(B, Sq, H * 64, seed=4)
        oref.backward(do.float())
        dq, dk, dv = ops.attention_bwd(q.cuda(), k.cuda(), v.cuda(), o, do.cuda(), lse, H, q_prescaled=prescaled)
        close_bf16(tag + " dQ", dq, qr.grad, ulps=4.0)
        close_bf16(tag + " dK", dk, kr.grad, ulps=4.0)
        close_bf16(tag + " dV", dv, vr.grad, ulps=4.0)
    finally:
        lib().pea_debug_set_attn_tr(1)


@pytest.mark.parametrize("prescaled", [True, False])
@pytest.mark.parametrize("B,H,Sq,Skv", [(2, 3, 1024, 77), (1, 2, 4096, 77), (2, 2, 1000, 52), (1, 2, 2048, 64), (1, 1, 192, 77),
                                        (1, 2, 576, 96), (3, 2, 640, 40)])
def test_cross_attention_bwd_specialised_waves(ops, B, H, Sq, Skv, prescaled):
    """The round-6 cross-attention backward kernels -- xattn_bwd3_kernel (<= 80 keys: S / dP / P / dS evaluated ONCE by the key
    waves, dS handed to the dQ wave through an LDS image: 5 products) and xattn_bwd2_kernel (<= 96 keys: 64-query units,
    double-buffered stage, waves 0-1 = dQ role, waves 2-3 = dK/dV role, 7 products) -- against the fp32 reference AND against the
    round-3 one-pass kernel (pea_debug_set_xattn_bwd_v2(0)) on the same inputs: 2 and 3 key blocks, whole and ragged
    query counts (1000 = 15 units + 40 rows), 1..16 query splits, the accumulate-into form of dQ / dK / dV (+=, as a tensor
    with a second consumer gets it), plain and prescaled Q.  Bit-reproducible: two launches agree exactly."""
    import ctypes
    from pea_diffusion_amd._lib import check, lib, ptr, stream_ptr
    L = lib()
    q, k, v = bfr(B, Sq, H * 64, seed=11), bfr(B, Skv, H * 64, seed=12), bfr(B, Skv, H * 64, seed=13)
    if prescaled:
        q = (q.float() * ALPHA).to(BF)
    qr = (q.float() / ALPHA if prescaled else q.float()).requires_grad_(True)
    kr, vr = [t.float().requires_grad_(True) for t in (k, v)]
    oref, _ = _attn_ref(qr, kr, vr, H)
    do = bfr(B, Sq, H * 64, seed=14)
    oref.backward(do.float())
    qc, kc, vc, doc = q.cuda(), k.cuda(), v.cuda(), do.cuda()
    o, lse = ops.attention_fwd(qc, kc, vc, H, q_prescaled=prescaled)
    tag = f"xattn-bwd2 pre{int(prescaled)} B{B} H{H} Sq{Sq} Skv{Skv}"
    try:
        got = {}
        for ver in (3, 0, 2, 3):           # newest that applies (five-product kernel up to 80 keys) / round-3 kernel / specialised waves
            L.pea_debug_set_xattn_bwd_v2(ver)
            g = ops.attention_bwd(qc, kc, vc, o, doc, lse, H, q_prescaled=prescaled)
            torch.cuda.synchronize()
            if ver in got:
                assert all(torch.equal(a, b_) for a, b_ in zip(got[ver], g)), tag + ": two launches differ"
            got[ver] = g
        for ver in (3, 2):
            for name, a, r in zip(("dQ", "dK", "dV"), got[ver], (qr.grad, kr.grad, vr.grad)):
                close_bf16(f"{tag} v{ver} {name}", a, r, ulps=4.0)
            for name, a, b_ in zip(("dQ", "dK", "dV"), got[ver], got[0]):       # same algorithm, other summation order
                close_bf16(f"{tag} {name} v{ver}-vs-v1", a, b_.float().cpu(), ulps=2.0)
        # accumulate-into: dX = X0 + gradient (direct bf16 form when there is no query split, i.e. without the scratch)
        L.pea_debug_set_xattn_bwd_v2(3)
        C = H * 64
        x0 = [bfr(*t.shape, seed=20 + i).cuda() for i, t in enumerate((q, k, v))]
        acc = [t.clone() for t in x0]
        delta = torch.empty(2, B, H, Sq, device="cuda", dtype=torch.float32)
        fn = L.pea_op_attention_bwd_prescaled if prescaled else L.pea_op_attention_bwd
        check(fn(ptr(qc), C, ptr(kc), C, ptr(vc), C, ptr(o), C, ptr(doc), C, ptr(lse), ptr(delta), ptr(acc[0]), C, ptr(acc[1]), C,
                 ptr(acc[2]), C, B, H, Sq, Skv, 0.125, 1, 1, 1, None, stream_ptr()))
        torch.cuda.synchronize()
        for name, a, x, r in zip(("dQ", "dK", "dV"), acc, x0, (qr.grad, kr.grad, vr.grad)):
            close_bf16(f"{tag} {name} accumulate", a, x.float().cpu() + r, ulps=4.0)
    finally:
        L.pea_debug_set_xattn_bwd_v2(3)


@pytest.mark.parametrize("prescaled", [True, False])
@pytest.mark.parametrize("spike,tile_key", [(4.0, 200), (4.0, 70), (0.6, 200), (1.0, 1000)])
def test_attention_softmax_spike(ops, prescaled, spike, tile_key):
    """Forces (spike 4: +46 in log2 units) or just avoids (0.6: below ATTN_MOVE_THR) a late jump of one query's score offset --
    the forward's rescale branch runs for ONE row of a wave in a late tile while the other rows keep their offset -- and
    checks forward and backward against fp32 on the full tensors.  With a plain Q the kernels round Q * scale * log2(e) to
    bf16 themselves, a relative 2^-9 on every logit: at a logit of 32 that is 0.02 in lse, hence the wider lse tolerance of
    that mode (the product path hands Q over prescaled by the projection's epilogue: exact inputs, tight tolerance)."""
    B, H, Sq, Skv = 1, 1, 64, 1024 if tile_key >= 256 else 256
    q, k, v = bfr(B, Sq, 64, seed=1), bfr(B, Skv, 64, seed=2), bfr(B, Skv, 64, seed=3)
    k[0, tile_key] = (q[0, 5].float() * spike).to(BF)
    if prescaled:
        q = (q.float() * ALPHA).to(BF)
    qr = (q.float() / ALPHA if prescaled else q.float()).requires_grad_(True)
    kr, vr = [t.float().requires_grad_(True) for t in (k, v)]
    oref, lref = _attn_ref(qr, kr, vr, H)
    o, lse = ops.attention_fwd(q.cuda(), k.cuda(), v.cuda(), H, q_prescaled=prescaled)
    close_bf16("attn spike O", o, oref, ulps=2.0)
    close_f32("attn spike lse", lse, lref, rtol=1e-3, atol=2e-3 if prescaled else 2.5e-2)
    if not prescaled and spike > 1.0:
        return          # (a plain Q with logits of 30+: the kernels' own rounding of the scaled operand dominates the gradients)
    do = bfr(B, Sq, 64, seed=4)
    oref.backward(do.float())
    dq, dk, dv = ops.attention_bwd(q.cuda(), k.cuda(), v.cuda(), o, do.cuda(), lse, H, q_prescaled=prescaled)
    # a query whose softmax sits on one key (P > 0.9 there) has dS = P (dP - delta) cancelling to rounding level: its dQ row is
    # ill-conditioned in ANY bf16 implementation, so those rows (the spiked one and whichever others the spiked key captures)
    # are held to an absolute bound and every other row to the usual one; dK / dV sum over all queries and are checked whole
    with torch.no_grad():
        pmax = torch.softmax(qr[0] @ kr[0].T * 0.125, -1).max(-1).values
    easy = pmax <= 0.9
    assert int(easy.sum()) >= Sq // 2
    close_bf16("attn spike dQ", dq.cpu()[:, easy], qr.grad[:, easy], ulps=4.0)
    if (~easy).any():
        rms = qr.grad[:, easy].pow(2).mean().sqrt().item()
        worst = (dq.cpu().float()[:, ~easy] - qr.grad[:, ~easy]).abs().max().item()
        print(f"[attn spike dQ, {int((~easy).sum())} concentrated rows] max_abs={worst:.3e} rms(other rows)={rms:.3e}")
        assert worst < 0.25 * rms
    keep = torch.ones(Skv, dtype=torch.bool)
    if spike > 1.0:
        keep[tile_key] = False              # the capturing key's own dK row: the same cancellation, summed over its queries
        rms_k = kr.grad[:, keep].pow(2).mean().sqrt().item()
        worst = (dk.cpu().float()[:, tile_key] - kr.grad[:, tile_key]).abs().max().item()
        print(f"[attn spike dK, spiked key] max_abs={worst:.3e} rms(other keys)={rms_k:.3e}")
        assert worst < 0.5 * rms_k
    close_bf16("attn spike dK", dk.cpu()[:, keep], kr.grad[:, keep], ulps=4.0)
    close_bf16("attn spike dV", dv, vr.grad, ulps=4.0)


def test_gemm_qscale_columns(ops):
    """the fused Q|K|V projection's epilogue: columns [0, C) (the Q block) leave multiplied by scale * log2(e) -- one rounding,
    from the fp32 accumulator -- the K and V blocks unchanged; all tile forms the shape rule picks for these sizes"""
    for M, C, K, with_bias in [(8192, 1280, 1280, False), (4096, 640, 640, True), (616, 768, 768, True), (300, 320, 320, False)]:
        a, w = bfr(M, K, seed=M), bfr(3 * C, K, seed=C, scale=K ** -0.5)
        bias = torch.randn(3 * C) * 0.1 if with_bias else None
        got = ops.gemm_qscale(a.cuda(), w.cuda(), bias.cuda() if with_bias else None, qscale_cols=C, qscale=ALPHA)
        ref = a.float() @ w.float().t() + (bias if with_bias else 0.0)
        ref[:, :C] *= ALPHA
        close_bf16(f"gemm qscale {M}x{3 * C}x{K}", got, ref)


def test_geglu(ops):
    hg = bfr(96, 2 * 320, seed=1)
    hr = hg.float().requires_grad_(True)
    h, gate = hr.chunk(2, -1)
    yr = h * F.gelu(gate)
    close_bf16("geglu fwd", ops.geglu_fwd(hg.cuda()), yr)
    dy = bfr(96, 320, seed=2)
    yr.backward(dy.float())
    close_bf16("geglu bwd", ops.geglu_bwd(hg.cuda(), dy.cuda()), hr.grad, ulps=2.0)


def test_timestep_embed_and_add_noise(ops):
    from oracle.step_ref import add_noise, ddpm_alphas_cumprod
    from oracle.unet_ref import timestep_embedding
    t = torch.tensor([0., 10., 250., 999., 1024.])
    close_bf16("timestep_embed 320", ops.timestep_embed(t.cuda(), 320), timestep_embedding(t, 320), ulps=2.0)
    close_bf16("timestep_embed 256", ops.timestep_embed(t.cuda(), 256), timestep_embedding(t, 256), ulps=2.0)
    g = torch.Generator().manual_seed(0)
    x0, eps = torch.randn(4, 4, 16, 16, generator=g), torch.randn(4, 4, 16, 16, generator=g)
    ts = torch.tensor([10, 250, 500, 999])
    xt = ops.add_noise(x0.cuda(), eps.cuda(), ts.cuda(), ddpm_alphas_cumprod().cuda())
    close_f32("add_noise", xt, add_noise(x0, eps, ts), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("zh", [[1, 0, 0, 1], [0, 0, 0, 0], [1, 1, 1, 1]])
def test_kd_loss(ops, zh):
    from oracle.step_ref import kd_losses
    B = 4
    shapes = [(B, 64, 8, 8), (B, 128, 4, 4), (B, 128, 4, 4), (B, 128, 8, 8), (B, 128, 16, 16), (B, 64, 16, 16),
              (B, 320, 24, 24)]
    fs = [bfr(*s, seed=10 + i) for i, s in enumerate(shapes)]
    ft = [bfr(*s, seed=30 + i) for i, s in enumerate(shapes)]
    g = torch.Generator().manual_seed(5)
    es, e, et = [torch.randn(B, 4, 16, 16, generator=g) for _ in range(3)]
    z = torch.tensor(zh)
    fsr = [t.float().requires_grad_(True) for t in fs]
    esr = es.clone().requires_grad_(True)
    tot, l0, l1, l2 = kd_losses(esr, e, et, fsr, [t.float() for t in ft], z)
    tot.backward()
    losses, dt, de = ops.kd_loss([t.cuda() for t in fs], [t.cuda() for t in ft], es.cuda(), e.cuda(), et.cuda(), z.cuda())
    close_f32("kd losses", losses, torch.stack([tot, l0, l1, l2]).detach(), rtol=1e-3, atol=1e-5)
    close_f32("kd deps", de, esr.grad, rtol=1e-3, atol=1e-9)
    for i, (d, r) in enumerate(zip(dt, fsr)):
        ref = r.grad if r.grad is not None else torch.zeros_like(r)
        close_bf16(f"kd dtap{i}", d, ref, ulps=2.0)


def test_adamw(ops):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(1000, generator=g)
    p = w.clone().requires_grad_(True)
    opt = torch.optim.AdamW([p], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    wd, m, v = w.cuda(), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g)
        p.grad = gr.clone()
        opt.step()
        ops.adamw_(wd, gr.cuda(), m, v, 1e-2, step, weight_decay=0.01)
    close_f32("adamw", wd, p.detach(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("d,H,Sq,Skv", [(40, 8, 256, 256), (80, 4, 128, 77), (160, 2, 64, 64), (128, 2, 192, 200), (16, 8, 64, 20)])
def test_attention_padded_heads(ops, d, H, Sq, Skv):
    """SD1.5 head widths (40 / 80 / 160) are stored zero-padded to 64 / 128 / 192 columns; scale = d^-0.5"""
    B = 2
    dp = (d + 63) // 64 * 64
    def padded(x):                       # [B,S,H,d] -> [B,S,H*dp] with zero padding
        out = torch.zeros(B, x.shape[1], H, dp, dtype=BF)
        out[..., :d] = x
        return out.reshape(B, x.shape[1], H * dp)
    q, k, v, do = [bfr(B, s_, H, d, seed=i) for i, s_ in enumerate((Sq, Skv, Skv, Sq))]
    qr, kr, vr = [t_.float().requires_grad_(True) for t_ in (q, k, v)]
    s = torch.einsum("bqhd,bkhd->bhqk", qr, kr) * d ** -0.5
    oref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), vr)
    o, lse = ops.attention_fwd(padded(q).cuda(), padded(k).cuda(), padded(v).cuda(), H, scale=d ** -0.5)
    og = o.float().cpu().view(B, Sq, H, dp)
    close_bf16(f"attn d{d} O", og[..., :d], oref, ulps=2.0)
    assert d == dp or og[..., d:].abs().max() == 0
    close_f32(f"attn d{d} lse", lse, torch.logsumexp(s, -1), rtol=1e-3, atol=2e-3)
    oref.backward(do.float())
    dq, dk, dv = ops.attention_bwd(padded(q).cuda(), padded(k).cuda(), padded(v).cuda(), o, padded(do).cuda(), lse, H,
                                   scale=d ** -0.5)
    for name, g, r, S in (("dQ", dq, qr.grad, Sq), ("dK", dk, kr.grad, Skv), ("dV", dv, vr.grad, Skv)):
        gg = g.float().cpu().view(B, S, H, dp)
        close_bf16(f"attn d{d} {name}", gg[..., :d], r, ulps=4.0)
        assert d == dp or gg[..., d:].abs().max() == 0


@pytest.mark.parametrize("B,H,S,causal,padded", [(2, 2, 77, True, False), (3, 4, 52, False, True), (2, 3, 200, True, True),
                                                 (1, 2, 130, True, False), (2, 2, 64, False, True)])
def test_attention_text_masks(ops, B, H, S, causal, padded):
    """causal and key-padding masks of the text encoders (separate instance of the forward kernel) vs torch SDPA"""
    C = H * 64
    q, k, v = bfr(B, S, C, seed=1), bfr(B, S, C, seed=2), bfr(B, S, C, seed=3)
    g = torch.Generator().manual_seed(4)
    lens = torch.randint(1, S + 1, (B,), generator=g) if padded else torch.full((B,), S)
    mask = torch.zeros(B, 1, S, S)
    if causal:
        mask = mask + torch.full((S, S), float("-inf")).triu(1)
    for b in range(B):
        mask[b, :, :, int(lens[b]):] = float("-inf")
    sp = lambda t: t.float().view(B, S, H, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=mask).transpose(1, 2).reshape(B, S, C)
    o = ops.attention_fwd_masked(q.cuda(), k.cuda(), v.cuda(), H, causal=causal,
                                 kv_len=lens.to(torch.int32).cuda() if padded else None)
    # with causal + padding a query row can lie beyond the valid keys only through padding; every row keeps key 0
    close_bf16(f"attn text B{B} H{H} S{S} causal={causal} padded={padded}", o, ref, ulps=2.0)   # as the other attention outputs


def test_gemm_tile_rules_on_a_smaller_device(ops):
    """The launcher's tile rules and persistent grids take the CU count from the device (a CPX partition of an MI355X has 32 CUs)
    or from PEA_CU_LIMIT: the plain / epilogue / GEGLU GEMM tests again in a child process limited to 40 CUs."""
    import subprocess
    env = dict(os.environ, PEA_CU_LIMIT="40")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_gemm_plain or test_gemm_epilogues or test_gemm_geglu_forward_and_stash_forms or test_conv3x3"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
