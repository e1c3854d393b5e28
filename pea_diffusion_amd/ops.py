"""Operator-level Python bindings over the C ABI (include/pea_hip.h, `pea_op_*`).

Thin: torch tensors provide device memory and the current stream, nothing else.  Every function
takes/returns CUDA(HIP) tensors; bf16 activations are token-major ("NHWC").
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

from ._lib import check, lib, ptr, stream_ptr

BF = torch.bfloat16


def _dev(t):
    return t.device


def gemm(a, w, bias=None, rowvec=None, rows_per_batch=1, act=0, res=None, want_preact=False, out_f32=False,
         alpha=1.0, out=None, accum_f32=False):
    """act(alpha * a @ w.T + bias + rowvec[m // rows_per_batch]) + res ; a [M,K] bf16, w [N,K] bf16."""
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32 if out_f32 else BF)
    pre = torch.empty(M, N, device=a.device, dtype=BF) if want_preact else None
    check(lib().pea_op_gemm(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(out), out.stride(0), M, N, K, alpha,
                            ptr(bias), ptr(rowvec), rowvec.stride(0) if rowvec is not None else 0, rows_per_batch,
                            act, ptr(pre), N, ptr(res), res.stride(0) if res is not None else 0,
                            int(out.dtype == torch.float32), int(accum_f32), stream_ptr()))
    return (out, pre) if want_preact else out


def gemm_geglu(a, w, bias=None, want_stash=True, stash_grad=True, stash_rows=0):
    """FF projection with GEGLU in its epilogue (pea_op_gemm_geglu): w [N,K] rows interleaved (h_i, gate_i);
    returns (h * gelu(gate) [M,N/2], stash [M,N] or None).  stash_grad: the stash holds (gelu(gate), h * gelu'(gate))."""
    M, K = a.shape
    N = w.shape[0]
    y = torch.empty(M, N // 2, device=a.device, dtype=BF)
    st = torch.zeros(M, N, device=a.device, dtype=BF) if want_stash else None
    check(lib().pea_op_gemm_geglu(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(y), ptr(st), M, N, K,
                                  int(stash_grad), stash_rows, stream_ptr()))
    return y, st


def gemm_geglu_bwd(a, w, pre, form=0):
    """dgrad GEMM of the FF output projection with the GEGLU backward in its epilogue (pea_op_gemm_geglu_bwd):
    dy = a @ w.T ([M,N], never stored); pre [M,2N] interleaved (h, gate) [form 0] or (gelu(gate), h gelu'(gate)) [form 1]
    -> d(pre) [M,2N] interleaved (dh, dgate)."""
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, 2 * N, device=a.device, dtype=BF)
    check(lib().pea_op_gemm_geglu_bwd(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(pre), pre.stride(0), ptr(out), out.stride(0),
                                      M, N, K, form, stream_ptr()))
    return out


def ln_linear(x, gamma, beta, w, bias=None, eps=1e-5, geglu=False):
    """LayerNorm folded into its consuming Linear (pea_op_ln_linear): x [M,K] bf16, w [N,K] bf16 -> LN(x) @ w.T + bias.
    geglu=True: w rows interleaved (h_i, gate_i); returns (h * gelu(gate) [M,N/2], pre-activation [M,N])."""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=BF)
    gy = torch.empty(M, N // 2, device=x.device, dtype=BF) if geglu else None
    wf = torch.empty(N, K, device=x.device, dtype=BF)
    sv = torch.empty(N, device=x.device, dtype=torch.float32)
    tv = torch.empty(N, device=x.device, dtype=torch.float32)
    st = torch.empty(M, 2, device=x.device, dtype=torch.float32)
    check(lib().pea_op_ln_linear(ptr(x), ptr(gamma), ptr(beta), ptr(w), ptr(bias), ptr(y), ptr(gy), M, N, K, eps, ptr(wf),
                                 ptr(sv), ptr(tv), ptr(st), stream_ptr()))
    return (gy, y) if geglu else y


def pack_conv(w_fp32, dgrad=False):
    Co, Ci = w_fp32.shape[:2]
    out = torch.empty((Ci, 9 * Co) if dgrad else (Co, 9 * Ci), device=w_fp32.device, dtype=BF)
    check(lib().pea_op_pack_conv(ptr(w_fp32.contiguous()), ptr(out), Co, Ci, int(dgrad), stream_ptr()))
    return out


def conv3x3(x, w_packed, bias=None, stride=1, upsample2x=False, transposed2=False, rowvec=None, res=None):
    """x [B,H,W,Cin] bf16 NHWC; w_packed [Cout, 9*Cin] -> [B,Ho,Wo,Cout]."""
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    sh = 1 if (upsample2x or transposed2) else 0
    Hv, Wv = H << sh, W << sh
    Ho, Wo = ((Hv + 1) // 2, (Wv + 1) // 2) if stride == 2 else (Hv, Wv)
    y = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=BF)
    check(lib().pea_op_conv3x3(ptr(x), ptr(w_packed), ptr(y), B, H, W, Cin, Cout, stride, int(upsample2x),
                               int(transposed2), ptr(bias), ptr(rowvec),
                               rowvec.stride(0) if rowvec is not None else 0, ptr(res), stream_ptr()))
    return y


def pack_conv_subpixel(w_fp32, dgrad=False):
    """[Co,Ci,3,3] fp32 -> the sub-pixel form of conv3x3(nearest_2x(.)): [4,Co,4*Ci] (forward) or [Ci,16*Co] (data gradient)"""
    Co, Ci = w_fp32.shape[:2]
    out = torch.empty((Ci, 16 * Co) if dgrad else (4, Co, 4 * Ci), device=w_fp32.device, dtype=BF)
    check(lib().pea_op_pack_conv_subpixel(ptr(w_fp32.contiguous()), ptr(out), Co, Ci, int(dgrad), stream_ptr()))
    return out


def upconv_subpixel(x, w_packed, bias=None):
    """x [B,H,W,Cin] -> conv3x3(nearest_2x(x)) stored depth-to-space [B,H,W,4,Cout] (block (y&1)*2 + (x&1))"""
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[1]
    y = torch.empty(B, H, W, 4, Cout, device=x.device, dtype=BF)
    check(lib().pea_op_upconv_subpixel(ptr(x), ptr(w_packed), ptr(y), B, H, W, Cin, Cout, ptr(bias), stream_ptr()))
    return y


def upconv_subpixel_dgrad(dy_d2s, wt_packed, res=None):
    """dy [B,H,W,4,Cout] (depth-to-space) -> dx [B,H,W,Cin] (+ res)"""
    B, H, W, _, Cout = dy_d2s.shape
    Cin = wt_packed.shape[0]
    dx = torch.empty(B, H, W, Cin, device=dy_d2s.device, dtype=BF)
    check(lib().pea_op_upconv_subpixel_dgrad(ptr(dy_d2s), ptr(wt_packed), ptr(dx), B, H, W, Cin, Cout, ptr(res), stream_ptr()))
    return dx


def d2s_to_nhwc(y):
    """[B,H,W,4,C] depth-to-space -> [B,2H,2W,C]"""
    B, H, W, _, C = y.shape
    return y.view(B, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, C)


def nhwc_to_d2s(y):
    """[B,2H,2W,C] -> [B,H,W,4,C]"""
    B, H2, W2, C = y.shape
    return y.view(B, H2 // 2, 2, W2 // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H2 // 2, W2 // 2, 4, C).contiguous()


def concat2(a, b, d2s_hw=None):
    """rows-wise channel concat; d2s_hw = (H, W): `a` is stored depth-to-space at that full resolution"""
    C1, C2 = a.shape[-1], b.shape[-1]
    rows = b.numel() // C2
    y = torch.empty(rows, C1 + C2, device=a.device, dtype=BF)
    H, W = d2s_hw if d2s_hw else (0, 0)
    check(lib().pea_op_concat2(ptr(a), C1, ptr(b), C2, ptr(y), rows, H, W, stream_ptr()))
    return y


def split2(dy, C1, C2, da=None, db=None, accum_a=False, accum_b=False, d2s_hw=None):
    rows = dy.numel() // (C1 + C2)
    H, W = d2s_hw if d2s_hw else (0, 0)
    check(lib().pea_op_split2(ptr(dy), C1, C2, ptr(da), int(accum_a), ptr(db), int(accum_b), rows, H, W, stream_ptr()))


def conv_in(x_nchw, w, bias):
    B, Cin, H, W = x_nchw.shape
    Cout = w.shape[0]
    y = torch.empty(B, H, W, Cout, device=x_nchw.device, dtype=BF)
    check(lib().pea_op_conv_in(ptr(x_nchw), ptr(w.contiguous()), ptr(bias), ptr(y), B, Cin, H, W, Cout, stream_ptr()))
    return y


def pack_conv_out(w):
    Co, Ci = w.shape[:2]
    out = torch.empty(Co, 3, 3, Ci, device=w.device, dtype=torch.float32)
    check(lib().pea_op_pack_conv_out(ptr(w.contiguous()), ptr(out), Co, Ci, stream_ptr()))
    return out


def conv_out(x_nhwc, w_packed, bias):
    B, H, W, Cin = x_nhwc.shape
    Cout = w_packed.shape[0]
    y = torch.empty(B, Cout, H, W, device=x_nhwc.device, dtype=torch.float32)
    check(lib().pea_op_conv_out(ptr(x_nhwc), ptr(w_packed), ptr(bias), ptr(y), B, Cin, H, W, Cout, stream_ptr()))
    return y


def conv_out_dgrad(dy_nchw, w_packed, Cin):
    B, Cout, H, W = dy_nchw.shape
    dx = torch.empty(B, H, W, Cin, device=dy_nchw.device, dtype=BF)
    check(lib().pea_op_conv_out_dgrad(ptr(dy_nchw), ptr(w_packed), ptr(dx), B, Cin, H, W, Cout, stream_ptr()))
    return dx


def groupnorm_fwd(x, gamma, beta, groups=32, eps=1e-5, silu=False):
    B, HW, C = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(B, groups, 2, device=x.device, dtype=torch.float32)
    scratch = torch.empty(lib().pea_op_groupnorm_scratch_bytes(B, HW, C, groups), device=x.device, dtype=torch.uint8)
    check(lib().pea_op_groupnorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(stats), ptr(scratch), B, HW, C, groups,
                                     eps, int(silu), stream_ptr()))
    return y, stats


def groupnorm_bwd(x, dy, gamma, beta, stats, groups=32, silu=False, accum_into=None):
    B, HW, C = x.shape
    dx = accum_into if accum_into is not None else torch.empty_like(x)
    scratch = torch.empty(lib().pea_op_groupnorm_scratch_bytes(B, HW, C, groups), device=x.device, dtype=torch.uint8)
    check(lib().pea_op_groupnorm_bwd(ptr(x), ptr(dy), ptr(gamma), ptr(beta), ptr(stats), ptr(dx), ptr(scratch), B, HW,
                                     C, groups, int(silu), int(accum_into is not None), stream_ptr()))
    return dx


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    R, C = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(R, 2, device=x.device, dtype=torch.float32)
    check(lib().pea_op_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(stats), R, C, eps, stream_ptr()))
    return y, stats


def layernorm_bwd(x, dy, gamma, stats, want_param_grads=False, accum_into=None):
    R, C = x.shape
    dx = accum_into if accum_into is not None else torch.empty_like(x)
    dg = torch.zeros(C, device=x.device, dtype=torch.float32) if want_param_grads else None
    db = torch.zeros(C, device=x.device, dtype=torch.float32) if want_param_grads else None
    check(lib().pea_op_layernorm_bwd(ptr(x), ptr(dy), ptr(gamma), ptr(stats), ptr(dx), ptr(dg), ptr(db), R, C,
                                     int(accum_into is not None), stream_ptr()))
    return (dx, dg, db) if want_param_grads else dx


def gemm_qscale(a, w, bias=None, qscale_cols=0, qscale=1.0):
    """a @ w.T + bias with the first qscale_cols output columns multiplied by qscale (the Q block of a fused Q|K|V projection)"""
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, device=a.device, dtype=BF)
    check(lib().pea_op_gemm_qscale(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(out), out.stride(0), M, N, K, ptr(bias),
                                   qscale_cols, qscale, stream_ptr()))
    return out


def attention_fwd(q, k, v, heads, scale=None, q_prescaled=False):
    """q [B,Sq,H*D], k/v [B,Skv,H*D] bf16 (D = 64, 128 or 192: zero-padded heads) -> (o, lse [B,H,Sq]).
    q_prescaled: q already holds (unscaled q) * scale * log2(e)."""
    B, Sq, C = q.shape
    Skv = k.shape[1]
    nd = C // heads // 64
    scale = scale if scale is not None else (C // heads) ** -0.5
    o = torch.empty(B, Sq, C, device=q.device, dtype=BF)
    lse = torch.empty(B, heads, Sq, device=q.device, dtype=torch.float32)
    fn = lib().pea_op_attention_fwd_prescaled if q_prescaled else lib().pea_op_attention_fwd
    check(fn(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(o), C, ptr(lse),
                                     B, heads, Sq, Skv, scale, nd, stream_ptr()))
    return o, lse


def attention_fwd_masked(q, k, v, heads, causal=False, kv_len=None, scale=None):
    """text-encoder attention: head_dim 64, optional causal mask and per-sample key counts (int32 [B])"""
    B, Sq, C = q.shape
    Skv = k.shape[1]
    scale = scale if scale is not None else (C // heads) ** -0.5
    o = torch.empty(B, Sq, C, device=q.device, dtype=BF)
    check(lib().pea_op_attention_fwd_masked(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(o), C, None,
                                            B, heads, Sq, Skv, scale, int(causal), ptr(kv_len), stream_ptr()))
    return o


def attention_bwd(q, k, v, o, do, lse, heads, scale=None, q_prescaled=False):
    """-> (dq, dk, dv); with q_prescaled dq is still the gradient w.r.t. the UNSCALED q"""
    B, Sq, C = q.shape
    Skv = k.shape[1]
    nd = C // heads // 64
    scale = scale if scale is not None else (C // heads) ** -0.5
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(2, B, heads, Sq, device=q.device, dtype=torch.float32)     # scratch: -delta and -lse*log2(e) per row
    nb = lib().pea_op_attention_bwd_scratch_bytes(B, heads, Sq, Skv, nd)
    scratch = torch.empty(nb, device=q.device, dtype=torch.uint8) if nb else None
    fn = lib().pea_op_attention_bwd_prescaled if q_prescaled else lib().pea_op_attention_bwd
    check(fn(ptr(q), q.stride(1), ptr(k), k.stride(1), ptr(v), v.stride(1), ptr(o), C, ptr(do),
                                     C, ptr(lse), ptr(delta), ptr(dq), C, ptr(dk), C, ptr(dv), C, B, heads, Sq, Skv,
                                     scale, 0, 0, nd, ptr(scratch), stream_ptr()))
    return dq, dk, dv


def geglu_fwd(hg):
    rows, two = hg.shape
    y = torch.empty(rows, two // 2, device=hg.device, dtype=BF)
    check(lib().pea_op_geglu_fwd(ptr(hg), ptr(y), rows, two // 2, stream_ptr()))
    return y


def geglu_bwd(hg, dy):
    d = torch.empty_like(hg)
    check(lib().pea_op_geglu_bwd(ptr(hg), ptr(dy), ptr(d), hg.shape[0], hg.shape[1] // 2, stream_ptr()))
    return d


def sumpool2(x):
    B, H2, W2, C = x.shape
    y = torch.empty(B, H2 // 2, W2 // 2, C, device=x.device, dtype=BF)
    check(lib().pea_op_sumpool2(ptr(x), ptr(y), B, H2 // 2, W2 // 2, C, 0, stream_ptr()))
    return y


def timestep_embed(t_f32, dim):
    y = torch.empty(t_f32.numel(), dim, device=t_f32.device, dtype=BF)
    check(lib().pea_op_timestep_embed(ptr(t_f32), ptr(y), t_f32.numel(), dim, stream_ptr()))
    return y


def add_noise(x0, eps, t, ac):
    xt = torch.empty_like(x0)
    check(lib().pea_op_add_noise(ptr(x0), ptr(eps), ptr(t), ptr(ac), ptr(xt), x0.shape[0], x0[0].numel(), stream_ptr()))
    return xt


def kd_loss(taps_s: Sequence[torch.Tensor], taps_t: Sequence[torch.Tensor], eps_s, eps, eps_t, zh,
            feat_weight=0.1, nan_guard=False, grad_scale=1.0, want_grads=True):
    """-> (losses fp32[4] device, [dL/dtap_s], dL/d eps_s).  taps: bf16 [B, ...] paired layouts."""
    n = len(taps_s)
    B = eps_s.shape[0]
    dev = eps_s.device
    dt = [torch.empty_like(t) for t in taps_s] if want_grads else []
    de = torch.empty_like(eps_s) if want_grads else None
    arr = ctypes.c_void_p * max(n, 1)
    a_s = arr(*[t.data_ptr() for t in taps_s])
    a_t = arr(*[t.data_ptr() for t in taps_t])
    a_d = arr(*[t.data_ptr() for t in dt]) if want_grads else None
    per = (ctypes.c_longlong * max(n, 1))(*[t[0].numel() for t in taps_s])
    losses = torch.empty(4, device=dev, dtype=torch.float32)
    ws = torch.empty(lib().pea_op_kd_loss_workspace_bytes(n, per, eps_s[0].numel(), B), device=dev, dtype=torch.uint8)
    check(lib().pea_op_kd_loss(n, a_s, a_t, a_d, per, ptr(eps_s), ptr(eps), ptr(eps_t), ptr(de), eps_s[0].numel(),
                               ptr(zh), B, feat_weight, int(nan_guard), grad_scale, ptr(losses), ptr(ws),
                               stream_ptr()))
    return losses, dt, de


def adamw_(w, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    check(lib().pea_op_adamw(ptr(w), ptr(g), ptr(m), ptr(v), w.numel(), lr, beta1, beta2, eps, weight_decay, step,
                             grad_scale, stream_ptr()))


# ---- inference denoise-loop glue (tests/test_sdxl_zh.py:376-406)
def cfg_combine(noise_pred2, guidance_scale, guidance_rescale=0.0):
    """`u, t = noise_pred.chunk(2); u + g * (t - u)` (:394-395) and, for guidance_rescale > 0, `rescale_noise_cfg`
    (:44-56).  noise_pred2: fp32 [2B, ...] on the GPU, unconditional half first."""
    assert noise_pred2.dtype == torch.float32 and noise_pred2.shape[0] % 2 == 0
    B = noise_pred2.shape[0] // 2
    x = noise_pred2.contiguous()
    out = torch.empty((B,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    ws = None
    if guidance_rescale > 0.0:
        ws = torch.empty(lib().pea_op_cfg_combine_workspace_bytes(B), device=x.device, dtype=torch.uint8)
    check(lib().pea_op_cfg_combine(ptr(x), ptr(out), B, out[0].numel(), float(guidance_scale), float(guidance_rescale),
                                   ptr(ws), stream_ptr()))
    return out


def dpm_update_(sample, eps, x0_prev, alpha_s, sigma_s, c_s, c_0, c_1):
    """In place: x0 = (sample - sigma_s*eps)/alpha_s; sample <- c_s*sample + c_0*x0 + c_1*x0_prev; x0_prev <- x0."""
    for t in (sample, eps, x0_prev):
        assert t.dtype == torch.float32 and t.is_contiguous()
    check(lib().pea_op_dpm_update(ptr(sample), ptr(eps), ptr(x0_prev), sample.numel(), float(alpha_s), float(sigma_s),
                                  float(c_s), float(c_0), float(c_1), stream_ptr()))
    return sample
