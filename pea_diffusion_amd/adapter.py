"""`PEAAdapter`: the reference's `MLP` adapter (`proj`) as an nn.Module whose forward AND backward
run in libpea_hip.so.

Drop-in for `MLP(in_dim, out_dim, hidden_dim, out_dim1, use_residual)` of train_sdxl_zh.py:43-67 /
tests/test_sdxl_zh.py:59-84 (`out_dim1=None`: the SD1.5 `MLP(in_dim, out_dim, hidden_dim)` of
train_sd_zh.py:41-56): same constructor argument meaning, same `state_dict()` keys
(`layernorm.weight/bias`, `projector.{0,2,4}.weight`, `fc.weight/bias`), `.parameters()` for
`configure_optimizers` (train_sdxl_zh.py:166-168), `x1, x2 = proj(x)`.
The seven parameters are views of ONE flat fp32 buffer (`flat_param`, gradients `flat_grad`) so the
data-parallel all-reduce and the fused AdamW touch a single contiguous region."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.nn as nn

from ._lib import PeaError, check, lib, ptr, stream_ptr


class _AdapterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, *params):
        pooled, tokens = mod._forward_impl(x)
        ctx.mod = mod
        ctx.has_pooled = pooled is not None
        if pooled is None:
            return tokens
        return pooled, tokens

    @staticmethod
    def backward(ctx, *grads):
        mod = ctx.mod
        if ctx.has_pooled:
            d_pooled, d_tokens = grads
        else:
            d_pooled, d_tokens = None, grads[0]
        g = mod._backward_impl(d_pooled, d_tokens)
        outs = [g[o:o + p.numel()].view_as(p) for o, p in zip(mod._offsets, mod._plist())]
        return (None, None, *outs)


class PEAAdapter(nn.Module):
    def __init__(self, in_dim=1024, out_dim=1280, hidden_dim=2048, out_dim1: Optional[int] = 2048, use_residual=True):
        super().__init__()
        if use_residual:
            assert in_dim == out_dim          # same check as the reference (train_sdxl_zh.py:46-47)
        self.in_dim, self.out_dim, self.hidden_dim, self.out_dim1 = in_dim, out_dim, hidden_dim, out_dim1
        self.use_residual = use_residual
        self.layernorm = nn.LayerNorm(in_dim)
        self.projector = nn.Sequential(
            nn.Linear(in_dim, hidden_dim, bias=False), nn.GELU(),
            nn.Linear(hidden_dim, hidden_dim, bias=False), nn.GELU(),
            nn.Linear(hidden_dim, out_dim, bias=False))
        if out_dim1 is not None:
            self.fc = nn.Linear(out_dim, out_dim1)
        self._h = ctypes.c_void_p()
        self._prepared = None
        self.flat_param: Optional[torch.Tensor] = None
        self.flat_grad: Optional[torch.Tensor] = None
        self._synced_version = None
        self._src_key = None

    # ---- flat parameter plumbing
    def _plist(self):
        ps = [self.layernorm.weight, self.layernorm.bias, self.projector[0].weight, self.projector[2].weight,
              self.projector[4].weight]
        if self.out_dim1 is not None:
            ps += [self.fc.weight, self.fc.bias]
        return ps

    def _flatten(self):
        ps = self._plist()
        dev = ps[0].device
        if dev.type != "cuda":
            raise PeaError("PEAAdapter runs on the MI355X only: call .to('cuda') (no CPU fallback)")
        n = sum(p.numel() for p in ps)
        ok = (self.flat_param is not None and self.flat_param.device == dev and all(
            p.dtype == torch.float32 and p.data_ptr() == self.flat_param.data_ptr() + 4 * o
            for p, o in zip(ps, self._offsets)))
        # .half() / .bfloat16() modules (tests/test_sdxl_zh.py:92 builds `MLP(...).to(DEVICE).half()`): the presented
        # parameters cannot alias the fp32 master buffer, so they are re-read only when one of them was replaced or
        # written in place since the last flatten
        key = tuple((p.data_ptr(), p._version, p.dtype) for p in ps)
        if ok or (self.flat_param is not None and self.flat_param.device == dev and key == self._src_key):
            return
        flat = torch.empty(n, device=dev, dtype=torch.float32)
        offs, o = [], 0
        for p in ps:
            flat[o:o + p.numel()].copy_(p.detach().float().reshape(-1))
            offs.append(o)
            o += p.numel()
        self._present_dtype = ps[0].dtype
        if all(p.dtype == torch.float32 for p in ps):
            for p, oo in zip(ps, offs):
                p.data = flat[oo:oo + p.numel()].view_as(p)
        self.flat_param, self._offsets = flat, offs
        self._src_key = tuple((p.data_ptr(), p._version, p.dtype) for p in ps)
        self.flat_grad = torch.zeros_like(flat)
        self._synced_version = None
        if not self._h.value:
            check(lib().pea_adapter_create(self.in_dim, self.out_dim, self.hidden_dim, self.out_dim1 or 0,
                                           int(self.use_residual), ctypes.byref(self._h)))
            assert lib().pea_adapter_num_params(self._h) == n
        check(lib().pea_adapter_bind(self._h, ptr(flat)))

    def mark_updated(self):
        """call after an in-place parameter update (optimizer step) so the bf16 working copies refresh"""
        self._synced_version = None

    def _sync(self):
        ver = tuple(p._version for p in self._plist()) + (self.flat_param._version,)
        if ver != self._synced_version:
            check(lib().pea_adapter_sync(self._h, stream_ptr()))
            self._synced_version = ver

    def prepare(self, batch: int, L: int):
        self._flatten()
        if self._prepared != (batch, L):
            check(lib().pea_adapter_prepare(self._h, batch, L))
            self._prepared = (batch, L)
            self._synced_version = None      # the arena (incl. the bf16 weight copies) was reallocated

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().pea_adapter_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    # ---- forward / backward through the C ABI
    def _forward_impl(self, x):
        B, L, D = x.shape
        if D != self.in_dim:
            raise PeaError(f"PEAAdapter: input dim {D} != in_dim {self.in_dim}")
        self.prepare(B, L)
        self._sync()
        xin = x.detach()
        dt = 1 if xin.dtype == torch.bfloat16 else 0
        xin = xin.contiguous() if dt else xin.float().contiguous()
        tok_dim = self.out_dim1 if self.out_dim1 is not None else self.out_dim
        tokens = torch.empty(B, L, tok_dim, device=x.device, dtype=torch.float32)
        pooled = torch.empty(B, self.out_dim, device=x.device, dtype=torch.float32) if self.out_dim1 is not None else None
        check(lib().pea_adapter_forward(self._h, ptr(xin), dt, ptr(pooled), ptr(tokens), stream_ptr()))
        odt = x.dtype if x.dtype in (torch.float16, torch.bfloat16) else torch.float32
        return (pooled.to(odt) if pooled is not None else None), tokens.to(odt)

    def _backward_impl(self, d_pooled, d_tokens):
        dp = d_pooled.detach().float().contiguous() if d_pooled is not None else None
        dtk = d_tokens.detach().float().contiguous() if d_tokens is not None else None
        g = torch.empty_like(self.flat_param)
        check(lib().pea_adapter_backward(self._h, ptr(dp), ptr(dtk), ptr(g), 0, stream_ptr()))
        return g

    def forward(self, x):
        self._flatten()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._plist()):
            return _AdapterFn.apply(self, x, *self._plist())
        pooled, tokens = self._forward_impl(x)
        return tokens if pooled is None else (pooled, tokens)
