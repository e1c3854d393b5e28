"""`HipUNet`: the UNet plug-in surface of the reference --
`unet(sample, t, encoder_hidden_states, added_cond_kwargs=..., return_dict=False)[0]`
(train_sdxl_zh.py:397,415; tests/test_sdxl_zh.py:384-391) -- backed by the HIP op tape in
libpea_hip.so.  `down_blocks[i]`, `mid_block`, `up_blocks[i]` accept `register_forward_hook`
so the reference's `cast_hook` (train_sdxl_zh.py:79-84) works unchanged."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import torch

from . import config as _cfg
from ._lib import PeaError, check, lib, ptr, stream_ptr


class _HookHandle:
    def __init__(self, lst, fn):
        self.lst, self.fn = lst, fn

    def remove(self):
        if self.fn in self.lst:
            self.lst.remove(self.fn)


class _BlockShim:
    """Hook-able stand-in for a diffusers block; `residuals_present` mirrors the `(hidden, res_samples)`
    tuple that down blocks return (reference getActivation, train_sdxl_zh.py:69-77)."""

    def __init__(self, name: str, tap_index: int, residuals_present: bool):
        self.name, self.tap_index, self.residuals_present = name, tap_index, residuals_present
        self._hooks: List = []

    def register_forward_hook(self, fn):
        self._hooks.append(fn)
        return _HookHandle(self._hooks, fn)


class _Config:
    def __init__(self, cfg):
        self.__dict__.update(cfg.__dict__)


class HipUNet:
    def __init__(self, cfg, batch: int, height: Optional[int] = None, width: Optional[int] = None, ctx_len: int = 77,
                 needs_grad: bool = False, share_weights_from: Optional["HipUNet"] = None,
                 residual_inputs: bool = False):
        if not torch.cuda.is_available():
            raise PeaError("HipUNet needs a MI355X (no CPU fallback)")
        self.cfg = cfg
        self.config = _Config(cfg)
        self.in_channels = cfg.in_channels
        self.B, self.H, self.W, self.L = batch, height or cfg.sample_size, width or cfg.sample_size, ctx_len
        self.needs_grad = needs_grad
        self.dtype = torch.bfloat16
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._h = ctypes.c_void_p()
        c = _cfg.to_c(cfg)
        self.residual_inputs = residual_inputs
        flags = (1 if needs_grad else 0) | (2 if residual_inputs else 0)   # PEA_UNET_GRAD | PEA_UNET_RESIDUAL_INPUTS
        check(lib().pea_unet_create(ctypes.byref(c), self.B, self.H, self.W, self.L, flags,
                                    int(share_weights_from is None), ctypes.byref(self._h)))
        if share_weights_from is not None:
            check(lib().pea_unet_share_weights(self._h, share_weights_from._h))
            self._weights_owner = share_weights_from      # keep alive
        n = len(cfg.block_out_channels)
        self.num_taps = lib().pea_unet_num_taps(self._h)
        nm = ctypes.create_string_buffer(16)
        self.tap_names = []
        for k in range(self.num_taps):
            check(lib().pea_unet_tap_name(self._h, k, nm, 16))
            self.tap_names.append(nm.value.decode())
        idx = {name: k for k, name in enumerate(self.tap_names)}
        self.down_blocks = [_BlockShim(f"d{i}", idx[f"d{i}"], True) for i in range(n)]
        # `mid_block_type: null` configs (SSD-1B) have no mid block: diffusers sets `unet.mid_block = None`
        self.mid_block = _BlockShim("m", idx["m"], False) if "m" in idx else None
        self.up_blocks = [_BlockShim(f"u{i}", idx[f"u{i}"], False) for i in range(n)]

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().pea_unet_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def release_activations(self):
        """free the activation / gradient arenas (weights stay); the next forward allocates them again"""
        check(lib().pea_unet_release_activations(self._h))

    # ---------------------------------------------------------------- weights
    def weight_table(self) -> Dict[str, tuple]:
        """{diffusers key: torch shape}"""
        out = {}
        name = ctypes.create_string_buffer(256)
        numel, kind, d0, d1 = ctypes.c_longlong(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        for i in range(lib().pea_unet_num_weights(self._h)):
            check(lib().pea_unet_weight_info(self._h, i, name, 256, ctypes.byref(numel), ctypes.byref(kind),
                                             ctypes.byref(d0), ctypes.byref(d1)))
            k = kind.value
            if k == 0:
                shape = (d0.value,)
            elif k == 1:
                shape = (d0.value, d1.value)
            else:
                shape = (d0.value, d1.value, 3, 3)
            out[name.value.decode()] = shape
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        table = self.weight_table()
        missing = [k for k in table if k not in sd]
        unexpected = [k for k in sd if k not in table]
        if strict and (missing or unexpected):
            raise PeaError(f"load_state_dict: missing={missing[:5]} unexpected={unexpected[:5]}")
        for k, shape in table.items():
            if k not in sd:
                continue
            t = sd[k]
            n = 1
            for s in shape:
                n *= s
            if t.numel() != n:
                raise PeaError(f"load_state_dict: {k} has shape {tuple(t.shape)}, expected {shape} (or 1x1 conv)")
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            check(lib().pea_unet_load_weight(self._h, k.encode(), ptr(t), t.numel(), stream_ptr()))
        torch.cuda.current_stream().synchronize()      # staging tensors above are freed after this call
        return missing, unexpected

    def init_random(self, seed: int = 0):
        check(lib().pea_unet_init_random(self._h, seed, stream_ptr()))

    def memory(self):
        w, a, g, n = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_int()
        check(lib().pea_unet_memory(self._h, ctypes.byref(w), ctypes.byref(a), ctypes.byref(g), ctypes.byref(n)))
        return {"weight_bytes": w.value, "activation_bytes": a.value, "grad_bytes": g.value, "n_ops": n.value}

    # ---------------------------------------------------------------- forward
    def __call__(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, cross_attention_kwargs=None,
                 return_dict=False, down_block_additional_residuals=None, mid_block_additional_residual=None):
        if down_block_additional_residuals is not None or mid_block_additional_residual is not None:
            self.set_additional_residuals(down_block_additional_residuals, mid_block_additional_residual)
        elif self.residual_inputs and self._residuals_set:
            self.set_additional_residuals(None, None)        # a call without the kwargs is a plain UNet call
        B = sample.shape[0]
        if tuple(sample.shape) != (self.B, self.in_channels, self.H, self.W):
            raise PeaError(f"HipUNet built for {(self.B, self.in_channels, self.H, self.W)}, got {tuple(sample.shape)}")
        x = sample.detach().to(self.device, torch.float32).contiguous()
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
        t = t.to(self.device, torch.float32).reshape(-1).expand(B).contiguous()
        ehs = encoder_hidden_states.detach().to(self.device)
        if tuple(ehs.shape) != (self.B, self.L, self.cfg.cross_attention_dim):
            raise PeaError(f"encoder_hidden_states {tuple(ehs.shape)} != {(self.B, self.L, self.cfg.cross_attention_dim)}")
        e_dt = 1 if ehs.dtype == torch.bfloat16 else 0
        ehs = ehs.contiguous() if e_dt else ehs.float().contiguous()
        text = tid = None
        t_dt = 0
        if self.cfg.addition_embed_type == "text_time":
            text = added_cond_kwargs["text_embeds"].detach().to(self.device)
            t_dt = 1 if text.dtype == torch.bfloat16 else 0
            text = text.contiguous() if t_dt else text.float().contiguous()
            tid = added_cond_kwargs["time_ids"].detach().to(self.device, torch.float32).contiguous()
        eps = torch.empty(self.B, self.cfg.out_channels, self.H, self.W, device=self.device, dtype=torch.float32)
        check(lib().pea_unet_forward(self._h, ptr(x), ptr(t), ptr(ehs), e_dt, ptr(text), t_dt, ptr(tid), ptr(eps),
                                     stream_ptr()))
        self._keep = (x, t, ehs, text, tid)
        for blk in list(self.down_blocks) + [self.mid_block] + list(self.up_blocks):
            if blk._hooks:
                tap = self.tap(blk.tap_index)
                out = (tap, ()) if blk.residuals_present else tap
                for fn in list(blk._hooks):
                    fn(blk, (), out)
        out = eps.to(sample.dtype) if sample.dtype in (torch.float16, torch.bfloat16) else eps
        return (out,)

    # ---------------------------------------------------------------- ControlNet residual inputs
    _residuals_set = False

    def residual_shapes(self):
        """[(C, H, W)] of `down_block_additional_residuals` followed by `mid_block_additional_residual` (last)."""
        n = lib().pea_unet_num_residuals(self._h)
        out = []
        for i in range(n):
            c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            check(lib().pea_unet_residual_info(self._h, i, ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)))
            out.append((c.value, h.value, w.value))
        return out

    def set_additional_residuals(self, down, mid, scale: float = 1.0):
        """`down_block_additional_residuals=..., mid_block_additional_residual=...` of the UNet call
        (tests/test_sdxl_zh_controlnet.py:534-535): NCHW tensors as a torch ControlNet returns them."""
        if not self.residual_inputs:
            raise PeaError("HipUNet was created without residual_inputs=True")
        shapes = self.residual_shapes()
        items = (list(down) if down is not None else [None] * (len(shapes) - 1)) + [mid]
        if len(items) != len(shapes):
            raise PeaError(f"{len(items) - 1} down residuals given, the UNet has {len(shapes) - 1}")
        keep, ptrs = [], (ctypes.c_void_p * len(items))()
        dts = {t.dtype for t in items if t is not None}
        dt = torch.bfloat16 if dts == {torch.bfloat16} else torch.float32
        for i, (t, (C, H, W)) in enumerate(zip(items, shapes)):
            if t is None:
                ptrs[i] = None
                continue
            if tuple(t.shape) != (self.B, C, H, W):
                raise PeaError(f"residual {i}: {tuple(t.shape)} != {(self.B, C, H, W)}")
            t = t.detach().to(self.device, dt).contiguous()
            keep.append(t)
            ptrs[i] = t.data_ptr()
        check(lib().pea_unet_set_residuals(self._h, len(items), ptrs, 1 if dt == torch.bfloat16 else 0, float(scale),
                                           stream_ptr()))
        self._keep_res = keep
        self._residuals_set = any(t is not None for t in items)

    def clear_residuals(self):
        self.set_additional_residuals(None, None)

    def tap(self, k: int, grad: bool = False) -> torch.Tensor:
        """feature tap k (cast_hook order) as an NCHW fp32 tensor"""
        B, H, W, C = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().pea_unet_tap_info(self._h, k, None, None, ctypes.byref(B), ctypes.byref(H), ctypes.byref(W),
                                      ctypes.byref(C)))
        out = torch.empty(B.value, C.value, H.value, W.value, device=self.device, dtype=torch.float32)
        check(lib().pea_unet_tap_export_nchw(self._h, k, int(grad), ptr(out), stream_ptr()))
        return out

    def set_tap_grad(self, k: int, seed: torch.Tensor):
        """write a gradient seed (NCHW, the tap's shape) into tap k's gradient buffer in its storage layout; pass bit k in
        `backward(..., tap_seed_mask)` afterwards"""
        s = seed.detach().to(self.device, torch.float32).contiguous()
        check(lib().pea_unet_tap_import_grad_nchw(self._h, k, ptr(s), stream_ptr()))

    def tap_layout(self, k: int) -> int:
        """0 = NHWC, 1 = depth-to-space (include/pea_hip.h: pea_unet_tap_layout)"""
        return int(lib().pea_unet_tap_layout(self._h, k))

    def tap_pointers(self, k: int):
        """-> (data ptr, grad ptr, (B, H, W, C), layout); layout as `tap_layout` -- the pointers of a depth-to-space tap are
        only handed out once the layout has been asked for (pea_unet_tap_info)"""
        layout = self.tap_layout(k)
        d, g = ctypes.c_void_p(), ctypes.c_void_p()
        B, H, W, C = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().pea_unet_tap_info(self._h, k, ctypes.byref(d), ctypes.byref(g), ctypes.byref(B), ctypes.byref(H),
                                      ctypes.byref(W), ctypes.byref(C)))
        return d.value, g.value, (B.value, H.value, W.value, C.value), layout

    # ---------------------------------------------------------------- backward (data gradients only)
    def backward(self, d_eps: Optional[torch.Tensor], tap_seed_mask: int = 0):
        """-> (d_encoder_hidden_states [B,L,cross] fp32, d_text_embeds [B,pooled] fp32 or None)"""
        de = d_eps.detach().to(self.device, torch.float32).contiguous() if d_eps is not None else None
        check(lib().pea_unet_backward(self._h, ptr(de), tap_seed_mask, stream_ptr()))
        pe, pt = ctypes.c_void_p(), ctypes.c_void_p()
        check(lib().pea_unet_input_grads(self._h, ctypes.byref(pe), ctypes.byref(pt)))
        d_ehs = d_text = None
        if pe.value:
            d_ehs = torch.empty(self.B, self.L, self.cfg.cross_attention_dim, device=self.device)
            check(lib().pea_op_cast_bf16_f32(pe, ptr(d_ehs), d_ehs.numel(), stream_ptr()))
        if pt.value:
            d_text = torch.empty(self.B, self.cfg.pooled_dim, device=self.device)
            check(lib().pea_op_cast_bf16_f32(pt, ptr(d_text), d_text.numel(), stream_ptr()))
        return d_ehs, d_text
