"""CPU ORACLE (test infrastructure, NOT product code) -- optional bf16-STORAGE mode of the restatements.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

The HIP path keeps every activation tensor AND every gradient tensor in bf16 between kernels (fp32 only inside a kernel:
MFMA accumulators, epilogue arithmetic, statistics; pea_diffusion_amd/csrc/model.hip).  The plain
oracle runs in fp32 throughout, so an end-to-end comparison against it carries the bf16 storage noise of ~900 chained ops
(measured 6-8e-3 relative L2) and cannot see a defect smaller than that.  Inside `with bf16_storage():` the restatements
round a tensor to bf16 wherever the product stores one -- forward value and, through autograd, the gradient that flows back
through the same point -- and keep fp32 wherever the product fuses (conv + bias + time-embedding row, projection + bias +
residual, GroupNorm + SiLU, GEGLU from the fp32 accumulators).  One WEIGHT tensor is part of it: the upsampler convs are
stored as four 2 x 2 kernels of summed taps (sub-pixel form), each sum rounded to bf16 -- unet_ref.Upsample2D._subpixel
restates exactly that in this mode.  Outside the context `st()` is the identity: the fp32 oracle
and every golden fixture generated from it are bit-for-bit what they were.

What the mode does NOT model (left in the tightened tolerances): the order of fp32 accumulation inside a kernel, the flash
softmax's running offset, gradients of one tensor added pairwise in bf16 by the backward's accumulate epilogues (autograd
sums them in fp32 and the sum is rounded once here), the bf16 GEGLU stash of the backward's two factors.
"""
from __future__ import annotations

import contextlib

import torch

_ON = False


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def st(x: torch.Tensor) -> torch.Tensor:
    """a tensor the product stores: bf16-rounded value and gradient in storage mode, the identity otherwise"""
    return _RoundBF16.apply(x) if _ON else x


def enabled() -> bool:
    return _ON


@contextlib.contextmanager
def bf16_storage(on: bool = True):
    global _ON
    prev, _ON = _ON, bool(on)
    try:
        yield
    finally:
        _ON = prev
