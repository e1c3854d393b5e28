"""Data parallelism for the KD step: one process per GPU, batches sharded across ranks, frozen UNets
replicated, and exactly ONE collective per step -- an all-reduce (sum) of the flat fp32 adapter-gradient
buffer followed by 1/world scaling.  This replaces DeepSpeed ZeRO-1's gradient all-reduce + parameter
all-gather + overflow all-reduce (train_sdxl_zh.sh:22,87; utils/model_utils.py:57-67): the optimizer is
replicated, so no parameter traffic exists.  `backend="nccl"` is RCCL on ROCm (xGMI); the same code
runs over `gloo` on CPU tensors (tests/test_dp_cpu.py).

On the GPU the collective is `NativeComm`: an RCCL communicator owned by libpea_hip.so (C ABI `pea_comm_*`,
include/pea_hip.h) with its own HIP stream -- the all-reduce is launched right after the adapter wgrad, the compute
stream never waits for it, and the optimizer joins it.  torch.distributed only carries the 128-byte ncclUniqueId to the
other ranks (and the bench's barrier / max-over-ranks timing).

Every rank computes the reference's local-batch mean loss (train_sdxl_zh.py:405,417,429: masks are not
renormalised), so averaging the per-rank gradients equals the gradient of the global-batch mean when
all ranks hold the same number of samples."""
from __future__ import annotations

import os
from typing import Optional

import ctypes

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> int:
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run). Returns world."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    else:
        dist.init_process_group(backend)
    return world


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_mean_(flat_grad: torch.Tensor, async_op: bool = False):
    """In-place average of the flat adapter gradient over all ranks (ONE collective)."""
    w = world_size()
    if w == 1:
        return None
    work = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=async_op)
    if async_op:
        return work                      # caller waits, then calls flat_grad.div_(world_size())
    flat_grad.div_(w)
    return None


def shard_batch(batch: dict, rank_: int, world: int) -> dict:
    """Rank r takes samples [r*B/world, (r+1)*B/world) of every batched tensor (independent samples)."""
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] % world == 0:
            n = v.shape[0] // world
            out[k] = v[rank_ * n:(rank_ + 1) * n].contiguous()
        else:
            out[k] = v
    return out


def broadcast_params_(flat_param: torch.Tensor, src: int = 0):
    """Make every replica start from rank `src`'s adapter parameters."""
    if world_size() > 1:
        dist.broadcast(flat_param, src=src)


class NativeComm:
    """RCCL communicator + dedicated comm stream inside libpea_hip.so (`pea_comm_init`, `pea_allreduce_grads`,
    `pea_comm_join`).  One per process / GPU."""

    def __init__(self, rank_: int, world: int, unique_id: bytes):
        from ._lib import check, lib
        assert len(unique_id) == 128
        self._h = ctypes.c_void_p()
        self.rank, self.world = rank_, world
        check(lib().pea_comm_init(rank_, world, unique_id, ctypes.byref(self._h)))

    @staticmethod
    def new_unique_id() -> bytes:
        from ._lib import check, lib
        buf = ctypes.create_string_buffer(128)
        check(lib().pea_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_env(cls) -> "NativeComm":
        """rank 0 creates the ncclUniqueId; an initialised torch.distributed group (any backend) ships it."""
        if dist.is_available() and dist.is_initialized():
            r, w = dist.get_rank(), dist.get_world_size()
            box = [cls.new_unique_id() if r == 0 else None]
            if w > 1:
                dist.broadcast_object_list(box, src=0)
            return cls(r, w, box[0])
        return cls(0, 1, cls.new_unique_id())

    def allreduce_mean_async(self, flat_grad: torch.Tensor, compute_stream=None):
        """comm stream: wait for `compute_stream`'s work so far, all-reduce(sum) in place, x 1/world.  Returns at once."""
        from ._lib import check, lib, ptr, stream_ptr
        assert flat_grad.is_cuda and flat_grad.dtype == torch.float32 and flat_grad.is_contiguous()
        s = stream_ptr() if compute_stream is None else ctypes.c_void_p(compute_stream.cuda_stream)
        check(lib().pea_allreduce_grads(self._h, ptr(flat_grad), flat_grad.numel(), s))

    def join(self, stream=None):
        """make `stream` (default: torch's current stream) wait for the last all-reduce"""
        from ._lib import check, lib, stream_ptr
        s = stream_ptr() if stream is None else ctypes.c_void_p(stream.cuda_stream)
        check(lib().pea_comm_join(self._h, s))

    def last_ms(self) -> float:
        from ._lib import check, lib
        ms = ctypes.c_float()
        check(lib().pea_comm_last_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def last_exposed_ms(self) -> float:
        """how long the joining stream stood still for the last all-reduce (0: fully overlapped)"""
        from ._lib import check, lib
        ms = ctypes.c_float()
        check(lib().pea_comm_last_exposed_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def broadcast_(self, flat: torch.Tensor, root: int = 0):
        from ._lib import check, lib, ptr, stream_ptr
        check(lib().pea_comm_broadcast(self._h, ptr(flat), flat.numel(), root, stream_ptr()))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            from ._lib import lib
            lib().pea_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
