"""Data parallelism for the KD step: one process per GPU, batches sharded across ranks, frozen UNets
replicated, and exactly ONE collective per step -- an all-reduce (sum) of the flat fp32 adapter-gradient
buffer followed by 1/world scaling.  This replaces DeepSpeed ZeRO-1's gradient all-reduce + parameter
all-gather + overflow all-reduce (train_sdxl_zh.sh:22,87; utils/model_utils.py:57-67): the optimizer is
replicated, so no parameter traffic exists.  `backend="nccl"` is RCCL on ROCm (xGMI); the same code
runs over `gloo` on CPU tensors (tests/test_dp_cpu.py).

On the GPU the collective is `NativeComm`: an RCCL communicator owned by libpea_hip.so (C ABI `pea_comm_*`,
include/pea_hip.h) with its own HIP stream -- the all-reduce is launched right after the adapter wgrad, the compute
stream never waits for it, and the optimizer joins it.  The CONTROL PLANE is a `gloo` group on CPU tensors
(`init_control_plane`): it carries the 128-byte ncclUniqueId to the other ranks, the bench's barriers and its
max-over-ranks of wall times, so the library's communicator is the only RCCL communicator in the process (one
bootstrap, one set of xGMI rings).  The rendezvous of that communicator is bounded (`pea_comm_init_timeout`): a
missing rank turns into `CommTimeout` on the ranks that did arrive, never into a hang.

Every rank computes the reference's local-batch mean loss (train_sdxl_zh.py:405,417,429: masks are not
renormalised), so averaging the per-rank gradients equals the gradient of the global-batch mean when
all ranks hold the same number of samples."""
from __future__ import annotations

import os
from typing import Optional

import ctypes

import torch
import torch.distributed as dist


class CommTimeout(RuntimeError):
    """the RCCL rendezvous did not complete in time (pea_comm_init_timeout -> PEA_E_TIMEOUT): the process must exit"""


def _single_node_env_defaults():
    """one node, rendezvous over loopback: make the socket bootstraps of gloo and RCCL use `lo` unless the caller chose
    an interface (the container's hostname need not resolve, and a loopback bootstrap cannot be filtered; the data path
    is xGMI / shared memory either way).  Must run before the first process group / communicator is created."""
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    if addr in ("127.0.0.1", "localhost", "::1"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this host driver)


def init_control_plane(timeout_s: float = 600.0) -> int:
    """gloo process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run or bench.py's own launcher): the
    control plane of an N-rank run -- id broadcast, barriers, scalar reductions on CPU tensors.  Never touches the GPU.
    Returns world.  A rank that does not arrive within `timeout_s` raises (gloo's store timeout) instead of hanging."""
    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if dist.is_initialized():
        return dist.get_world_size()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    _single_node_env_defaults()
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))
    assert dist.get_world_size() == world
    return world


def control_barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def control_max(x: float) -> float:
    """max over ranks of a host scalar (CPU tensor: works on gloo and needs no device)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([x], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


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

    def __init__(self, rank_: int, world: int, unique_id: bytes, timeout_s: Optional[float] = None):
        from ._lib import PeaError, lib
        assert len(unique_id) == 128
        self._h = ctypes.c_void_p()
        self.rank, self.world = rank_, world
        if timeout_s is None:
            timeout_s = float(os.environ.get("PEA_COMM_TIMEOUT_S", "600"))
        rc = lib().pea_comm_init_timeout(rank_, world, unique_id, ctypes.c_double(timeout_s), ctypes.byref(self._h))
        if rc == -6:                                   # PEA_E_TIMEOUT: the rendezvous thread is still inside RCCL -- exit
            raise CommTimeout(lib().pea_last_error().decode())
        if rc != 0:
            raise PeaError(f"pea error {rc}: {lib().pea_last_error().decode()}")

    @staticmethod
    def new_unique_id() -> bytes:
        from ._lib import check, lib
        buf = ctypes.create_string_buffer(128)
        check(lib().pea_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_env(cls, timeout_s: Optional[float] = None) -> "NativeComm":
        """rank 0 creates the ncclUniqueId; an initialised torch.distributed group (the gloo control plane; any backend
        works) ships it.  The caller has bound its HIP device (torch.cuda.set_device) before this."""
        _single_node_env_defaults()
        if dist.is_available() and dist.is_initialized():
            r, w = dist.get_rank(), dist.get_world_size()
            box = [cls.new_unique_id() if r == 0 else None]
            if w > 1:
                dist.broadcast_object_list(box, src=0)
            return cls(r, w, box[0], timeout_s)
        return cls(0, 1, cls.new_unique_id(), timeout_s)

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
