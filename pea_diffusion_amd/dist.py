"""Data parallelism for the KD step: one process per GPU, batches sharded across ranks, frozen UNets
replicated, and exactly ONE collective per step -- an all-reduce (sum) of the flat fp32 adapter-gradient
buffer followed by 1/world scaling.  This replaces DeepSpeed ZeRO-1's gradient all-reduce + parameter
all-gather + overflow all-reduce (train_sdxl_zh.sh:22,87; utils/model_utils.py:57-67): the optimizer is
replicated, so no parameter traffic exists.  `backend="nccl"` is RCCL on ROCm (xGMI); the same code
runs over `gloo` on CPU tensors (tests/test_dp_cpu.py).

Every rank computes the reference's local-batch mean loss (train_sdxl_zh.py:405,417,429: masks are not
renormalised), so averaging the per-rank gradients equals the gradient of the global-batch mean when
all ranks hold the same number of samples."""
from __future__ import annotations

import os
from typing import Optional

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
