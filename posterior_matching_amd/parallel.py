"""Data parallelism for the PM-VAE step: one process per GPU, gradients summed with ONE
all-reduce of the flat gradient buffer per step (RCCL over xGMI on MI355X, `nccl` backend; `gloo`
on CPU for tests).  This is what bax.Trainer(num_devices=N) does with pmap + pmean in the
reference (train_pm_vdvae.py:146-154): batch sizes in the configs are PER DEVICE
(README.md:139-141), so scaling is weak and the loss is the mean of per-rank means.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend: str = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def shard_rows(global_batch: int, rank: int, world: int) -> slice:
    """Rank r owns rows [r*B/N, (r+1)*B/N) of a global batch."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by {world} ranks")
    per = global_batch // world
    return slice(rank * per, (rank + 1) * per)


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of the flat gradient buffer (the optimizer divides by world size)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def allreduce_mean_scalars(values: torch.Tensor) -> torch.Tensor:
    """Mean over ranks of a few logging scalars (loss / aux), only at logging steps."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
        values /= dist.get_world_size()
    return values
