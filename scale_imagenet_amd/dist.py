"""Batch sharding across GPUs: one process per GPU, RCCL (torch.distributed "nccl") over xGMI.

The path shards by image: eval-mode BatchNorm uses running statistics and nothing in the
forward crosses samples.  Every rank builds its own tables from the same
state_dict (no parameter broadcast, unlike DataParallel.replicate / the DDP constructor,
main.py:181-192); the only exchange is the gather of logits that
``torch.nn.DataParallel.gather`` performs on GPU 0 in the reference (main.py:192), here one
all-gather of ``[B/world, 1000]`` fp32.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """(first image, count) of ``rank``'s contiguous shard; earlier ranks take the remainder."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(n_total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world, local_rank).  Single process when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("TTNET_DIST_BACKEND", backend)     # rehearsal on one GPU: "gloo"
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def all_gather_logits(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Gather per-rank logits ``[count_r, C]`` (shards from ``shard_bounds``) into ``[n_total, C]``
    in image order on every rank.  Ragged shards are padded to the largest and trimmed."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if local.shape[0] != n_total:
            raise ValueError("single process must hold the whole batch")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    first, count = shard_bounds(n_total, rank, world)
    if local.shape[0] != count:
        raise ValueError(f"rank {rank}: expected {count} rows, got {local.shape[0]}")
    width = shard_bounds(n_total, 0, world)[1]
    if count < width:
        pad = torch.zeros((width - count, local.shape[1]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad])
    if local.is_cuda and dist.get_backend(group) == "gloo":     # rehearsal path: gloo gathers on the host
        host = torch.empty((world * width, local.shape[1]), dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.cpu().contiguous(), group=group)
        out = host.to(local.device)
    else:
        out = torch.empty((world * width, local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    if n_total == world * width:
        return out
    rows = [out[r * width: r * width + shard_bounds(n_total, r, world)[1]] for r in range(world)]
    return torch.cat(rows)
