"""Multi-GPU plumbing: one process per GPU, frames sharded, weights broadcast once.

The reference is single-device (test/yolo_test.cpp:16 `cudaSetDevice(0)`); frames
are independent, so the path shards as pure data parallelism:
rank r owns frames r, r+N, r+2N, ...; every rank holds a full replica of the
6 MB weight blob, which only rank 0 reads or generates and which reaches the
other ranks through ONE `broadcast` (RCCL over xGMI with backend "nccl"; gloo on
CPU in the tests).  There is no per-frame collective.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def device_index(local_rank: int) -> int:
    """HIP device of this rank: LOCAL_RANK, unless IRMV_FORCE_DEVICE pins every rank to one card (rehearsal)."""
    forced = os.environ.get("IRMV_FORCE_DEVICE")
    return int(forced) if forced is not None else local_rank


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            # IRMV_DIST_BACKEND=gloo: rehearse the N > 1 path with several ranks on ONE GPU (RCCL refuses that)
            backend = os.environ.get("IRMV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(device_index(local_rank))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_frames(total_frames: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: frame i -> rank i mod world."""
    return list(range(rank, total_frames, world))


def broadcast_blob(blob: Optional[bytes], device: torch.device, src: int = 0) -> torch.Tensor:
    """Rank `src` passes the .irmw bytes, the others None; returns a uint8 tensor
    on `device` holding the blob on every rank (one size broadcast + one payload
    broadcast)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        assert blob is not None
        return torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    if rank == src:
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(t, src)
    return t


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float, device: torch.device) -> float:
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device: torch.device) -> float:
    if not dist.is_initialized():
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
