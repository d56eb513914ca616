"""Multi-GPU glue: one process per GPU, pixel rows dealt round-robin over the ranks, one
sum-reduction of the accumulation buffer at the end of a render (RCCL over xGMI on GPUs,
gloo in the CPU tests).  The reference is single-GPU (main.cpp:94 computes `multi_gpu` and
never uses it); SURVEY.md section 8e defines this scheme.

Why rows interleaved by rank: sky rows and geometry rows cost very different numbers of
segments per path; dealing rows y % R == r gives every rank the same mix.  Why one reduce:
ranks own disjoint pixels, every other element of their full-frame buffer is zero, so a
sum over ranks IS the frame; 33 MB at 1080p is ~0.75 ms on a per-link-bound xGMI ring.
"""
from __future__ import annotations

import os


def env_rank_world() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend: str):
    import torch.distributed as dist

    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend)
    return dist


def shard_spec(rank: int, world: int, height: int) -> dict:
    """the (rank, nranks) pair of tyr_config; image rows must deal out evenly"""
    if height % world != 0:
        raise ValueError(f"height {height} is not divisible by {world} ranks")
    return {"rank": rank, "nranks": world}


def owned_rows(rank: int, world: int, height: int):
    return range(rank, height, world)


def reduce_accum(accum, dst: int = 0):
    """sum the ranks' full-frame accumulation buffers (float4 per pixel) onto rank `dst`"""
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
