"""Multi-GPU glue: one process per GPU, pixel rows dealt round-robin over the ranks, one
sum-reduction of the accumulation buffer at the end of a render (RCCL over xGMI on GPUs,
gloo in the CPU tests).  The reference is single-GPU (main.cpp:94 computes `multi_gpu` and
never uses it); SURVEY.md section 8e defines this scheme.

Two ways to combine the ranks' buffers (both give rank 0 the same frame): `reduce_accum`, the sum north_star
names (every rank ships the full 33 MB frame, mostly zeros, through a ring), and `gather_rows`, which ships
only the rows a rank owns (1/R of the frame, point-to-point into rank 0: xGMI is point-to-point, so rank 0
takes the R-1 pieces over R-1 different links).  bench.py uses the gather when every rank reports that a probe
of it worked, the reduce otherwise.

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


def gather_rows(accum, height: int, width: int, rank: int, world: int, dst: int = 0):
    """Ranks own disjoint rows (y % world == rank) of the full-frame buffer `accum` (flat, float4 per pixel, zero
    outside the owned rows): collect every rank's rows into `accum` on rank `dst`.  Same result as reduce_accum there."""
    import torch
    import torch.distributed as dist

    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return accum
    frame = accum.view(height, width, 4)
    mine = frame[rank::world].contiguous()
    if rank == dst:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.gather(mine, gather_list=parts, dst=dst)
        for r, part in enumerate(parts):
            if r != dst:
                frame[r::world] = part
    else:
        dist.gather(mine, gather_list=None, dst=dst)
    return accum


def agree_gather_works(device) -> bool:
    """probe dist.gather on a tiny tensor on every rank and agree on the outcome (an all-reduce of the success
    flags), so that all ranks take the same branch afterwards"""
    import torch
    import torch.distributed as dist

    ok = 1
    try:
        rank, world = dist.get_rank(), dist.get_world_size()
        t = torch.full((4,), float(rank), device=device)
        parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, gather_list=parts, dst=0)
        if rank == 0 and any(float(p[0]) != float(r) for r, p in enumerate(parts)):
            ok = 0
    except Exception:  # noqa: BLE001 -- any failure means: use the reduce
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))
