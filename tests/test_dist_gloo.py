"""The N > 1 path on CPU: two gloo ranks, pixel rows dealt round-robin, each rank renders its
shard (the oracle stands in for the GPU renderer here), the accumulation buffers are summed by
tyrant_amd.dist.reduce_accum -- the same function bench.py calls with RCCL tensors."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

W, H, N, SPP = 48, 32, 1024, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch

    from oracle import pyorc
    from tyrant_amd import dist as tdist
    from tyrant_amd import scenes

    r, lr, w = tdist.env_rank_world()
    assert (r, w) == (rank, world)
    tdist.init_process_group("gloo")
    sc = scenes.cornell_box()
    nodes, prims = pyorc.bvh_build(sc.triangles, scenes.triangle_bboxes(sc.triangles))
    o = pyorc.Oracle(W, H, N, **tdist.shard_spec(rank, world, H))
    o.load_scene(sc, nodes, prims)
    o.render(SPP)
    mine = o.blit_buffer()
    np.save(os.path.join(out_dir, f"shard{rank}.npy"), mine)
    accum = torch.from_numpy(mine.copy()).reshape(-1)
    gathered = accum.clone()
    tdist.reduce_accum(accum, dst=0)
    assert tdist.agree_gather_works("cpu")
    tdist.gather_rows(gathered, H, W, rank, world, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "reduced.npy"), accum.numpy().reshape(-1, 4))
        np.save(os.path.join(out_dir, "gathered.npy"), gathered.numpy().reshape(-1, 4))
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_reduce(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0 = np.load(tmp_path / "shard0.npy").reshape(H, W, 4)
    s1 = np.load(tmp_path / "shard1.npy").reshape(H, W, 4)
    red = np.load(tmp_path / "reduced.npy").reshape(H, W, 4)
    # disjoint ownership: rank r wrote only rows y % 2 == r
    assert np.all(s0[1::2] == 0) and np.all(s1[0::2] == 0)
    assert np.all(s0[0::2, :, 3] == SPP) and np.all(s1[1::2, :, 3] == SPP)
    # the reduce is the frame: every pixel has exactly SPP completed paths, values are the shards'
    assert np.array_equal(red, s0 + s1)
    assert np.all(red[:, :, 3] == SPP)
    # the cheaper way to the same frame: gather of the rows each rank owns
    assert np.array_equal(np.load(tmp_path / "gathered.npy").reshape(H, W, 4), red)


def _gather_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch

    from tyrant_amd import dist as tdist

    tdist.init_process_group("gloo")
    h, w = 24, 5
    frame = torch.zeros(h * w * 4)
    view = frame.view(h, w, 4)
    for y in tdist.owned_rows(rank, world, h):
        view[y] = float(100 * rank + y)  # what this rank "rendered" into its rows
    summed = frame.clone()
    assert tdist.agree_gather_works("cpu")
    tdist.gather_rows(frame, h, w, rank, world, dst=0)
    tdist.reduce_accum(summed, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "g.npy"), frame.numpy())
        np.save(os.path.join(out_dir, "s.npy"), summed.numpy())
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_gather_equals_reduce(tmp_path, world):
    """the two ways of combining the frame agree for R = 4 and R = 8 (rows y % R == rank, the scaling runs' layouts),
    on synthetic row contents"""
    import torch.multiprocessing as mp

    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = np.load(tmp_path / "g.npy").reshape(24, 5, 4)
    assert np.array_equal(g, np.load(tmp_path / "s.npy").reshape(24, 5, 4))
    for y in range(24):
        assert np.all(g[y] == 100 * (y % world) + y)


def test_shard_spec_and_rows():
    sys.path.insert(0, ROOT)
    from tyrant_amd import dist as tdist

    assert tdist.shard_spec(3, 8, 1080) == {"rank": 3, "nranks": 8}
    with pytest.raises(ValueError):
        tdist.shard_spec(0, 7, 1080)
    rows = [set(tdist.owned_rows(r, 4, 24)) for r in range(4)]
    assert set().union(*rows) == set(range(24)) and sum(len(r) for r in rows) == 24
