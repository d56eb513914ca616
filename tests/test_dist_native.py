"""tyr_dist_*: the multi-GPU combine behind the C ABI (tyrant_amd/csrc/host/dist.cpp).

CPU: the row-ownership arithmetic (no device, no communicator).  GPU (`-m gpu`): the pack / scatter kernels for R > 1
on synthetic frames, and a ONE-rank RCCL communicator end to end (ncclCommInitRank with nranks = 1 -- RCCL refuses two
ranks on one device, and the pool gives one GPU per box): gather and reduce both hand the root the frame a one-GPU
render holds.  The N > 1 exchange itself runs on the driver's 8-GPU node (bench.py --gpus N verifies every combined
frame and falls back to torch.distributed if the native path fails its check)."""
import ctypes as C

import numpy as np
import pytest

from conftest import built_scene


def test_row_ownership_math(hip):
    L = hip.lib()
    for H, R in ((1080, 1), (1080, 2), (1080, 4), (1080, 8), (2160, 8), (48, 3)):
        owners = np.zeros(H, dtype=np.int64)
        seen = set()
        for r in range(R):
            first, n = hip.dist_owned_rows(H, r, R)
            assert first == r and n == H // R
            for yl in range(n):
                y = first + yl * R
                rr, ll = hip.dist_row_owner(y, R)
                assert (rr, ll) == (r, yl)
                assert y not in seen
                seen.add(y)
                owners[y] = r
        assert len(seen) == H  # every row has exactly one owner
        assert np.array_equal(owners, np.arange(H) % R)
    f, n = C.c_uint32(), C.c_uint32()
    assert L.tyr_dist_owned_rows(1080, 0, 7, C.byref(f), C.byref(n)) == -1  # rows must deal out evenly
    assert L.tyr_dist_owned_rows(1080, 8, 8, C.byref(f), C.byref(n)) == -1  # rank >= nranks
    assert L.tyr_dist_owned_rows(1080, 0, 0, C.byref(f), C.byref(n)) == -1
    assert L.tyr_dist_row_owner(5, 0, C.byref(f), C.byref(n)) == -1
    # argument checks of the calls that would need a device: rejected before anything touches one
    assert L.tyr_dist_create(None, None, None, 0, 1) == -1 and L.tyr_dist_combine(None, 0, 0, None) == -1 and L.tyr_dist_wait(None) == -1
    assert L.tyr_dist_destroy(None) == 0
    assert L.tyr_dist_pack_rows(None, None, 64, 64, 0, 2, None) == -1 and L.tyr_dist_scatter_rows(None, None, 64, 64, 2, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("R", [2, 4, 8])
def test_pack_and_scatter_kernels(hip, R):
    """R full-frame buffers, each non-zero only on the rows its rank owns (what R ranks hold after a render): pack every
    rank's rows, lay the slabs end to end (what the root's receive buffer looks like), scatter -> the sum of the frames"""
    import torch

    L = hip.lib()
    W, H = 200, 16 * R
    rng = np.random.default_rng(R)
    frames = np.zeros((R, H, W, 4), dtype=np.float32)
    for r in range(R):
        frames[r, r::R] = rng.random((H // R, W, 4), dtype=np.float32) + 1.0
    slabs = torch.zeros(R * (H // R) * W * 4, dtype=torch.float32, device="cuda")
    slab_bytes = (H // R) * W * 16
    for r in range(R):
        fr = torch.from_numpy(frames[r].reshape(-1)).cuda()
        assert L.tyr_dist_pack_rows(fr.data_ptr(), slabs.data_ptr() + r * slab_bytes, W, H, r, R, None) == 0
        torch.cuda.synchronize()
        got = slabs.cpu().numpy()[r * (H // R) * W * 4 : (r + 1) * (H // R) * W * 4].reshape(H // R, W, 4)
        assert np.array_equal(got, frames[r, r::R])
    out = torch.full((H * W * 4,), -1.0, dtype=torch.float32, device="cuda")
    assert L.tyr_dist_scatter_rows(slabs.data_ptr(), out.data_ptr(), W, H, R, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(H, W, 4), frames.sum(axis=0))


@pytest.mark.gpu
def test_one_rank_communicator_end_to_end(hip):
    """ncclGetUniqueId / ncclCommInitRank / ncclReduce through the library on this GPU (one rank), around a real render:
    both combine modes leave the one-GPU accumulation buffer in frame_out; combines are asynchronous and double-buffered
    (two in flight before the wait), and a destroyed communicator leaves the ctx usable"""
    import torch

    W, H, N, spp = 160, 96, 8192, 2
    sc, nodes, prims = built_scene("cornell36")
    g = hip.Renderer(W, H, N)
    g.load_scene(sc, nodes, prims)
    uid = hip.dist_unique_id()
    assert len(uid) == 128 and any(uid)
    d = hip.Dist(g, uid, 0, 1)
    frame = torch.full((W * H * 4,), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for mode in (hip.TYR_DIST_GATHER, hip.TYR_DIST_REDUCE, hip.TYR_DIST_GATHER):
        g.reset_accum()
        g.render(spp)
        expect = g.blit_buffer()
        d.combine(frame.data_ptr(), mode=mode, root=0)
        g.reset_accum()  # the blit buffer is the renderer's again as soon as combine returns (gather) / once the reduce has read it
        d.wait()
        got = frame.cpu().numpy().reshape(-1, 4)
        assert np.all(expect[:, 3] == spp) and np.array_equal(got, expect), mode
        frame.fill_(-1.0)
        torch.cuda.synchronize()
    with pytest.raises(hip.TyrError):
        hip.Dist(g, uid, 1, 2)  # the communicator's rank must be the ctx's pixel shard
    d.close()
    g.reset_accum()
    assert g.render(1) > 0 and g.counters()["device_error"] == 0


@pytest.mark.gpu
def test_two_host_threads_two_contexts_render_the_two_shards_at_once(orc, hip):
    """The host model INTEGRATION.md describes -- one host thread and one ctx per shard, rendering CONCURRENTLY -- on
    the one device a test box has: the 1 M-triangle scene (BASELINE config C4's) at 1080p dealt to two ranks (rows
    y % 2 == rank), each rendered by its own thread through its own ctx and streams, the rows combined with
    tyr_dist_pack_rows / tyr_dist_scatter_rows as a host would that moves the slabs itself (two devices: hipMemcpyPeer
    in between).  Every rank's counters equal its own oracle run's, the combined frame has exactly spp finished paths in
    every pixel and the oracle's radiance.  (The exchange over RCCL with N > 1 needs N devices: unmeasured on hardware.)
    Also checks what the communicator reports about itself on one rank (ncclCommCount)."""
    import threading

    import torch

    W, H, R, spp = 1920, 1080, 2, 1
    sc, nodes, prims = built_scene("mesh706")
    N = W * (H // R) * spp
    L = hip.lib()
    accum = [torch.zeros(W * H * 4, dtype=torch.float32, device="cuda") for _ in range(R)]
    torch.cuda.synchronize()
    rs = [hip.Renderer(W, H, N, rank=r, nranks=R, flags=1, blit_buffer=accum[r].data_ptr()) for r in range(R)]
    for g in rs:
        g.load_scene(sc, nodes, prims)
    iters, errors = [0] * R, []

    def work(r):
        try:
            iters[r] = rs[r].render(spp)
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    ts = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errors and all(not t.is_alive() for t in ts), errors
    slab_bytes = (H // R) * W * 16
    slabs = torch.zeros(R * (H // R) * W * 4, dtype=torch.float32, device="cuda")
    frame = torch.full((H * W * 4,), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for r in range(R):
        assert L.tyr_dist_pack_rows(accum[r].data_ptr(), slabs.data_ptr() + r * slab_bytes, W, H, r, R, None) == 0
    assert L.tyr_dist_scatter_rows(slabs.data_ptr(), frame.data_ptr(), W, H, R, None) == 0
    torch.cuda.synchronize()
    got = frame.cpu().numpy().reshape(H, W, 4)
    assert np.all(got[:, :, 3] == spp)
    for r in range(R):
        o = orc.Oracle(W, H, N, rank=r, nranks=R, flags=1)
        o.load_scene(sc, nodes, prims)
        assert o.render(spp) == iters[r]
        ko, kg = o.counters(), rs[r].counters()
        assert kg["device_error"] == 0
        for f in ("total_primary_rays", "total_extend_rays", "total_shadow_rays", "n_survive", "n_shadow_visible"):
            assert ko[f] == kg[f], (r, f)
        want = o.blit_buffer().reshape(H, W, 4)[r::R]
        assert np.allclose(got[r::R, :, :3], want[:, :, :3], rtol=1e-5, atol=1e-6), r
        o.close()
    # one-rank communicator: what RCCL reports about itself
    uid = hip.dist_unique_id()
    g1 = hip.Renderer(64, 64, 4096)
    d = hip.Dist(g1, uid, 0, 1)
    info = d.info()
    assert info["rank"] == 0 and info["comm_ranks"] in (1, -1)
    g1.close()  # closes its communicator first
    for g in rs:
        g.close()
