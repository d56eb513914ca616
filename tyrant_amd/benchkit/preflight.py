"""the pre-flight check of the native multi-GPU exchange (tyr_dist_*: RCCL behind the C ABI): a child of every bench rank, run before the rank
touches its GPU"""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile

from .common import BENCH

PREFLIGHT_TIMEOUT_S = 150.0
PREFLIGHT_TORCH_NCCL_ONLY = 2  # exit code of the pre-flight child: the native exchange failed, torch's nccl backend works


def dist_preflight(args) -> int:
    """One rank of the pre-flight check of the native exchange (tyr_dist_*: RCCL behind the C ABI), run as a CHILD of the
    bench rank of the same number before that rank has touched its GPU: a small frame, rows dealt y % world == rank, a
    2-spp render per rank, GATHER and REDUCE onto rank 0, every pixel must hold exactly 2 finished paths.  The exchange
    has only met one GPU per box before the driver's multi-GPU run; a hang, a crash or a wrong frame here costs this child,
    not the measurement: the bench ranks then use torch.distributed for the combine.  Exit code 0 = verified on every rank."""
    import datetime

    import torch
    import torch.distributed as dist

    from tyrant_amd import binding, scenes

    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("TYR_BENCH_PREFLIGHT_ONE_DEVICE"):
        local_rank = 0  # rehearsal on a one-GPU box: RCCL refuses two ranks on one device, which is the failure path under test
    local_rank %= max(torch.cuda.device_count(), 1)  # (fewer GPUs than ranks: the same failure path, not an invalid-device crash)
    dist.init_process_group("gloo", init_method=f"file://{args.preflight_store}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    ok = 1
    try:
        torch.cuda.set_device(local_rank)
        W, H, spp = 64, 8 * world, 2
        sc = scenes.cornell_box()
        nodes, prims = binding.bvh_build(sc.triangles)
        r = binding.Renderer(W, H, 4096, device=local_rank, rank=rank, nranks=world)
        r.load_scene(sc, nodes, prims)
        ids = [binding.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = binding.Dist(r, ids[0], rank, world)
        frame = torch.zeros(H * W * 4, dtype=torch.float32, device=f"cuda:{local_rank}") if rank == 0 else None
        torch.cuda.synchronize()
        for mode in (binding.TYR_DIST_GATHER, binding.TYR_DIST_REDUCE):
            r.reset_accum()
            r.render(spp)
            comm.combine(frame.data_ptr() if frame is not None else None, mode=mode, root=0)
            comm.wait()
            torch.cuda.synchronize()
            if rank == 0:
                a = frame.view(H * W, 4)[:, 3]
                if not (float(a.min()) == float(a.max()) == float(spp)):
                    print(f"[bench preflight] mode {mode}: combined frame holds {float(a.min())}..{float(a.max())} paths per pixel, expected {spp}", file=sys.stderr)
                    ok = 0
                frame.zero_()
        comm.close()
        r.close()
    except Exception as e:  # noqa: BLE001
        print(f"[bench preflight] rank {rank}: {e!r}", file=sys.stderr)
        ok = 0
    # ... and torch's own RCCL backend (what the combine falls back to when the native exchange does not verify): one
    # all-reduce on this rank's device.  If that fails too, the bench ranks combine over gloo, host-staged.
    nccl_ok = 1
    if not ok or os.environ.get("TYR_BENCH_PREFLIGHT_PROBE_TORCH_NCCL"):
        try:
            g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=45))
            t = torch.ones(4, dtype=torch.float32, device=f"cuda:{local_rank}")
            dist.all_reduce(t, group=g)
            torch.cuda.synchronize()
            nccl_ok = int(float(t[0].item()) == float(world))
        except Exception as e:  # noqa: BLE001
            print(f"[bench preflight] rank {rank}: torch's nccl backend: {e!r}", file=sys.stderr)
            nccl_ok = 0
    flag = torch.tensor([ok, nccl_ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    return 0 if int(flag[0].item()) == 1 else (PREFLIGHT_TORCH_NCCL_ONLY if int(flag[1].item()) == 1 else 1)


def run_dist_preflight() -> int:
    """spawn this rank's pre-flight child (this process has not initialised the GPU yet) and wait for it, bounded:
    0 = the native exchange verified on every rank; PREFLIGHT_TORCH_NCCL_ONLY = it did not, torch's nccl backend does;
    1 = neither (or the child crashed / ran out of time)"""
    # the children make their own rendezvous through a FILE (no second port to find free and to agree on): one name per
    # launch -- the launcher's pid is the parent of every rank, its master port tells concurrent launches apart -- and
    # without the launcher's TORCHELASTIC_* variables, which would tell them that an agent already hosts a store
    store = os.path.join(tempfile.gettempdir(), f"tyr_preflight_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}")
    cmd = [sys.executable, BENCH, "--dist-preflight", "--preflight-store", store]
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    try:
        p = subprocess.run(cmd, timeout=PREFLIGHT_TIMEOUT_S, env=env, stdout=subprocess.DEVNULL)
        return p.returncode if p.returncode in (0, PREFLIGHT_TORCH_NCCL_ONLY) else 1
    except subprocess.TimeoutExpired:
        print(f"[bench] rank {os.environ.get('RANK', '?')}: the native exchange's pre-flight did not finish in {PREFLIGHT_TIMEOUT_S:.0f} s", file=sys.stderr)
        return 1
