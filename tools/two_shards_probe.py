#!/usr/bin/env python3
"""tools/two_shards_probe.py -- can a SECOND shard's feed hide the first one's drain?  (round 6)

46 % of every traversal launch is its drain: the device runs a thinning set of waves and nothing else of the render may start, because
every kernel of a render depends on the one before.  Two SHARDS of the frame do not depend on each other: DESIGN.md section 6's row sharding
(rank r renders the rows y % R == r with its own queue; each rank's result equals the oracle's for that shard, tests/test_gpu_configs.py C4)
run as R contexts on ONE device, each on its own stream from its own host thread, would let one shard's feed fill the CUs the other's drain
leaves idle.  Same pixels, same samples per pixel; the seeds are the sharded job's (a different, equally valid serial order).

    python tools/two_shards_probe.py [workload=c3] [R=2] [rounds=20] [queue_total=0 (8 W H)]

Prints ms per whole frame: one context rendering the whole frame, R contexts one after the other (what sharding costs by itself), and R
contexts at once.
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tyrant_amd import binding, scenes  # noqa: E402
from tyrant_amd.benchkit.common import build_workload  # noqa: E402

W, H, SPP = 1920, 1080, 8


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    qtot = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    sc, nodes, prims, label, _ = build_workload(wl, binding, scenes)
    flags = binding.TYR_FLAG_TRIANGLE_MATERIALS if sc.triangle_materials else 0
    Ntot = qtot if qtot else SPP * W * H

    def make(rank, nranks):
        r = binding.Renderer(W, H, Ntot // nranks, flags=flags, rank=rank, nranks=nranks)
        r.load_scene(sc, nodes, prims)
        return r

    def render(r):
        r.set_frame(1)
        r.reset_accum()
        r.render(SPP)

    whole = make(0, 1)
    shards = [make(k, R) for k in range(R)]
    for _ in range(3):
        render(whole)
        for s in shards:
            render(s)
    ref = [s.blit_buffer() for s in shards]

    def timed(fn):
        ts = []
        for _ in range(rounds):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return ts[len(ts) // 2], ts[0]

    def serial():
        for s in shards:
            render(s)

    go = [threading.Barrier(R + 1), threading.Barrier(R + 1)]
    stop = [False]

    def worker(s):
        while True:
            go[0].wait()
            if stop[0]:
                return
            render(s)
            go[1].wait()

    th = [threading.Thread(target=worker, args=(s,)) for s in shards]
    for t in th:
        t.start()

    def together():
        go[0].wait()
        go[1].wait()

    a = timed(lambda: render(whole))
    b = timed(serial)
    c = timed(together)
    a2 = timed(lambda: render(whole))
    stop[0] = True
    go[0].wait()
    for t in th:
        t.join()
    same = all(np.array_equal(s.blit_buffer()[:, 3], x[:, 3]) and np.allclose(s.blit_buffer()[:, :3], x[:, :3], rtol=1e-5, atol=1e-6) for s, x in zip(shards, ref))
    k = [s.counters() for s in shards]
    rays = sum(x["total_extend_rays"] + x["total_shadow_rays"] for x in k)
    print(f"{label}\nqueue {Ntot} slots in all, {rounds} rounds, ms per whole frame: median (min)")
    print(f"  one context, the whole frame          {a[0]:7.3f} ({a[1]:.3f})   again {a2[0]:7.3f} ({a2[1]:.3f})")
    print(f"  {R} shards, one after the other         {b[0]:7.3f} ({b[1]:.3f})")
    print(f"  {R} shards at once (own streams/threads) {c[0]:7.3f} ({c[1]:.3f})   = {a[0] / c[0]:.3f} x one context; shards' pictures as rendered alone: {same}")


if __name__ == "__main__":
    main()
