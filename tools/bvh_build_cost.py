#!/usr/bin/env python3
"""Development tool: rt_build_bvh (GPU, level by level) against rt_build_bvh_host (the reference's recursion, one host
thread), both from vertices already on the device; the trees are compared node for node on the way."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import workloads  # noqa: E402


def soup(n, seed=1):
    rng = np.random.default_rng(seed)
    size = 100.0 / n ** (1.0 / 3.0)
    c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
    return rt.triangles_from_xyz9((c + rng.normal(0, 0.6 * size, (n, 3, 3)).astype(np.float32)).astype(np.float32).reshape(n, 9))


core = rt.TraceCore(0)
scenes = [("static model (3 132 triangles)", lambda: rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/car_tris.npy")))),
          ("animated model (4 064 triangles)", lambda: rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/phoenix_tris.npy")))),
          ("benchmark scene, 100 000 triangles (10 copies of each)", lambda: workloads.generate_test_triangles(100000)),
          ("100 000 distinct random triangles", lambda: soup(100000)),
          ("1 000 000 distinct random triangles", lambda: soup(1000000)),
          ("10 000 000 distinct random triangles", lambda: soup(10000000))]
for name, make in scenes:
    if "10 000 000" in name and os.environ.get("RT_BIG") != "1":
        continue
    core.set_scene(None, make())
    core.sync()
    times = {}
    trees = {}
    for on_host in (False, True):
        reps = 1 if on_host else 3
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter()
            st = core.build_bvh(on_host=on_host)
            best = min(best, time.perf_counter() - t0)
        times[on_host] = best
        trees[on_host] = core.read_bvh()
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(trees[False], trees[True]))
    print(f"{name}: {st['node_count']} nodes, depth {st['max_depth']}: GPU {times[False] * 1e3:9.2f} ms, host {times[True] * 1e3:10.2f} ms "
          f"({times[True] / times[False]:.1f} x), trees {'identical' if same else 'DIFFER'}", flush=True)
