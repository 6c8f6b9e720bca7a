#!/usr/bin/env python3
"""Development tool: cost of rebuilding the animated model's grid every frame (rt_update_vertices + rt_build_grid),
host wall time per call and which build path each frame took. Run under `rocprofv3 --kernel-trace --stats` for the
per-kernel times."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import workloads as wl  # noqa: E402

xyz9 = np.load(os.path.join(ROOT, "tests", "golden", "phoenix_tris.npy"))
core = rt.TraceCore(0)
core.set_scene(None, rt.triangles_from_xyz9(xyz9))
core.build_grid()
frames = [torch.from_numpy(wl.rotate_about_y(xyz9, f)).cuda() for f in range(120)]
torch.cuda.synchronize()
for rep in range(2):
    paths, up, build = [], [], []
    for v in frames:
        t0 = time.perf_counter()
        core.update_vertices(v)  # device to device: no PCIe in the loop
        t1 = time.perf_counter()
        core.build_grid()
        t2 = time.perf_counter()
        up.append((t1 - t0) * 1e3)
        build.append((t2 - t1) * 1e3)
        paths.append(core.build_path())
print(f"paths: {paths.count(1)} device-side, {paths.count(0)} general; update {np.median(up):.3f} ms, build median {np.median(build):.3f} ms, "
      f"min {np.min(build):.3f} ms, device-side builds median {np.median([b for b, p in zip(build, paths) if p == 1] or [0]):.3f} ms")
core.close()
