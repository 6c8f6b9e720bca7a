"""Development aid: one scene of n distinct random triangles, grid built three times (run under rocprofv3 --kernel-trace
--stats to see where the binning pass spends its time)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_amd as rt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
rng = np.random.default_rng(1)
size = 100.0 / n ** (1.0 / 3.0)
c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
v = (c + rng.normal(0, 0.6 * size, (n, 3, 3)).astype(np.float32)).astype(np.float32)
core = rt.TraceCore(0)
core.set_scene(None, rt.triangles_from_xyz9(v.reshape(-1, 9)))
for k in range(3):
    t0 = time.perf_counter()
    meta, n_cells, n_refs = core.build_grid(cell_size=2.0 * size)
    print(f"build {k}: {(time.perf_counter() - t0) * 1e3:.2f} ms, {n_cells} cells, {n_refs} refs", flush=True)
core.close()
