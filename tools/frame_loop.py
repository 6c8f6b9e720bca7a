#!/usr/bin/env python3
"""Development tool: traces one scene's primary + shadow frame a number of times (for rocprofv3 runs and quick timing).
    python tools/frame_loop.py <scene> [frames]
scenes: car | phoenix | stream[:n[:cell factor]] | massive[:n[:skip]] (camera inside the volume, 3840x2160;
        skip = 0 tests every list entry) | massive8k[:n] (7680x4320, camera inside)"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import workloads  # noqa: E402


def setup(spec):
    parts = spec.split(":")
    name = parts[0]
    params = rt.default_params()
    cell = 0.0
    if name == "car":
        w, h = 1920, 1080
        tris = rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/car_tris.npy")))
        cam = rt.Camera.for_scene("gltf")
    elif name == "phoenix":
        w, h = 1920, 1080
        tris = rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/phoenix_tris.npy")))
        cam = rt.Camera((-300, 250, 600), 2.7, -0.2)
    elif name == "stream":
        w, h = 3840, 2160
        n = int(parts[1]) if len(parts) > 1 else 10000000
        factor = float(parts[2]) if len(parts) > 2 else 1.0
        rng = np.random.default_rng(1)
        size = 100.0 / n ** (1.0 / 3.0)
        c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
        v = (c + rng.normal(0, 0.6 * size, (n, 3, 3)).astype(np.float32)).astype(np.float32)
        tris = rt.triangles_from_xyz9(v.reshape(-1, 9))
        cell = factor * size
        cam = rt.Camera((0, 0, 75), math.pi, 0.0)
    elif name in ("massive", "massive8k"):
        w, h = (7680, 4320) if name == "massive8k" else (3840, 2160)
        n = int(parts[1]) if len(parts) > 1 else 100000
        if len(parts) > 2:
            params = rt.default_params(skip_duplicate_triangles=int(parts[2]))
        tris = workloads.generate_test_triangles(n)
        cam = rt.Camera((3.0, -2.0, 49.0), math.pi + 0.2, -0.1)
    else:
        raise SystemExit(f"unknown scene {spec}")
    params.repack_candidates = int(os.environ.get("RT_REPACK", "0"))  # rt_params.repack_candidates
    core = rt.TraceCore(0, params)
    core.set_scene(None, tris)
    core.build_grid(cell_size=cell)
    return core, cam.to_uniform(0.0, float(h)), w, h


if __name__ == "__main__":
    core, cam, w, h = setup(sys.argv[1])
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    p = torch.empty((w * h, 2), dtype=torch.int32, device="cuda")
    s = torch.empty_like(p)
    for _ in range(2):
        core.trace_primary(cam, w, h, out_primary=p, out_shadow=s)
    core.sync()
    core.timer_begin()
    for _ in range(frames):
        core.trace_primary(cam, w, h, out_primary=p, out_shadow=s)
    ms = core.timer_end() / frames
    hits = int((p[:, 0] != -1).sum())
    print(f"{sys.argv[1]}: {ms * 1e3:.1f} us per frame, {(w * h + hits) / ms / 1e6:.2f} Grays/s, hit fraction {hits / (w * h):.3f}, "
          f"trace refs {core.trace_refs()}")
    if os.environ.get("RT_COUNT"):
        core.set_collect_stats(2)
        core.trace_primary(cam, w, h, out_primary=p, out_shadow=s)
        st = core.stats()
        print({k: st[k] for k in ("rays", "cell_records_fetched", "cell_records_uniform", "ref_records_fetched", "ref_records_staged")},
              f"uniform share of cell fetches {st['cell_records_uniform'] / max(1, st['cell_records_fetched']):.3f}")
    core.close()
