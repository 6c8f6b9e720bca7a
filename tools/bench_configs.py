#!/usr/bin/env python3
"""Measurements of the workload configurations other than the headline frame (which bench.py covers):
the criterion micro-bench scenes, the massive_benchmark scene at 3840x2160, the animated model with a
per-frame grid rebuild, and the reference's own ray set. Run on the GPU box:
    python tools/bench_configs.py [micro] [massive] [animated] [rays]  > gpurun_out/configs.json
One JSON object per line; progress goes to stderr."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import ray_tracer_amd as rt
from ray_tracer_amd import workloads as wl


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def emit(**kw):
    print(json.dumps(kw), flush=True)


def timed(core, fn, reps):
    fn()
    core.sync()
    core.timer_begin()
    for _ in range(reps):
        fn()
    return core.timer_end() / reps  # ms


def micro(core):
    """benches/bounding_volumes.rs:147-179: rays x particle boxes, every pair; and the same with triangles."""
    dev = "cuda"
    for n_prims in (100, 1000, 10000):
        n_rays = min(1 << 20, (1 << 28) // n_prims)
        d = torch.from_numpy(np.resize(wl.random_unit_vectors(100), (n_rays, 3)).copy()).to(dev)
        o = torch.zeros_like(d)
        mn, mx = (torch.from_numpy(a).to(dev) for a in wl.particle_aabbs(n_prims))
        out = torch.empty((n_rays, n_prims), dtype=torch.float32, device=dev)
        ms = timed(core, lambda: core.intersect_aabb_pairs(o, d, mn, mx, out=out), 5)
        hits = int((out >= 0).sum())
        emit(config="micro/aabb_pairs", n_rays=n_rays, n_boxes=n_prims, ms=ms, Gpairs_per_s=n_rays * n_prims / ms / 1e6,
             hit_fraction=hits / (n_rays * n_prims), out_GBps=n_rays * n_prims * 4 / ms / 1e6)
        t = wl.generate_test_triangles(n_prims)
        v = torch.from_numpy(np.concatenate([t["v0"], t["v1"], t["v2"]], 1).copy()).to(dev)
        ro, rd = wl.generate_test_rays(n_rays)
        ro, rd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
        ms = timed(core, lambda: core.intersect_tri_pairs(ro, rd, v, out=out), 5)
        emit(config="micro/tri_pairs", n_rays=n_rays, n_tris=n_prims, ms=ms, Gpairs_per_s=n_rays * n_prims / ms / 1e6,
             hit_fraction=int((out >= 0).sum()) / (n_rays * n_prims), out_GBps=n_rays * n_prims * 4 / ms / 1e6)


def frame_stats(core, cam, w, h):
    core.set_collect_stats(True)
    core.trace_primary(cam, w, h, out_primary=P[: w * h], out_shadow=S[: w * h])
    core.sync()
    st = core.stats()
    core.set_collect_stats(False)
    return st


def massive(core, sizes):
    """examples/massive_benchmark.rs scene (src/core/benchmark.rs:99-123) at 3840x2160, primary + shadow."""
    w, h = 3840, 2160
    cam = rt.Camera((0, 0, 150), math.pi, 0.0).to_uniform(0.0, float(h))
    for n in sizes:
        tris = wl.generate_test_triangles(n)
        t0 = time.perf_counter()
        core.set_scene(None, tris)
        core.sync()
        upload_ms = (time.perf_counter() - t0) * 1e3
        for label, cell in (("reference grid: cell 16, cap 8192", 16.0), ("tuned grid: cell 2", 2.0)):
            t0 = time.perf_counter()
            meta, n_cells, n_refs = core.build_grid(cell_size=cell)
            build_ms = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            core.build_grid(cell_size=cell)
            build2_ms = (time.perf_counter() - t0) * 1e3
            st = frame_stats(core, cam, w, h)
            reps = 1 if st["tri_tests"] > 2e11 else (3 if st["tri_tests"] > 2e10 else 10)
            ms = timed(core, lambda: core.trace_primary(cam, w, h, out_primary=P[: w * h], out_shadow=S[: w * h]), reps)
            alg = 8 * st["cells_visited"] + 40 * st["tri_tests"] + 8 * st["rays"]
            emit(config="massive_benchmark 3840x2160 primary+shadow", triangles=n, grid=label, dims=list(meta.fine_dims),
                 cells=n_cells, refs=n_refs, trace_refs=core.trace_refs(), upload_ms=upload_ms, build_ms_first=build_ms, build_ms=build2_ms, trace_ms=ms,
                 rays=st["rays"], primary_hits=st["primary_hits"], shadow_hits=st["shadow_hits"],
                 Mrays_per_s=st["rays"] / ms / 1e3, tri_tests=st["tri_tests"], Gtests_per_s=st["tri_tests"] / ms / 1e6,
                 algorithmic_GBps=alg / ms / 1e6)
            log(f"massive n={n} {label}: build {build2_ms:.2f} ms trace {ms:.3f} ms {st['rays'] / ms / 1e3:.0f} Mrays/s")


def unique(core, sizes):
    """Scenes too large for the caches: n distinct small random triangles in the massive scene's volume (no copies,
    so nothing is left out), grid cells sized for about eight references each, 3840x2160 primary + shadow."""
    w, h = 3840, 2160
    cam = rt.Camera((0, 0, 150), math.pi, 0.0).to_uniform(0.0, float(h))
    rng = np.random.default_rng(1)
    for n in sizes:
        size = 100.0 / n ** (1.0 / 3.0)  # mean spacing
        c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
        v = (c + rng.normal(0, 0.6 * size, (n, 3, 3)).astype(np.float32)).astype(np.float32)
        tris = rt.triangles_from_xyz9(v.reshape(-1, 9))
        del c, v
        core.set_scene(None, tris)
        cell = 2.0 * size
        t0 = time.perf_counter()
        meta, n_cells, n_refs = core.build_grid(cell_size=cell)
        build_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        core.build_grid(cell_size=cell)
        build2_ms = (time.perf_counter() - t0) * 1e3
        p = rt.default_params(max_dda_steps=4096)
        st = frame_stats(core, cam, w, h)
        ms = timed(core, lambda: core.trace_primary(cam, w, h, out_primary=P[: w * h], out_shadow=S[: w * h]), 5)
        alg = 8 * st["cells_visited"] + 40 * st["tri_tests"] + 8 * st["rays"]
        emit(config="distinct random triangles 3840x2160 primary+shadow", triangles=n, cell_size=cell, dims=list(meta.fine_dims),
             cells=n_cells, refs=n_refs, trace_refs=core.trace_refs(), record_bytes=48 * core.trace_refs(), build_ms_first=build_ms,
             build_ms=build2_ms, trace_ms=ms, rays=st["rays"], primary_hits=st["primary_hits"], Mrays_per_s=st["rays"] / ms / 1e3,
             cells_visited=st["cells_visited"], tri_tests=st["tri_tests"], algorithmic_GBps=alg / ms / 1e6)
        log(f"unique n={n}: build {build2_ms:.2f} ms trace {ms:.3f} ms {st['rays'] / ms / 1e3:.0f} Mrays/s")


def rays(core, sizes):
    """The example's own measurement: generate_test_rays through the structure (examples/massive_benchmark.rs:90-98)."""
    for n in sizes:
        core.set_scene(None, wl.generate_test_triangles(n))
        for n_rays in (50000, 100000):
            ro, rd = wl.generate_test_rays(n_rays)
            ro, rd = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
            out = torch.empty((n_rays, 2), dtype=torch.int32, device="cuda")
            for label, cell in (("cell 16", 16.0), ("cell 2", 2.0)):
                core.build_grid(cell_size=cell)
                ms = timed(core, lambda: core.trace_rays(ro, rd, out=out, n=n_rays), 5)
                hits = int((out[:, 0] != -1).sum())
                emit(config="massive_benchmark ray set", triangles=n, rays=n_rays, grid=label, trace_ms=ms,
                     Mrays_per_s=n_rays / ms / 1e3, hits=hits)


def animated(core):
    """Bundled animated model (bind pose fixture), synthetic rigid motion, grid rebuilt on the GPU every frame."""
    w, h = 1920, 1080
    xyz9 = np.load(os.path.join(ROOT, "tests", "golden", "phoenix_tris.npy"))
    core.set_scene(None, rt.triangles_from_xyz9(xyz9))
    cam = rt.Camera((-300, 250, 600), 2.7, -0.2).to_uniform(0.0, float(h))
    frames = [wl.rotate_about_y(xyz9, f) for f in range(120)]
    up, build, trace, nrays = [], [], [], []
    for f, v in enumerate(frames):
        t0 = time.perf_counter()
        core.update_vertices(v)
        core.sync()
        t1 = time.perf_counter()
        core.build_grid()
        t2 = time.perf_counter()
        core.timer_begin()
        core.trace_primary(cam, w, h, out_primary=P[: w * h], out_shadow=S[: w * h])
        trace.append(core.timer_end())
        up.append((t1 - t0) * 1e3)
        build.append((t2 - t1) * 1e3)
        if f % 30 == 0:
            nrays.append(w * h + int((P[: w * h, 0] != -1).sum()))
    med = lambda a: float(np.median(a[5:]))
    emit(config="animated model 1920x1080 primary+shadow, per-frame rebuild", triangles=len(xyz9), frames=len(frames),
         upload_ms=med(up), build_ms=med(build), trace_ms=med(trace), frame_ms=med(up) + med(build) + med(trace),
         rays_sampled=nrays, Mrays_per_s_trace_only=float(np.mean(nrays)) / med(trace) / 1e3,
         Mrays_per_s_with_rebuild=float(np.mean(nrays)) / (med(up) + med(build) + med(trace)) / 1e3)


def animated_pipelined(n_ctx=2):
    """Same frames, two contexts used alternately (each owns a copy of the scene, its grid and a stream): while
    one context's trace kernel drains its long tail, the other uploads and bins the next frame."""
    w, h = 1920, 1080
    xyz9 = np.load(os.path.join(ROOT, "tests", "golden", "phoenix_tris.npy"))
    cores = []
    for _ in range(n_ctx):
        c = rt.TraceCore(0)
        c.set_scene(None, rt.triangles_from_xyz9(xyz9))
        cores.append(c)
    outs = [(torch.empty((w * h, 2), dtype=torch.int32, device="cuda"), torch.empty((w * h, 2), dtype=torch.int32, device="cuda"))
            for _ in range(n_ctx)]
    cam = rt.Camera((-300, 250, 600), 2.7, -0.2).to_uniform(0.0, float(h))
    frames = [wl.rotate_about_y(xyz9, f) for f in range(120)]

    def run():
        for f, v in enumerate(frames):
            c, (p, s) = cores[f % n_ctx], outs[f % n_ctx]
            c.update_vertices(v)
            c.build_grid()
            c.trace_primary(cam, w, h, out_primary=p, out_shadow=s)
        torch.cuda.synchronize()

    run()
    t0 = time.perf_counter()
    run()
    ms = (time.perf_counter() - t0) * 1e3 / len(frames)
    rays = w * h + int((outs[(len(frames) - 1) % n_ctx][0][:, 0] != -1).sum())
    emit(config="animated model 1920x1080 primary+shadow, per-frame rebuild, contexts used alternately", contexts=n_ctx,
         triangles=len(xyz9), frames=len(frames), frame_ms=ms, Mrays_per_s_with_rebuild=rays / ms / 1e3)
    for c in cores:
        c.close()


def bvh(core):
    """The structure the reference's own benchmark times (BVH build + both-children traversal), on the GPU, beside
    the grid walk on the same rays."""
    cases = []
    car = np.load(os.path.join(ROOT, "tests", "golden", "car_tris.npy"))
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, 1080.0)
    cases.append(("bundled static model, the 1920x1080 frame's primary rays", rt.triangles_from_xyz9(car),
                  wl.primary_rays(cam, 1920, 1080), 16.0))
    for n in (100000, 1000000):
        cases.append((f"massive_benchmark scene, {n} triangles, 2,000,000 generate_test_rays", wl.generate_test_triangles(n),
                      wl.generate_test_rays(2000000), 2.0))
    for label, tris, (o, d), cell in cases:
        core.set_scene(None, tris)
        t0 = time.perf_counter()
        st = core.build_bvh()
        build_ms = (time.perf_counter() - t0) * 1e3
        to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
        out = torch.empty(len(o), dtype=torch.float32, device="cuda")
        ms = timed(core, lambda: core.trace_rays_bvh(to, td, out=out), 5)
        core.build_grid(cell_size=cell)
        hits = torch.empty((len(o), 2), dtype=torch.int32, device="cuda")
        gms = timed(core, lambda: core.trace_rays(to, td, out=hits, n=len(o)), 5)
        emit(config="BVH path: " + label, triangles=len(tris), rays=len(o), bvh_nodes=st["node_count"], bvh_depth=st["max_depth"],
             bvh_build_ms_host=build_ms, bvh_trace_ms=ms, bvh_Mrays_per_s=len(o) / ms / 1e3, bvh_hits=int((out >= 0).sum()),
             grid_cell=cell, grid_trace_ms=gms, grid_Mrays_per_s=len(o) / gms / 1e3, grid_hits=int((hits[:, 0] != -1).sum()))


def animated_skinned(n_ctx=2, heavy=0):
    """The bundled animated model with its own skeleton animation: per frame 89 joint matrices go to the GPU
    (5.7 KB), the skinning kernel writes the vertex array, the binning pass reads it, the frame is traced.
    heavy: rt_params.heavy_block_threshold (microseconds; 0 = no plan of expensive pixel blocks)."""
    from ray_tracer_amd.animation import SkinnedModel
    w, h = 1920, 1080
    model = SkinnedModel(os.path.join(ROOT, "tests", "golden", "phoenix_skin.npz"))
    mats = [model.joint_matrices(model.duration * f / 120) for f in range(120)]
    cores = []
    for _ in range(n_ctx):
        c = rt.TraceCore(0, rt.default_params(heavy_block_threshold=heavy))
        c.set_scene(None, rt.triangles_from_xyz9(model.bind_pose_triangles()))
        c.set_skin(model.positions, model.joints, model.weights, model.tri_indices)
        cores.append(c)
    outs = [(torch.empty((w * h, 2), dtype=torch.int32, device="cuda"), torch.empty((w * h, 2), dtype=torch.int32, device="cuda"))
            for _ in range(n_ctx)]
    cam = rt.Camera((-300, 250, 600), 2.7, -0.2).to_uniform(0.0, float(h))
    hits = []

    def run(count):
        for f, jm in enumerate(mats):
            c, (p, s) = cores[f % n_ctx], outs[f % n_ctx]
            c.skin_vertices(jm)
            c.build_grid()
            c.trace_primary(cam, w, h, out_primary=p, out_shadow=s)
            if count and f % 30 == 0:
                hits.append(w * h + int((p[:, 0] != -1).sum()))
        torch.cuda.synchronize()

    run(True)
    t0 = time.perf_counter()
    run(False)
    ms = (time.perf_counter() - t0) * 1e3 / len(mats)
    emit(config="animated model, skeleton animation skinned on the GPU, 1920x1080 primary+shadow, per-frame rebuild",
         contexts=n_ctx, heavy_block_threshold_us=heavy, plan_heavy_early_threshold_us=cores[0].heavy_plan(),
         triangles=len(model.tri_indices), joints=len(mats[0]), frames=len(mats), frame_ms=ms,
         rays_sampled=hits, Mrays_per_s_with_skinning_and_rebuild=float(np.mean(hits)) / ms / 1e3)
    for c in cores:
        c.close()


if __name__ == "__main__":
    what = set(sys.argv[1:]) or {"micro", "massive", "animated", "rays", "bvh"}
    big = "big" in what
    core = rt.TraceCore(0)
    P = torch.empty((3840 * 2160, 2), dtype=torch.int32, device="cuda")
    S = torch.empty_like(P)
    if "micro" in what:
        micro(core)
    if "animated" in what:
        animated(core)
        animated_pipelined(2)
        for heavy in [int(v) for v in os.environ.get("RT_HEAVY_LIST", "0,64").split(",")]:
            animated_skinned(1, heavy)
            animated_skinned(2, heavy)
    if "bvh" in what:
        bvh(core)
    if "rays" in what:
        rays(core, (100000, 1000000))
    if "unique" in what:
        unique(core, (1000000, 10000000, 40000000) if big else (1000000, 10000000))
    if "massive" in what:
        massive(core, (10000000,) if big else (100000, 1000000))
    core.close()
