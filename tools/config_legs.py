#!/usr/bin/env python3
"""The BASELINE configurations other than the headline frame, each as one bounded, timed leg.

bench.py runs them after its headline measurement and puts the results on its JSON line as `configs`; run alone
(`python tools/config_legs.py <leg> [repeats]`) a leg just loops its timed launches, which is the command the counter
passes of profiles/collect_cmd.sh profile (its kernels are then the only large ones in the trace).

  criterion      configs[0]  the criterion particle scene (benches/bounding_volumes.rs:147-179), all (ray, box) pairs
  massive4k      configs[2]  examples/massive_benchmark.rs scene (src/core/benchmark.rs:99-123), 3840x2160 primary +
                             shadow, the reference's cell size 16, 100 k and 1 M triangles, exact duplicate triangles
                             left out of the trace records (the default) and kept (every list entry tested)
  animated       configs[3]  the bundled animated model driven by its own skeleton: per frame skinning on the GPU,
                             grid rebuild (binning) on the GPU, 1920x1080 primary + shadow; the three parts timed
                             separately in one context, and the whole frame with two contexts used alternately
  massive8k      configs[4]  on ONE GPU: the massive_benchmark scene, 1 M triangles, 7680x4320 primary + shadow
Every time is HIP-event time on the launch stream (rt_timer_begin / rt_timer_end) unless a field says wall."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 2  # 1 024 SIMDs, one 64-wide vector instruction per 2 cycles each, 2.4 GHz
HBM_PEAK_GBS = 8000.0


def _timed(core, fn, reps):
    fn()
    core.sync()
    core.timer_begin()
    for _ in range(reps):
        fn()
    return core.timer_end() / reps  # ms


def _timed_in_flight(core, frame, streams, reps):
    """ms per frame with len(streams) frames in flight, as the headline leg runs them: frame k goes to stream k mod n
    (and to output buffers of its own); the host's clock around `reps` frames between two waits for all streams."""
    for k in range(len(streams)):
        core.set_stream(streams[k].cuda_stream)
        frame(k)
    core.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        core.set_stream(streams[k % len(streams)].cuda_stream)
        frame(k % len(streams))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    core.sync()
    core.set_stream(None)
    return ms


def _count(core, cam, w, h, outs):
    """The reference algorithm's work on one frame (cells visited, candidates tested), by the counting kernel."""
    core.set_collect_stats(1)
    core.trace_primary(cam, w, h, shadow=True, out_primary=outs[0], out_shadow=outs[1])
    st = core.stats()
    core.set_collect_stats(0)
    return st


def _algorithmic_bytes(st):
    return 8 * st["cells_visited"] + 40 * st["tri_tests"] + 28 * st["box_tests"] + 8 * st["rays"]


def _counters(name):
    """Counter summary of this leg's dominant kernel from the committed pass (profiles/r03_config_counters.json:
    vector instructions and HBM bytes per launch), or None."""
    path = os.path.join(ROOT, "profiles", "r03_config_counters.json")
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get(name)


def _roof(name, ms, alg_bytes):
    """bound / frac of one launch of `ms` milliseconds: vector-instruction issue when a counter pass of this leg is
    committed (these scenes sit in the caches: the algorithmic byte rate is given, never as a fraction), plus the measured
    HBM traffic as a fraction of the HBM peak."""
    out = {"algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBps": alg_bytes / ms / 1e6}
    c = _counters(name)
    if c:
        g = c["SQ_INSTS_VALU_per_launch"] / ms / 1e6
        out.update({"bound": "valu_issue", "achieved": g, "peak": VALU_PEAK_GINSTR, "unit": "Ginstr/s", "frac": g / VALU_PEAK_GINSTR,
                    "traffic": c.get("hbm_bytes_per_launch"), "traffic_source": c.get("source")})
        if c.get("hbm_bytes_per_launch"):
            out["hbm_traffic_frac"] = c["hbm_bytes_per_launch"] / ms / 1e6 / HBM_PEAK_GBS
        if "lanes_active_per_valu_instruction" in c:
            out["lanes_active_per_valu_instruction"] = c["lanes_active_per_valu_instruction"]
    else:
        out.update({"bound": "valu_issue", "achieved": None, "peak": VALU_PEAK_GINSTR, "unit": "Ginstr/s", "frac": None,
                    "traffic": None, "traffic_source": "no counter pass committed for this leg"})
    return out


def criterion(rt, dev, repeats=5):
    from ray_tracer_amd import workloads as wl
    core = rt.TraceCore(dev.index)
    n_prims, n_rays = 10000, 1 << 14
    d = torch.from_numpy(np.resize(wl.random_unit_vectors(100), (n_rays, 3)).copy()).to(dev)
    o = torch.zeros_like(d)
    mn, mx = (torch.from_numpy(a).to(dev) for a in wl.particle_aabbs(n_prims))
    out = torch.empty((n_rays, n_prims), dtype=torch.float32, device=dev)
    ms = _timed(core, lambda: core.intersect_aabb_pairs(o, d, mn, mx, out=out), repeats)
    core.close()
    pairs = n_rays * n_prims
    return [{"config": "configs[0] criterion particle scene, every (ray, box) pair on the GPU", "kernel": "k_aabb_pairs",
             "rays": n_rays, "boxes": n_prims, "ms": ms, "Gpairs_per_s": pairs / ms / 1e6, "bound": "hbm",
             "achieved": pairs * 4 / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pairs * 4 / ms / 1e6 / HBM_PEAK_GBS,
             "traffic": pairs * 4, "traffic_source": "algorithmic: the 4-byte result per pair (inputs are 0.4 MB)",
             "note": "the reference times 100 rays x 10 000 boxes on one CPU core (1.0005 ms, benchmark_results.txt:106); "
                     "cpu_baseline.criterion_100_rays_x_10000_aabbs_ms is that loop on this box"}]


def massive(rt, dev, width, height, sizes, dup_modes, repeats=None, name="massive4k"):
    from ray_tracer_amd import workloads as wl
    cam = rt.Camera((0, 0, 150), math.pi, 0.0).to_uniform(0.0, float(height))
    flight = [(torch.empty((width * height, 2), dtype=torch.int32, device=dev), torch.empty((width * height, 2), dtype=torch.int32, device=dev))
              for _ in range(3)]
    outs = flight[0]
    streams = [rt.shared_stream(dev, k) for k in range(3)]
    legs = []
    for n in sizes:
        tris = wl.generate_test_triangles(n)
        for skip in dup_modes:
            core = rt.TraceCore(dev.index, rt.default_params(skip_duplicate_triangles=skip))
            core.set_scene(None, tris)
            t0 = time.perf_counter()
            meta, n_cells, n_refs = core.build_grid()
            core.sync()
            build_ms = (time.perf_counter() - t0) * 1e3
            st = _count(core, cam, width, height, outs)
            heavy = not skip and n >= 1000000  # tens of milliseconds a frame: one timed launch is enough
            reps = repeats or (1 if heavy else 10)
            one = _timed(core, lambda: core.trace_primary(cam, width, height, shadow=True, out_primary=outs[0], out_shadow=outs[1]), reps)
            ms = min(one, _timed_in_flight(
                core, lambda k: core.trace_primary(cam, width, height, shadow=True, out_primary=flight[k][0], out_shadow=flight[k][1]),
                streams, 3 * reps))
            leg = {"config": f"configs[{2 if height == 2160 else 4}] massive_benchmark scene, {n} triangles, {width}x{height} primary + shadow, "
                             f"reference grid (cell 16), duplicate triangles {'left out' if skip else 'tested'}",
                   "kernel": "k_trace_primary_lean<shadow>", "triangles": n, "skip_duplicate_triangles": skip,
                   "grid": list(meta.fine_dims), "cell_refs": n_refs, "trace_refs": core.trace_refs(), "build_ms_wall_first": build_ms,
                   "ms": ms, "ms_one_frame_at_a_time": one, "frames_in_flight": 1 if ms == one else 3, "rays": st["rays"], "primary_hits": st["primary_hits"], "Grays_per_s": st["rays"] / ms / 1e6,
                   "tri_tests_reference": st["tri_tests"], "Gtests_per_s_reference": st["tri_tests"] / ms / 1e6}
            leg.update(_roof(f"{name}_{n}_{'skip' if skip else 'all'}", ms, _algorithmic_bytes(st)))
            legs.append(leg)
            core.close()
    return legs


def animated(rt, dev, frames=int(os.environ.get("RT_ANIM_FRAMES", "200"))):
    """The bundled animated model's clip (3.33 s) played at 60 frames per second: 200 poses, each skinned, rebuilt and
    traced. (The plan of expensive blocks follows the previous frame: stepping through the clip in 60 poses instead - 18
    frames per second - moves the wings by more than the plan tolerates, 0.26 instead of 0.22 ms per frame in one context.)"""
    from ray_tracer_amd.animation import SkinnedModel
    w, h = 1920, 1080
    model = SkinnedModel(os.path.join(ROOT, "tests", "golden", "phoenix_skin.npz"))
    mats = [torch.from_numpy(np.ascontiguousarray(model.joint_matrices(model.duration * f / frames), np.float32)).to(dev) for f in range(frames)]
    cam = rt.Camera((-300, 250, 600), 2.7, -0.2).to_uniform(0.0, float(h))
    if os.environ.get("RT_ANIM_STATIC") == "1":  # development: the same pose every frame (what the plan gains without motion)
        mats = [mats[7]] * frames
    legs = []
    # four contexts are slower again (0.23 ms: a process has four hardware queues). threads: one host thread per context
    # (a rebuild ends with a 56-byte read-back the host waits for; with a thread each, the other contexts' launches go on)
    legs_wanted = ((1, 0, False), (1, 64, False), (2, 0, False), (2, 64, False), (3, 0, False), (3, 64, False), (3, 0, True), (3, 64, True))
    if os.environ.get("RT_ANIM_LEGS"):  # development: "contexts,threshold,threads;..." instead
        legs_wanted = tuple((int(a), int(b), bool(int(c))) for a, b, c in (leg.split(",") for leg in os.environ["RT_ANIM_LEGS"].split(";")))
    for n_ctx, heavy, threads in legs_wanted:
        only = os.environ.get("RT_ANIM_ONLY")  # development: one leg, e.g. "1,64,0" (contexts, threshold, threads)
        if only and tuple(int(v) for v in only.split(",")) != (n_ctx, heavy, int(threads)):
            continue
        cores = []
        for _ in range(n_ctx):
            c = rt.TraceCore(dev.index, rt.default_params(heavy_block_threshold=heavy))
            c.set_scene(None, rt.triangles_from_xyz9(model.bind_pose_triangles()))
            c.set_skin(model.positions, model.joints, model.weights, model.tri_indices)
            cores.append(c)
        outs = [(torch.empty((w * h, 2), dtype=torch.int32, device=dev), torch.empty((w * h, 2), dtype=torch.int32, device=dev))
                for _ in range(n_ctx)]

        def frames_of(k, step):
            c, (p, s) = cores[k], outs[k]
            for f in range(k, frames, step):
                c.skin_vertices(mats[f])
                c.build_grid()
                c.trace_primary(cam, w, h, shadow=True, out_primary=p, out_shadow=s)

        def run():
            if threads:
                import threading
                workers = [threading.Thread(target=frames_of, args=(k, n_ctx)) for k in range(n_ctx)]
                for t in workers:
                    t.start()
                for t in workers:
                    t.join()
            else:
                for f, jm in enumerate(mats):
                    c, (p, s) = cores[f % n_ctx], outs[f % n_ctx]
                    c.skin_vertices(jm)
                    c.build_grid()
                    c.trace_primary(cam, w, h, shadow=True, out_primary=p, out_shadow=s)
            torch.cuda.synchronize()

        run()  # the first build of a scene takes the general path; the timed frames rebuild
        t0 = time.perf_counter()
        run()
        frame_ms = (time.perf_counter() - t0) * 1e3 / frames
        rays = w * h + int((outs[(frames - 1) % n_ctx][0][:, 0] != -1).sum())
        leg = {"config": "configs[3] bundled animated model (4 064 triangles, 89 joints), skinning + grid rebuild + 1920x1080 primary + shadow "
                         f"every frame of its clip at 60 frames per second, {n_ctx} context{'s used alternately' if n_ctx > 1 else ''}{', a host thread each' if threads else ''}"
                         + (f", plan of expensive blocks (heavy_block_threshold {heavy})" if heavy else ""),
               "kernel": "k_skin + binning kernels + " + ("k_trace_primary_planned<shadow>" if heavy else "k_trace_primary_lean<shadow>"),
               "contexts": n_ctx, "host_threads": n_ctx if threads else 1, "heavy_block_threshold": heavy, "frames": frames, "frame_ms_wall": frame_ms, "rays_last_frame": rays,
               "Grays_per_s_with_skinning_and_rebuild": rays / frame_ms / 1e6, "bound": "latency",
               "note": "one frame is a chain of about a dozen small launches and a trace whose longest waves are 100 us dependent "
                       "chains: nothing in it is near a throughput roof"}
        if n_ctx == 1:  # the three parts, each timed by HIP events on its own (frame 7 of the animation)
            c, (p, s) = cores[0], outs[0]
            jm = mats[7]
            leg["skin_ms"] = _timed(c, lambda: c.skin_vertices(jm), 20)
            leg["rebuild_ms"] = _timed(c, lambda: c.build_grid(), 20)
            leg["trace_ms"] = _timed(c, lambda: c.trace_primary(cam, w, h, shadow=True, out_primary=p, out_shadow=s), 20)
            leg["build_path"] = c.build_path()
        legs.append(leg)
        for c in cores:
            c.close()
    return legs


def bvh_benchmark(rt, dev, n_tris=1000000):
    """The loop the reference's own example times (examples/massive_benchmark.rs:87-98,226-270): its scene, its tree
    (BVHNode::build, src/core/bvh.rs:44-232 - built on the GPU, node for node the same), its ray set
    (generate_test_rays) through its traversal (both children always), beside the grid walk on the same rays."""
    from ray_tracer_amd import workloads as wl
    core = rt.TraceCore(dev.index)
    core.set_scene(None, wl.generate_test_triangles(n_tris))
    core.build_bvh()  # the first build allocates
    core.sync()
    t0 = time.perf_counter()
    st = core.build_bvh()
    core.sync()
    build_ms = (time.perf_counter() - t0) * 1e3
    core.build_grid(cell_size=2.0)  # the tuned grid of this scene (mean occupancy 4-8)
    legs = []
    for n_rays in (100000, 2000000):  # the example's own count, and one that fills the GPU
        o, d = wl.generate_test_rays(n_rays)
        to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        out = torch.empty(n_rays, dtype=torch.float32, device=dev)
        hits = torch.empty((n_rays, 2), dtype=torch.int32, device=dev)
        ms = _timed(core, lambda: core.trace_rays_bvh(to, td, out=out), 5)
        gms = _timed(core, lambda: core.trace_rays(to, td, out=hits, n=n_rays), 5)
        legs.append({"config": f"configs[2] massive_benchmark's own loop: {n_tris} triangles, BVH (the reference's tree, built on the GPU) "
                               f"walked as the example walks it, generate_test_rays({n_rays})",
                     "kernel": "kb_* (build), k_trace_bvh", "triangles": n_tris, "rays": n_rays, "bvh_nodes": st["node_count"],
                     "bvh_depth": st["max_depth"], "bvh_build_ms_wall": build_ms, "ms": ms, "Mrays_per_s": n_rays / ms / 1e3,
                     "hits": int((out >= 0).sum()), "grid_walk_same_rays_ms": gms, "grid_walk_Mrays_per_s": n_rays / gms / 1e3,
                     "grid_walk_hits": int((hits[:, 0] != -1).sum()), "bound": "latency",
                     "note": "the example opens both children of every node whose box the ray meets, no ordering, no pruning: the "
                             "reference documents ~10-20 Mrays/s for it on its author's CPU (src/core/OPTIMIZATION_TOOLS.md:212); "
                             "its box test loses hits on axis-aligned triangles that the grid walk finds (reproduced as is)"})
    core.close()
    return legs


def all_legs(rt, dev):
    """Every leg, bounded to well under half a minute on one MI355X."""
    legs = []
    t0 = time.perf_counter()
    legs += criterion(rt, dev)
    legs += massive(rt, dev, 3840, 2160, (100000, 1000000), (1, 0))
    legs += animated(rt, dev)
    legs += massive(rt, dev, 7680, 4320, (1000000,), (1,), name="massive8k")
    legs += bvh_benchmark(rt, dev)
    return legs, time.perf_counter() - t0


if __name__ == "__main__":
    import ray_tracer_amd as rt
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    if what == "all":
        legs, wall = all_legs(rt, dev)
        print(json.dumps({"legs": legs, "wall_s": wall}))
    elif what == "criterion":
        print(json.dumps(criterion(rt, dev, reps or 50)))
    elif what == "bvh":
        print(json.dumps(bvh_benchmark(rt, dev)))
    elif what == "animated":
        print(json.dumps(animated(rt, dev)))
    elif what.startswith("massive"):  # e.g. massive4k_100000_skip, massive8k_1000000_skip: one leg, its launches repeated
        kind, n, mode = what.split("_")
        w, h = (7680, 4320) if kind == "massive8k" else (3840, 2160)
        print(json.dumps(massive(rt, dev, w, h, (int(n),), (1 if mode == "skip" else 0,), repeats=reps, name=kind)))
    else:
        raise SystemExit(f"unknown leg {what}")
