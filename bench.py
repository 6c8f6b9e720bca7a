#!/usr/bin/env python3
"""Headline benchmark: Mrays/s at 1920x1080, primary + shadow rays, bundled static model.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one whole frame: W*H primary rays through the DDA grid plus one shadow ray per primary
hit, all inside one kernel launch, with scene, grid and output buffers resident in HBM. Like a renderer
with a two-image swap chain, consecutive frames go to two streams with their own output buffers (two frames in
flight): a frame's last waves - a few pixel blocks over the model's densest cells, one long dependent chain
each - then overlap the next frame's bulk instead of leaving the GPU idle. At N > 1
the SAME frame is cut into 32x32 tiles dealt to the ranks (rank 0, which also gathers and assembles, gets 3 of every 59
at N = 8 where the others get 8; tiles that show nothing but sky are neither rendered nor sent), each rank shades its tiles (same
primary + shadow rays) and the RGBA8 framebuffer is gathered on rank 0 by RCCL (total work fixed: strong
scaling; 3 bytes per pixel cross the links - the picture without its constant alpha byte - instead of 16 for hit
records). A rank's share of one 1080p frame is
only ~10 us of GPU work, far less than one launch + one collective cost on the host, so at N > 1 the steps run as
groups of GROUP consecutive frames: one launch renders the rank's tiles of the whole group, one gather moves them,
one launch assembles them (a renderer running a camera path ahead of display); every frame is still traced,
gathered and assembled in full inside the timed region. Prints one JSON line on rank 0.

The frame's work is counted by the kernel's own counters in two untimed launches: what the REFERENCE algorithm
does (cells visited, candidates tested: the algorithmic byte count) and what the shipped kernel fetches (cell and
reference records it asks the memory system for, after re-test skipping, duplicate removal and LDS staging).

Two roofline objects on the line (N = 1):
  roofline            the headline frame. Its scene (2.5 MB) lives in the caches, so the bound is not HBM: `bound` is
                      "valu_issue" (vector instructions issued per second against the 1 024 SIMDs' issue rate, from the
                      committed hardware-counter pass), and the measured HBM traffic is given beside it as a fraction
                      of the HBM peak. Algorithmic bytes per second are reported, never as a fraction of anything.
  roofline_streaming  a second leg whose scene does NOT fit the caches (RT_BENCH_STREAM_TRIANGLES distinct triangles,
                      default 10 M: 1.6 GB of reference records, 3840x2160, camera inside the volume so every pixel's
                      ray traverses it): `bound` "hbm", `achieved` = algorithmic bytes per launch / average launch
                      duration (HIP events on the launch stream), `traffic` = HBM bytes per launch from the committed
                      counter pass of the same command. The CPU oracle is used for one thing only, outside the timed region: the `cpu_baseline`
leg times the reference's CPU trace path (BVH build + both-children traversal, restated in oracle/) on
the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP gives a process a few hardware queues (four by default) and deals its streams out to them as they are first used;
# a stream keeps its queue, and two busy streams on one queue run one after the other. This program keeps three streams
# busy in its headline leg, and afterwards up to three contexts at once in the legs of the other configurations. When
# every leg made streams of its own - torch hands out another of its 32 pooled streams each time - the contexts' streams
# ended up on ONE queue (the animated model's two-context frame 0.19 -> 0.35 ms after as few as six headline frames on
# three fresh streams: tools/hw_queues_probe.py). Every leg now takes its streams from one small set
# (ray_tracer_amd.shared_stream), which cures it with four queues and with more. RT_BENCH_HW_QUEUES (development)
# sets GPU_MAX_HW_QUEUES for this process; it must be in the environment before the HIP runtime starts.
if os.environ.get("RT_BENCH_HW_QUEUES"):
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["RT_BENCH_HW_QUEUES"]

import numpy as np  # noqa: E402
import torch  # noqa: E402

# The headline workload is the 1080p frame of the bundled static model. RT_BENCH_WORKLOAD=massive8k runs the same
# program on the reference's benchmark scene (src/core/benchmark.rs:99-123, RT_BENCH_TRIANGLES of them, default 1 M) at
# 7680x4320 instead: the configuration meant for a whole node. Never set by the driver.
WORKLOAD = os.environ.get("RT_BENCH_WORKLOAD", "car1080")
WIDTH, HEIGHT = (7680, 4320) if WORKLOAD == "massive8k" else (1920, 1080)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
TILE = 32
# Measured on the headline frame (k_trace_primary_lean, eight waves per SIMD): 2 in flight 70.9 us per frame, 3 in
# flight 68.8, 4 in flight 78.2 (a process has four hardware queues). RT_BENCH_IN_FLIGHT is a development switch; the
# default is what is reported.
FRAMES_IN_FLIGHT = int(os.environ.get("RT_BENCH_IN_FLIGHT", "3"))
# At N > 1 a rank owns 1/N of the tiles but its slowest pixel block still is a ~100 us dependent chain, and every
# frame also spends time on the links: more frames have to be in flight to keep the GPUs busy.
# Frames per gather / assemble at N > 1 (rendered 16 to a launch). At N = 8 a rank's share of a frame is ~10 us of GPU
# time, so a group must hold enough frames for one collective's host cost (a few hundred us through Python) to hide
# behind its rendering.
GROUP = int(os.environ.get("RT_BENCH_GROUP", "4" if WORKLOAD == "massive8k" else "32"))
# Groups in flight: the gathering rank's stream holds render, receive and assemble of a group one after the other, and
# each waits for the last waves of the one before; measured with one GPU playing rank 0 of 8 (tools/group_cost.py,
# RT_INFLIGHT): 12.2 us per frame with two groups in flight, 10.1 with three, 9.3 with four (the other ranks: 8.2-9.6
# whatever the number). Three: a process has four hardware queues and the collective wants one.
GROUPS_IN_FLIGHT = int(os.environ.get("RT_BENCH_GROUPS_IN_FLIGHT", "3"))
# Tiles that show nothing but sky from the frame's camera are neither rendered nor sent at N > 1: which they are follows
# from the camera alone, and rank 0's assemble kernel writes them (byte-identical pictures; RT_BENCH_SPARSE=0: every tile
# travels).
SPARSE = os.environ.get("RT_BENCH_SPARSE", "1") == "1" and WORKLOAD != "massive8k"
PREHEAT_MS = float(os.environ.get("RT_BENCH_PREHEAT_MS", "150"))  # untimed steps before the warm-up steps (see main)


def algorithmic_bytes(stats):
    """Per launch: 8 B per visited cell record, 4 B index + 36 B vertices per triangle candidate
    (24 B per box candidate), 8 B hit record per ray."""
    return 8 * stats["cells_visited"] + 40 * stats["tri_tests"] + 28 * stats["box_tests"] + 8 * stats["rays"]


def moved_bytes(stats, pixels, shadow=True):
    """Per launch: what the shipped kernel requests from the memory system - 8 B per cell record fetched, 48 B per
    reference record fetched (by a lane for itself, look-ahead included, or once per wave into LDS), 8 B per hit
    record written (primary and shadow record of every pixel)."""
    return (8 * stats["cell_records_fetched"] + 48 * (stats["ref_records_fetched"] + stats["ref_records_staged"])
            + 8 * pixels * (2 if shadow else 1))


VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 2  # 1 024 SIMDs, one 64-wide vector instruction per 2 cycles each, 2.4 GHz


def profile_json(name):
    """The committed counter summary `name` of the newest round that has one (profiles/r03_<name>, else r02_<name>)."""
    for rnd in ("r03_", "r02_"):
        path = os.path.join(ROOT, "profiles", rnd + name)
        if os.path.exists(path):
            return json.load(open(path))
    return None


def count_frame(core, cam, width, height, outs):
    """(reference-algorithm counters, shipped-kernel fetch counters) of one frame, untimed."""
    core.set_collect_stats(1)
    core.trace_primary(cam, width, height, shadow=True, out_primary=outs[0], out_shadow=outs[1])
    ref = core.stats()
    core.set_collect_stats(2)
    core.trace_primary(cam, width, height, shadow=True, out_primary=outs[0], out_shadow=outs[1])
    mov = core.stats()
    # the count above follows the frame's mask of pixel blocks that can meet geometry, as launches with frames in
    # flight do: (blocks of the frame, blocks traced); zeros when this frame gets no mask
    mov["pixel_blocks"], mov["pixel_blocks_traced"] = core.cull_stats()
    core.set_collect_stats(0)
    return ref, mov


def streaming_leg(rt, dev, steps, warmup):
    """A frame whose scene does not fit the caches: n distinct small random triangles (seeded) in a 100-unit cube,
    cells of one mean spacing, 3840x2160 primary + shadow, camera inside the cube's front face region so
    that every primary ray enters the grid. One stream, launches back to back: the HIP-event time divided by the
    launches is one launch's duration."""
    import math
    n = int(os.environ.get("RT_BENCH_STREAM_TRIANGLES", "10000000"))
    w, h = 3840, 2160
    rng = np.random.default_rng(1)
    size = 100.0 / n ** (1.0 / 3.0)  # mean spacing
    c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
    v = (c + rng.normal(0, 0.6 * size, (n, 3, 3)).astype(np.float32)).astype(np.float32)
    tris = rt.triangles_from_xyz9(v.reshape(-1, 9))
    del c, v
    # the rays of a wave sit in different cells here: their (ray, candidate) pairs are dealt out again to the lanes
    # (rt_params.repack_candidates; RT_BENCH_STREAM_REPACK=0: the plain loop, 2.50 ms against 1.96)
    repack = int(os.environ.get("RT_BENCH_STREAM_REPACK", "1"))
    core = rt.TraceCore(dev.index, rt.default_params(repack_candidates=repack))
    core.set_scene(None, tris)
    del tris
    cell = float(os.environ.get("RT_BENCH_STREAM_CELL_FACTOR", "1.0")) * size  # the fastest of 1.0 / 1.5 / 2.0 spacings
    meta, n_cells, n_refs = core.build_grid(cell_size=cell)
    cam = rt.Camera((0, 0, 75), math.pi, 0.0).to_uniform(0.0, float(h))
    outs = (torch.empty((w * h, 2), dtype=torch.int32, device=dev), torch.empty((w * h, 2), dtype=torch.int32, device=dev))
    ref, mov = count_frame(core, cam, w, h, outs)
    stream = rt.shared_stream(dev, 0)
    core.set_stream(stream.cuda_stream)
    for _ in range(warmup):
        core.trace_primary(cam, w, h, shadow=True, out_primary=outs[0], out_shadow=outs[1])
    core.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps):
        core.trace_primary(cam, w, h, shadow=True, out_primary=outs[0], out_shadow=outs[1])
    e1.record(stream)
    core.sync()
    kernel_ms = e0.elapsed_time(e1) / steps
    # the same frames three in flight (three streams, three pairs of output arrays), as the headline leg runs its own:
    # the host's clock around the frames between two device-wide waits
    flight = [outs] + [(torch.empty_like(outs[0]), torch.empty_like(outs[1])) for _ in range(2)]
    streams = [stream] + [rt.shared_stream(dev, k) for k in (1, 2)]

    def frames(count):
        for k in range(count):
            core.set_stream(streams[k % 3].cuda_stream)
            core.trace_primary(cam, w, h, shadow=True, out_primary=flight[k % 3][0], out_shadow=flight[k % 3][1])
        torch.cuda.synchronize()

    frames(3)
    t0 = time.perf_counter()
    frames(max(3, steps))
    flight_ms = (time.perf_counter() - t0) * 1e3 / max(3, steps)
    core.sync()
    del flight
    abytes = algorithmic_bytes(ref)
    mbytes = moved_bytes(mov, w * h)
    achieved = abytes / (kernel_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": None, "kernel": "k_trace_primary_repack<shadow>" if repack else "k_trace_primary_lean<shadow>",
           "repack_candidates": repack, "kernel_ms": kernel_ms, "launches": steps, "frame_ms_three_in_flight": flight_ms,
           "Mrays_per_s_three_in_flight": ref["rays"] / flight_ms / 1e3,
           "workload": f"{n} distinct random triangles (seed 1) in a 100-unit cube, grid cell {cell:.4f} ({list(meta.fine_dims)} cells, "
                       f"{n_refs} references = {48 * core.trace_refs() / 1e9:.2f} GB of reference records), {w}x{h} primary + "
                       "shadow rays, camera (0,0,75) looking down -z (every primary ray enters the grid)",
           "rays_per_frame": ref["rays"], "primary_hits": ref["primary_hits"], "Mrays_per_s": ref["rays"] / kernel_ms / 1e3,
           "algorithmic_bytes_per_launch": abytes, "cells_visited": ref["cells_visited"], "tri_tests": ref["tri_tests"],
           "fetched_bytes_per_launch": mbytes, "fetched_GBps": mbytes / (kernel_ms * 1e-3) / 1e9,
           "cell_records_fetched": mov["cell_records_fetched"], "ref_records_fetched": mov["ref_records_fetched"],
           "ref_records_staged": mov["ref_records_staged"]}
    if repack:
        out["fetched_note"] = ("counted by the counting build of the timed kernel (k_trace_primary_repack<count>): a cell record per step "
                               "of a ray still followed, a reference record per look-ahead load of the plain scan and one per dealt pair")
    # What reaches HBM (hardware counters of the same command, committed): `frac` is THAT over the HBM peak. The
    # algorithmic figure stays beside it: neighbouring rays share cells, so most of those bytes are cache hits.
    out["algorithmic_GBps"] = achieved
    t = profile_json("streaming_traffic.json")
    if t and t.get("triangles") == n and abs(t.get("cell_size", cell) - cell) < 1e-3 and t.get("repack_candidates", 0) == repack:
        out["traffic"] = t["bytes_per_launch"]
        out["achieved"] = t["bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
        out["frac"] = out["achieved"] / HBM_PEAK_GBS
        out["algorithmic_over_traffic"] = abytes / t["bytes_per_launch"]
        out["traffic_source"] = t["source"]
        for k in ("l2_hit_rate", "lanes_active_per_valu_instruction", "ta_busy_frac", "valu_issue_frac"):
            if k in t:
                out[k] = t[k]
    else:  # no counter pass for this configuration: nothing to call a fraction of the HBM peak
        out["achieved"] = None
        out["frac"] = None
    core.close()
    return out


def same_route_single_gpu(core, sharding, cam, dev, steps):
    """ms per frame of ONE GPU running the route the N > 1 lines time - every tile of the frame rendered as the shaded
    picture (direct lighting with the shadow-ray term: the same primary + shadow rays), groups of GROUP frames, 3 bytes
    per pixel, assembled to RGBA8 - without a collective. The N > 1 lines are compared with THIS, not with the
    hit-record figure of the N = 1 line, which is another kernel and another output."""
    sh = sharding.FrameSharder(core, WIDTH, HEIGHT, TILE, TILE, 0, 1, dev, shadow=True, mode="rgba", frames_in_flight=1,
                               max_bounces=1, collective=False)
    sh.native_comm = False
    sh.enable_groups(GROUP, GROUPS_IN_FLIGHT, rgb8=True)
    if SPARSE:
        sh.enable_sparse()
        sh.prepare_cameras([cam])
    cams = [cam] * GROUP
    for _ in range(2):
        sh.trace_group(cams)
    sh.synchronize()
    n_groups = max(2, steps // GROUP)
    t0 = time.perf_counter()
    for _ in range(n_groups):
        sh.trace_group(cams)
    sh.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n_groups * GROUP) * 1e3


def cpu_baseline(xyz9, width, height):
    """The reference's CPU trace path (binned-SAH BVH, recursive both-children traversal, Moller-Trumbore
    leaves; examples/massive_benchmark.rs:62-98,226-270) on this frame's primary + shadow rays, all host
    cores, rays chunked per thread. 2 warm-up + 5 timed passes as the example does, bounded to ~20 s."""
    from oracle import binding as ob
    cores = min(os.cpu_count() or 1, 16)
    tris = ob.triangles_from_xyz9(xyz9)
    scene = ob.Scene(None, tris)
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, float(height))
    p, _, st, so, sd = ob.trace_primary(scene, cam, width, height, want_shadow_rays=True, n_threads=cores)
    o, d = ob.primary_rays(cam, width, height)
    hit = p["id"] != ob.MISS
    ro, rd = np.concatenate([o, so[hit]]), np.concatenate([d, sd[hit]])
    bvh = ob.Bvh(tris)
    t0 = time.perf_counter()
    bvh.trace(ro[::64], rd[::64], n_threads=cores)
    probe = (time.perf_counter() - t0) * 64  # estimated seconds for one full pass
    stride = max(1, int(np.ceil(probe * 7 / 20.0)))  # 7 passes within ~20 s
    ro, rd = np.ascontiguousarray(ro[::stride]), np.ascontiguousarray(rd[::stride])
    for _ in range(2):
        bvh.trace(ro, rd, n_threads=cores)
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        bvh.trace(ro, rd, n_threads=cores)
        times.append(time.perf_counter() - t0)
    mean = float(np.mean(times))
    # like-for-like: the oracle's own grid traversal, same rays, same cores
    t0 = time.perf_counter()
    ob.trace_rays(scene, ro, rd, n_threads=cores)
    dda = time.perf_counter() - t0
    t0 = time.perf_counter()
    r1, d1 = np.ascontiguousarray(ro[::8 * cores]), np.ascontiguousarray(rd[::8 * cores])  # spread over the frame
    n1 = len(r1)
    t0 = time.perf_counter()
    bvh.trace(r1, d1, n_threads=1)
    one = time.perf_counter() - t0
    # the criterion scene the reference publishes a number for: 100 rays x 10 000 particle boxes, one thread
    # (benches/bounding_volumes.rs:147-179; 1.0005 ms on the author's machine, benchmark_results.txt:106)
    bmn, bmx = ob.particle_aabbs(10000)
    dirs = np.stack([ob.random_unit_vector(sd) for sd in range(100)]).astype(np.float32)
    zeros = np.zeros_like(dirs)
    ob.intersect_aabb_pairs(zeros, dirs, bmn, bmx)
    t0 = time.perf_counter()
    for _ in range(20):
        ob.intersect_aabb_pairs(zeros, dirs, bmn, bmx)
    pairs_ms = (time.perf_counter() - t0) / 20 * 1e3
    return {"value": len(ro) / mean / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "criterion_100_rays_x_10000_aabbs_ms": pairs_ms,
            "sample": (f"all of the frame's {len(ro)} primary+shadow rays" if stride == 1 else
                       f"one in {stride} of the frame's {len(p) + int(hit.sum())} primary+shadow rays ({len(ro)} rays)")
                      + ", BVH build + both-children traversal, mean of 5 passes",
            "min_ms": min(times) * 1e3, "max_ms": max(times) * 1e3,
            "single_thread_Mrays": n1 / one / 1e6, "grid_traversal_port_Mrays": len(ro) / dda / 1e6}


def preheat(step, sync, steps_per_pass, dist, device):
    """Untimed passes of `steps_per_pass` steps + sync for PREHEAT_MS of wall time. At N > 1 (dist given) every pass
    holds a collective, so the ranks must make the same number of passes: rank 0's clock decides and the decision is
    handed round after every pass (a rank that went on alone would stay one gather ahead of the others for good).
    Returns the number of passes."""
    t_pre = time.perf_counter()
    go_on = PREHEAT_MS > 0
    passes = 0
    while go_on:
        for _ in range(steps_per_pass):
            step()
        sync()
        passes += 1
        go_on = (time.perf_counter() - t_pre) * 1e3 < PREHEAT_MS
        if dist is not None:
            word = torch.tensor([1 if go_on else 0], dtype=torch.int32, device=device)
            dist.broadcast(word, 0)
            go_on = bool(int(word.item()))
    return passes


def secondary(name, leg):
    """A leg beside the headline measurement: whatever goes wrong in it is reported in its place (and on stderr), and the
    line with the headline figure is still printed."""
    try:
        return leg()
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc(file=sys.stderr)
        return {"error": f"{name}: {type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # `python bench.py --gpus N` started plainly (no launcher, WORLD_SIZE unset): this process becomes the launcher. It
    # starts torch.distributed.run as a CHILD with one rank per GPU, hands its stdout (the one JSON line of rank 0) and
    # stderr through, and exits with the child's code. Decided here, before anything has touched a GPU.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
               "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        raise SystemExit(subprocess.call(cmd))

    import torch.distributed as dist

    # The contract is ONE line on stdout. Libraries print there too (RCCL writes a version banner when its first
    # communicator is made), so everything else this process writes to file descriptor 1 goes to stderr, and the JSON
    # line is written to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(real_stdout, (line + "\n").encode())

    if os.environ.get("RT_BENCH_WATCHDOG"):  # development: dump every thread's stack if the run takes this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["RT_BENCH_WATCHDOG"]), exit=True)

    import ray_tracer_amd as rt
    from ray_tracer_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if os.environ.get("RT_BENCH_DRY_RUN") == "1":  # the launch path alone, on CPU (tests): rendezvous, one collective, one line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ones = torch.ones(1)
        dist.all_reduce(ones)
        # the pre-heat loop with ranks of different speed (a pass = one collective): every rank must make as many passes
        count = torch.zeros(1)

        def slow_pass():
            time.sleep(0.004 * (1 + rank))
            dist.all_reduce(count.add_(1))

        passes = torch.tensor([float(preheat(lambda: None, slow_pass, 1, dist if world > 1 else None, "cpu"))])
        lo, hi = passes.clone(), passes.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if rank == 0:
            emit(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": int(ones.item()), "steps": args.steps, "warmup": args.warmup,
                             "preheat_passes_min": int(lo.item()), "preheat_passes_max": int(hi.item())}))
        dist.destroy_process_group()
        return
    # Rehearsal on a one-GPU box: RT_BENCH_ONE_GPU=1 puts every rank on GPU 0 and uses gloo (RCCL refuses two
    # ranks on one device). Never set by the driver.
    rehearsal = os.environ.get("RT_BENCH_ONE_GPU") == "1"
    # Development: RT_BENCH_RCCL_WORLD1=1 takes the N > 1 route (RCCL communicator, broadcast, gather, assemble) with
    # a single rank, which is all of that route a one-GPU box can run on the real backend. Never set by the driver.
    multi = world > 1 or os.environ.get("RT_BENCH_RCCL_WORLD1") == "1"
    if multi and world == 1:
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if os.environ.get("RT_BENCH_LEG") == "streaming" and not multi:  # development / profiling: the second leg alone
        emit(json.dumps({"metric": "streaming leg only", "n_gpus": 1,
                         "roofline_streaming": streaming_leg(rt, dev, int(os.environ.get("RT_BENCH_STREAM_STEPS", "20")), 3)}))
        return

    xyz9 = None
    if WORKLOAD == "massive8k":
        import math
        from ray_tracer_amd import workloads
        n_tris = int(os.environ.get("RT_BENCH_TRIANGLES", "1000000"))
        tris = workloads.generate_test_triangles(n_tris) if rank == 0 else None
        pose = rt.Camera((0, 0, 150), math.pi, 0.0)
        data = f"the reference's benchmark scene, generate_test_triangles({n_tris}) (10 000 positions, repeated)"
        workload = (f"configs[4]: massive_benchmark scene, {n_tris} triangles, 7680x4320 primary rays through the DDA uniform "
                    "grid (the reference's cell size 16) + one shadow ray per primary hit, camera (0,0,150) looking down -z")
    else:
        xyz9 = np.load(os.path.join(ROOT, "tests", "golden", "car_tris.npy"))
        tris = rt.triangles_from_xyz9(xyz9) if rank == 0 else None
        pose = rt.Camera.for_scene("gltf")
        data = ("one frame, repeated every step (same camera, caches warm), of the reference's bundled static model "
                "(models/no_animation, 3132 triangles, flattened to a fixture); no file or dataset is read in the timed region")
        workload = ("configs[1]: static model, 1920x1080 primary rays through the DDA uniform grid "
                    "+ one shadow ray per primary hit, SCENE=gltf start camera")
    # RT_BENCH_CULL=0 (development): every ray walks the grid, no mask of pixel blocks that cannot meet geometry
    core = rt.TraceCore(local_rank, rt.default_params(cull_empty_blocks=int(os.environ.get("RT_BENCH_CULL", "1"))))
    if multi:
        # scene replicated by ONE RCCL broadcast through the C ABI (rt_comm_init + rt_broadcast_scene: device to
        # device, straight into the scene buffers), grid binned locally on every GPU. The rehearsal on one GPU cannot
        # have an RCCL communicator (two ranks on one device are refused): torch.distributed over gloo stands in.
        if not rehearsal:
            # every rank must end up on the same route: if the library's communicator cannot be made on any of them
            # (said loudly), all fall back to torch.distributed's RCCL process group for the broadcast and the gathers
            try:
                if os.environ.get("RT_BENCH_NO_NATIVE_COMM") == "1":  # development: exercise the fall-back
                    raise RuntimeError("RT_BENCH_NO_NATIVE_COMM")
                sharding.init_comm(core, rank, world)
                ok = 1
            except Exception as e:  # noqa: BLE001
                print(f"rank {rank}: rt_comm_init failed ({e}); torch.distributed collectives instead", file=sys.stderr, flush=True)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and getattr(core, "comm_size", 0):
                core.comm_destroy()
        sharding.replicate_scene(core, np.zeros(0, rt.BOX_DTYPE) if rank == 0 else None, tris)
    else:
        core.set_scene(None, tris)
    meta, n_cells, n_refs = core.build_grid()
    cam = pose.to_uniform(0.0, float(HEIGHT))

    # the frame's work, counted by the kernels themselves (untimed): the reference algorithm's, and what is fetched
    cnt = (torch.empty((WIDTH * HEIGHT, 2), dtype=torch.int32, device=dev), torch.empty((WIDTH * HEIGHT, 2), dtype=torch.int32, device=dev))
    stats, moved = count_frame(core, cam, WIDTH, HEIGHT, cnt)
    del cnt
    rays_per_frame = stats["rays"]
    cull_blocks, cull_kept = moved["pixel_blocks"], moved["pixel_blocks_traced"]

    if not multi:
        streams = [rt.shared_stream(dev, k) for k in range(FRAMES_IN_FLIGHT)]  # the same few streams in every leg: see there
        bufs = [(torch.empty((WIDTH * HEIGHT, 2), dtype=torch.int32, device=dev),
                 torch.empty((WIDTH * HEIGHT, 2), dtype=torch.int32, device=dev)) for _ in range(FRAMES_IN_FLIGHT)]
        counter = [0]

        def step():
            k = counter[0] % FRAMES_IN_FLIGHT
            counter[0] += 1
            core.set_stream(streams[k].cuda_stream)
            core.trace_primary(cam, WIDTH, HEIGHT, shadow=True, out_primary=bufs[k][0], out_shadow=bufs[k][1])

        def sync():
            core.sync()
    else:
        # the framebuffer that travels is the shaded RGBA8 picture; one bounce = the same primary + shadow rays
        # rank 0 also receives and assembles every frame, so it renders a smaller share: 5 tiles of every 61 at
        # N = 8 where the others render 8 (RT_BENCH_WEIGHTS=equal, or a comma-separated list, for experiments)
        wenv = os.environ.get("RT_BENCH_WEIGHTS", "")
        # with the sky tiles left out a rank's share costs less and rank 0's assemble (the whole frame, sky included) as much
        # as before: 3 tiles of every 59 at N = 8 (11 of 59 at N = 4, 27 of 59 at N = 2) level the ranks, measured as above
        weights = (None if wenv == "equal" else [int(k) for k in wenv.split(",")] if wenv
                   else sharding.gatherer_weights(world, lag=5 if SPARSE else 3))
        sh = sharding.FrameSharder(core, WIDTH, HEIGHT, TILE, TILE, rank, world, dev, shadow=True, mode="rgba",
                                   frames_in_flight=1, max_bounces=1, collective=True, weights=weights, host_gather=rehearsal)
        sh.enable_groups(GROUP, GROUPS_IN_FLIGHT, rgb8=os.environ.get("RT_BENCH_WIRE", "rgb") == "rgb")  # the constant alpha byte stays at home
        if SPARSE:
            sh.enable_sparse()
            sh.prepare_cameras([cam])  # one small synchronous pass on every rank, before the timed region
        if os.environ.get("RT_BENCH_TRACE"):  # development: where every rank is, every few seconds
            import threading

            def report():
                while True:
                    time.sleep(5)
                    print(f"[rank {rank}] groups started {sh.group_no}, streams idle "
                          f"{[sl.stream.query() for sl in sh.gslots]}", file=sys.stderr, flush=True)
            threading.Thread(target=report, daemon=True).start()
        sh.trace_group([cam])  # set-up, untimed: the first collective creates the RCCL communicator
        sh.synchronize()
        pending = []

        def step():
            if os.environ.get("RT_BENCH_PER_FRAME"):  # development: one launch, gather and assemble per frame
                sh.trace(cam)
                return
            pending.append(cam)
            if len(pending) == GROUP:
                sh.trace_group(pending)
                pending.clear()

        def sync():
            if pending:  # a last, partial group
                sh.trace_group(pending)
                pending.clear()
            sh.synchronize()

    base_ms = None
    if multi and rank == 0:  # the single-GPU figure of the same route, measured while the other ranks wait
        base_ms = secondary("scaling_base", lambda: same_route_single_gpu(core, sharding, cam, dev, max(64, args.steps)))
        base_ms = None if isinstance(base_ms, dict) else base_ms  # an extra beside the measurement: reported on stderr if it fails
    # Untimed, before the W warm-up steps: the same steps for PREHEAT_MS of wall time, so that the timed region runs at
    # the clocks a renderer sees in steady state however few steps the caller asks for (the first tens of milliseconds
    # after an idle period run ~6 % slower: clock ramp).
    preheat(step, sync, 8 if not multi else GROUP, dist if multi and world > 1 else None, "cpu" if rehearsal else dev)
    for _ in range(args.warmup):
        step()
    sync()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = None
    if not multi:  # HIP events on the streams the kernels are launched on
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in streams]
        for (e0, _), st in zip(ev, streams):
            e0.record(st)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    submitted = time.perf_counter() - t0  # host time to queue the steps (the GPU runs behind it)
    if not multi:
        for (_, e1), st in zip(ev, streams):
            e1.record(st)
    if multi:
        sync()  # starts what is left of the last group, then waits for this rank's streams
    torch.cuda.synchronize()  # the whole device: every stream the steps were queued on
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if not multi:
        sync()  # nothing left to wait for: the library notes that no frame is in flight any more
    if not multi:
        # average time one launch occupies its stream; with several frames in flight as many launches overlap, so
        # this is about that many times the time per frame
        per_stream = [(args.steps + FRAMES_IN_FLIGHT - 1 - k) // FRAMES_IN_FLIGHT for k in range(FRAMES_IN_FLIGHT)]
        kernel_ms = float(np.mean([e0.elapsed_time(e1) / n for (e0, e1), n in zip(ev, per_stream) if n]))
    single_us = None
    if not multi:  # one frame at a time, nothing else in flight: launch-to-done latency
        core.set_stream(streams[0].cuda_stream)
        lat = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(streams[0])
            core.trace_primary(cam, WIDTH, HEIGHT, shadow=True, out_primary=bufs[0][0], out_shadow=bufs[0][1])
            e1.record(streams[0])
            core.sync()
            lat.append(e0.elapsed_time(e1) * 1e3)
        single_us = float(np.median(lat[5:]))
    if not multi:
        base_ms = secondary("scaling_base", lambda: same_route_single_gpu(core, sharding, cam, dev, max(64, args.steps)))
        base_ms = None if isinstance(base_ms, dict) else base_ms  # an extra beside the measurement: reported on stderr if it fails
    hbm_copy = None
    if not multi:  # what this GPU's HBM delivers on the library's plain copy kernel (2 GiB read + 2 GiB written per pass)
        a = torch.zeros(1 << 29, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        torch.cuda.synchronize()
        core.set_stream(None)
        nbytes = a.numel() * 4
        core.probe_copy(a.data_ptr(), b.data_ptr(), nbytes)
        core.timer_begin()
        for _ in range(5):
            core.probe_copy(a.data_ptr(), b.data_ptr(), nbytes)
        hbm_copy = 2 * nbytes * 5 / (core.timer_end() * 1e-3) / 1e9
        del a, b
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_per_frame * args.steps / elapsed / 1e6
        out = {
            "metric": "Mrays/sec @1080p primary+shadow" if WORKLOAD != "massive8k" else "Mrays/sec @8K primary+shadow", "value": value, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": data,
            "config": {"workload": workload,
                       "width": WIDTH, "height": HEIGHT, "rays_per_frame": rays_per_frame,
                       "primary_hits": stats["primary_hits"], "grid": list(meta.fine_dims), "cell_size": 16.0,
                       "cells": n_cells, "cell_refs": n_refs, "trace_refs": core.trace_refs(),
                       "pixel_blocks": cull_blocks, "pixel_blocks_traced": cull_kept,
                       "frames_in_flight": FRAMES_IN_FLIGHT if not multi else GROUP * GROUPS_IN_FLIGHT,
                       "frames_per_gather": 1 if not multi else GROUP, "frames_per_launch": 1 if not multi else min(GROUP, 16),
                       "output": "rt_hit records, 16 B per pixel, resident in HBM" if not multi else
                                 "shaded RGBA8 picture (ambient + diffuse with the shadow-ray term) assembled on rank 0; 3 B per pixel on the links (alpha is constant)",
                       "parallelism": "single GPU" if not multi else f"{TILE}x{TILE} tiles dealt over {world} GPUs, "
                                      f"{sh.weights or [1] * world} of every {sum(sh.weights or [1] * world)} (the gathering rank renders less), RCCL gather to rank 0"},
        }
        # what N > 1 lines are to be compared with: one GPU on the route they time (picture, groups, assemble)
        out["host_submit_us_per_step"] = submitted / args.steps * 1e6
        out["preheat_ms"] = PREHEAT_MS
        out["scaling_base_ms_per_step"] = base_ms
        out["scaling_base_note"] = ("one GPU, the N > 1 route: k_render direct lighting + shadow term (same rays), groups of "
                                    f"{GROUP} frames, RGB8, assembled to RGBA8, no collective")
        if multi and base_ms:
            out["speedup_vs_scaling_base"] = base_ms / ms_per_step
        if multi:
            plan = sh._sparse_plan([cam]) if sh.sparse else None
            tiles = sharding.n_tiles(WIDTH, HEIGHT, TILE, TILE)
            out["config"]["tiles"] = tiles
            out["config"]["tiles_rendered_and_sent"] = plan["kept_tiles"] if plan else tiles
            out["config"]["wire_bytes_per_frame_per_rank"] = sh.last_wire_bytes // GROUP
        if multi:  # false: the gathers of this line are torch.distributed's (or the rehearsal's), not rt_gather_frames
            out["native_comm"] = bool(getattr(core, "comm_size", 0))
        out["timed_kernel"] = ("k_trace_primary_lean<shadow>: rt_hit records" if not multi else
                               "k_render<single bounce, shadow term>: RGB8 picture, "
                               + ("rt_gather_frames (RCCL through the C ABI)" if getattr(core, "comm_size", 0) else
                                  "torch.distributed gather (the library's communicator could not be made)" if not rehearsal else
                                  "host-staged gloo gather (one-GPU rehearsal)") + ", rt_assemble_frames")
        if not multi:
            abytes = algorithmic_bytes(stats)
            mbytes = moved_bytes(moved, WIDTH * HEIGHT)
            frame_s = elapsed / args.steps
            roof = {"kernel": "k_trace_primary_lean<shadow>", "kernel_ms": kernel_ms, "concurrent_launches": FRAMES_IN_FLIGHT,
                    "single_frame_us": single_us,
                    "algorithmic_bytes_per_launch": abytes, "algorithmic_GBps": abytes / frame_s / 1e9,
                    "cells_visited": stats["cells_visited"], "tri_tests": stats["tri_tests"],
                    "fetched_bytes_per_launch": mbytes, "fetched_GBps": mbytes / frame_s / 1e9,
                    "cell_records_fetched": moved["cell_records_fetched"], "ref_records_fetched": moved["ref_records_fetched"],
                    "ref_records_staged": moved["ref_records_staged"], "measured_copy_GBps": hbm_copy,
                    "mask_kernel": ("k_block_mask (the frame's mask of 8x8 pixel blocks that can meet geometry, rt_params.cull_empty_blocks) "
                                    "runs ahead of every launch on the same stream; kernel_ms and single_frame_us cover both"),
                    "note": ("algorithmic bytes are the reference algorithm's (every list entry of every visited cell, 8 B per cell, "
                             "40 B per candidate, 8 B per ray); fetched bytes are what the timed kernel requests from the memory "
                             "system; traffic is what reached HBM (hardware counters). Rates are per frame of the timed region "
                             "(launches overlap), not per launch.")}
            t = profile_json("headline_traffic.json") if WORKLOAD == "car1080" else None
            v = profile_json("headline_valu.json") if WORKLOAD == "car1080" else None
            if t:
                roof["traffic"] = t["bytes_per_launch"]
                roof["hbm_traffic_frac"] = t["bytes_per_launch"] / frame_s / 1e9 / HBM_PEAK_GBS
                roof["algorithmic_over_traffic"] = abytes / t["bytes_per_launch"]
                roof["traffic_source"] = t["source"]
            else:
                roof["traffic"] = None
            if v:
                # the scene is read from the caches; what a frame costs is instruction issue
                ginstr = v["SQ_INSTS_VALU_per_launch"] / frame_s / 1e9
                roof.update({"bound": "valu_issue", "achieved": ginstr, "peak": VALU_PEAK_GINSTR, "unit": "Ginstr/s",
                             "frac": ginstr / VALU_PEAK_GINSTR, "valu_instructions_per_launch": v["SQ_INSTS_VALU_per_launch"],
                             "salu_instructions_per_launch": v["SQ_INSTS_SALU_per_launch"],
                             "lanes_active_per_valu_instruction": v["lanes_active_per_valu_instruction"],
                             "counter_source": v["source"]})
            else:  # no counter pass committed for this workload: the HBM view only, on measured traffic where there is one
                a = (roof["traffic"] or 0) / frame_s / 1e9
                roof.update({"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS})
            out["roofline"] = roof
            if WORKLOAD != "car1080":
                out["roofline"]["note"] += (" The timed kernels leave out exact duplicate triangles (config.trace_refs of "
                                            "config.cell_refs); the fetched figure counts what they walk.")
            if os.environ.get("RT_BENCH_LEG", "both") != "headline" and WORKLOAD == "car1080":
                core.close()
                out["roofline_streaming"] = secondary("roofline_streaming", lambda: streaming_leg(
                    rt, dev, int(os.environ.get("RT_BENCH_STREAM_STEPS", "20")), 3))
            # the other BASELINE configurations on this GPU, one bounded leg each (tools/config_legs.py); RT_BENCH_CONFIGS=0
            # leaves them out (profiling runs of the headline kernel)
            if WORKLOAD == "car1080" and os.environ.get("RT_BENCH_CONFIGS", "1") != "0" and os.environ.get("RT_BENCH_LEG", "both") == "both":
                if core._h:
                    core.close()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import config_legs
                legs = secondary("configs", lambda: config_legs.all_legs(rt, dev))
                if isinstance(legs, tuple):
                    out["configs"], out["configs_wall_s"] = legs
                else:
                    out["configs"] = legs
            if not args.no_cpu_baseline and xyz9 is not None:
                out["cpu_baseline"] = secondary("cpu_baseline", lambda: cpu_baseline(xyz9, WIDTH, HEIGHT))
        emit(json.dumps(out))
    if core._h:
        core.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
