#!/usr/bin/env python3
"""Development tool: times kernel-shape variants of the primary+shadow trace against each other in ONE process,
interleaved rounds, on the headline frame. Needs `make -C ray-tracer_amd/csrc exp` and
RT_LIB_PATH=ray-tracer_amd/librt_mi355_exp.so. Every variant is checked against variant-less output first."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RT_LIB_PATH", os.path.join(ROOT, "ray-tracer_amd", "librt_mi355_exp.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402

variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2]
rounds, steps = 7, 50
W, H = (int(x) for x in os.environ.get("RT_SIZE", "1920x1080").split("x"))
core = rt.TraceCore(0, rt.default_params(heavy_block_threshold=int(os.environ.get("RT_HEAVY", "64"))))
MODEL = os.environ.get("RT_MODEL", "car")
CELL = 0.0
if MODEL.startswith("stream"):  # bench.py's streaming leg: n distinct random triangles, camera inside the volume
    import math
    n = int(MODEL.split(":")[1]) if ":" in MODEL else 10000000
    W, H = 3840, 2160
    rounds, steps = 3, 5
    rng = np.random.default_rng(1)
    size = 100.0 / n ** (1.0 / 3.0)
    c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
    v = (c + rng.normal(0, 0.6 * size, (n, 3, 3)).astype(np.float32)).astype(np.float32)
    core.set_scene(None, rt.triangles_from_xyz9(v.reshape(-1, 9)))
    del c, v
    CELL = float(os.environ.get("RT_CELL_FACTOR", "2.0")) * size
    cam = rt.Camera((0, 0, 75), math.pi, 0.0).to_uniform(0.0, float(H))
elif MODEL.startswith("massive"):  # the reference's benchmark scene, every list entry tested or duplicates left out
    import math
    from ray_tracer_amd import workloads
    n = int(MODEL.split(":")[1]) if ":" in MODEL else 100000
    W, H = 3840, 2160
    rounds, steps = 3, 5
    core.set_scene(None, workloads.generate_test_triangles(n))
    CELL = float(os.environ.get("RT_CELL", "16.0"))
    cam = rt.Camera((0, 0, 75), math.pi, 0.0).to_uniform(0.0, float(H))
elif MODEL == "phoenix":  # the animated model's bind pose, the camera of tools/bench_configs.py
    core.set_scene(None, rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/phoenix_tris.npy"))))
    cam = rt.Camera((-300, 250, 600), 2.7, -0.2).to_uniform(0.0, float(H))
else:
    core.set_scene(None, rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/car_tris.npy"))))
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
core.build_grid(cell_size=CELL)
p = torch.empty((W * H, 2), dtype=torch.int32, device="cuda")
s = torch.empty_like(p)
os.environ.pop("RT_VARIANT", None)
core.trace_primary(cam, W, H, out_primary=p, out_shadow=s)
core.sync()
ref_p, ref_s = p.clone(), s.clone()
if any(v >= 50 for v in variants):
    # hand-out order of the persistent variants 5x: most expensive 8x8 block first, by the wave times of a launch of
    # the instrumented variant 30 (which must be in the build: make exp VARS="0 30 ...")
    import ctypes as C
    from ray_tracer_amd import _lib
    nw = ((W + 15) // 16) * ((H + 15) // 16) * 4
    os.environ["RT_VARIANT"] = "30"
    for _ in range(2):
        core.trace_primary(cam, W, H, out_primary=p, out_shadow=s)
    core.sync()
    buf = np.zeros(2 * nw, np.uint64)
    L = _lib.load()
    L.rt_dbg_wave_times(core._h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(2 * nw))
    dur = buf[1::2].astype(np.int64) - buf[0::2].astype(np.int64)
    order = np.argsort(-dur, kind="stable").astype(np.uint32)
    L.rt_dbg_set_order(core._h, order.ctypes.data_as(C.c_void_p), C.c_size_t(nw))
    print(f"order from {nw} wave times: longest {dur.max() / 100:.1f} us, mean {dur.mean() / 100:.2f} us")
times = {v: [] for v in variants}
# RT_FLIGHT=n: n frames in flight on n streams, as bench.py runs them (RT_TWO=1: two, the older spelling)
FLIGHT = int(os.environ.get("RT_FLIGHT", "2" if os.environ.get("RT_TWO") == "1" else "1"))
TWO = FLIGHT > 1
if TWO:
    streams = [torch.cuda.Stream() for _ in range(FLIGHT)]
    bufs = [(torch.empty_like(p), torch.empty_like(p)) for _ in range(FLIGHT)]


def frames(n):
    if not TWO:
        for _ in range(n):
            core.trace_primary(cam, W, H, out_primary=p, out_shadow=s)
        return
    for k in range(n):
        core.set_stream(streams[k % FLIGHT].cuda_stream)
        core.trace_primary(cam, W, H, out_primary=bufs[k % FLIGHT][0], out_shadow=bufs[k % FLIGHT][1])


for v in variants:
    os.environ["RT_VARIANT"] = str(v)
    p.zero_(); s.zero_()
    core.trace_primary(cam, W, H, out_primary=p, out_shadow=s)
    core.sync()
    ok = torch.equal(p, ref_p) and torch.equal(s, ref_s)
    print(f"variant {v}: {'bit-identical' if ok else 'MISMATCH'}", flush=True)
for r in range(rounds):
    for v in variants:
        os.environ["RT_VARIANT"] = str(v)
        frames(6)
        core.sync()
        if TWO:
            t0 = time.perf_counter()
            frames(4 * steps)
            core.sync()
            times[v].append((time.perf_counter() - t0) / (4 * steps) * 1e6)
        else:
            core.timer_begin()
            frames(steps)
            times[v].append(core.timer_end() / steps * 1e3)
print(f"plan (four lanes per ray, early, threshold us): {core.heavy_plan()}")
for v in variants:
    t = np.array(times[v])
    print(f"variant {v}: median {np.median(t):7.1f} us  min {t.min():7.1f} us")
