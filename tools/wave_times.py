#!/usr/bin/env python3
"""Development tool: per-wave start/end times of one launch of the headline frame (variant 30 = shipped body with timestamps)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RT_LIB_PATH", os.path.join(ROOT, "ray-tracer_amd", "librt_mi355_exp.so"))
import numpy as np, torch
import ray_tracer_amd as rt
from ray_tracer_amd import _lib
W, H = 1920, 1080
core = rt.TraceCore(0)
core.set_scene(None, rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/car_tris.npy"))))
core.build_grid()
cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
p = torch.empty((W * H, 2), dtype=torch.int32, device="cuda"); s = torch.empty_like(p)
os.environ["RT_VARIANT"] = sys.argv[1] if len(sys.argv) > 1 else "30"
nw = (W // 8) * (H // 8)
L = _lib.load()
core.trace_primary(cam, W, H, out_primary=p, out_shadow=s)
core.trace_primary(cam, W, H, out_primary=p, out_shadow=s)
core.sync()
buf = np.zeros(2 * nw, np.uint64)
_lib.load().rt_dbg_wave_times(core._h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(2 * nw))
t0, t1 = buf[0::2].astype(np.int64), buf[1::2].astype(np.int64)
base = t0.min()
start, end = (t0 - base) / 100.0, (t1 - base) / 100.0  # us (100 MHz)
dur = end - start
print(f"kernel span {end.max():.1f} us; wave duration: mean {dur.mean():.2f} median {np.median(dur):.2f} p90 {np.percentile(dur,90):.2f} p99 {np.percentile(dur,99):.2f} max {dur.max():.2f} us")
print("sum of wave durations / (span * 1024 SIMDs) = avg waves per SIMD:", dur.sum() / (end.max() * 1024))
edges = np.linspace(0, end.max(), 15)
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2
    active = ((start <= mid) & (end > mid)).sum()
    started = ((start >= a) & (start < b)).sum()
    print(f"t={mid:6.1f} us  resident waves {active:6d} ({active/1024:.2f}/SIMD)  started in bin {started:6d}  mean dur of those {dur[(start>=a)&(start<b)].mean() if started else 0:.1f}")
# which image rows hold the longest waves
g = dur.reshape(H // 16, 2, W // 16, 2) if False else None
order = np.argsort(-dur)[:10]
print("longest waves (global wave id, start, dur):", [(int(i), round(float(start[i]),1), round(float(dur[i]),1)) for i in order])

