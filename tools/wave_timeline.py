#!/usr/bin/env python3
"""Development tool: when each wave of one primary+shadow launch started and ended. RT_VARIANT=71 / 72: the shipped
kernel without / with the frame's block mask (stamped in development builds). RT_VARIANT=30: the instrumented
plain kernel, one wave per 8x8 pixel block (build with `make -C ray-tracer_amd/csrc exp VARS="0 30"`); RT_VARIANT=99: the
shipped kernel with its plan of expensive blocks (RT_HEAVY = the threshold), which stamps itself in development builds.
Prints the launch's span, the distribution of run times, how many waves were resident over time, and what ended last."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RT_LIB_PATH", os.path.join(ROOT, "ray-tracer_amd", "librt_mi355_exp.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import _lib  # noqa: E402

W, H = (int(x) for x in os.environ.get("RT_SIZE", "1920x1080").split("x"))
MODEL = os.environ.get("RT_MODEL", "car")
VARIANT = os.environ.get("RT_VARIANT", "30")
core = rt.TraceCore(0, rt.default_params(heavy_block_threshold=int(os.environ.get("RT_HEAVY", "64"))))
if MODEL == "phoenix":
    core.set_scene(None, rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/phoenix_tris.npy"))))
    cam = rt.Camera((-300, 250, 600), 2.7, -0.2).to_uniform(0.0, float(H))
else:
    core.set_scene(None, rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/car_tris.npy"))))
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
core.build_grid()
p = torch.empty((W * H, 2), dtype=torch.int32, device="cuda")
s = torch.empty_like(p)
nw = ((W + 15) // 16) * ((H + 15) // 16) * 4
os.environ["RT_VARIANT"] = VARIANT
for _ in range(int(os.environ.get("RT_FRAMES", "5"))):
    core.trace_primary(cam, W, H, out_primary=p, out_shadow=s)
core.sync()
buf = np.zeros(2 * nw, np.uint64)
L = _lib.load()
L.rt_dbg_wave_times(core._h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(2 * nw))
chain = np.zeros(nw, np.uint32)
if VARIANT == "99":
    L.rt_dbg_wave_chain(core._h, chain.ctypes.data_as(C.c_void_p), C.c_size_t(nw))
    print(f"blocks listed for the next frame (heavy, early): {core.heavy_blocks()}; this frame: {(chain >> 31).sum()} four lanes per ray, {((chain >> 30) == 1).sum()} early")
t0, t1 = buf[0::2].astype(np.int64), buf[1::2].astype(np.int64)
live = t1 > 0
base = t0[live].min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0  # microseconds
dur = t1 - t0
print(f"{MODEL} {W}x{H}: {live.sum()} waves, span {t1[live].max():.1f} us, last start {t0[live].max():.1f} us")
print("run time percentiles (us): " + "  ".join(f"p{q}={np.percentile(dur[live], q):.1f}" for q in (50, 90, 99, 99.9, 100)))
print(f"sum of run times {dur[live].sum() / 1e3:.2f} wave-ms = {dur[live].sum() / (256 * 4 * 7):.1f} us on 7168 wave slots")
edges = np.arange(0, t1[live].max() + 5, 5.0)
for e in edges:
    n = ((t0 <= e) & (t1 > e) & live).sum()
    print(f"  t={e:6.1f} us  resident waves {n:6d}")
hits = p.cpu().numpy().view(rt.HIT_DTYPE).reshape(H, W)
bx = (W + 15) // 16
if W % 16 == 0 and H % 8 == 0:
    # run time of the waves whose 8x8 block holds no primary hit at all
    any_hit = (hits["id"] != rt.RT_MISS).reshape(H // 8, 8, W // 8, 8).any(axis=(1, 3))
    gws = np.arange(nw)
    b16s, g4s = gws >> 2, gws & 3
    x8s, y8s = (b16s % bx) * 2 + (g4s & 1), (b16s // bx) * 2 + (g4s >> 1)
    ok = live & (y8s < H // 8)
    miss_block = ok & ~any_hit[np.minimum(y8s, H // 8 - 1), x8s]
    print(f"blocks without a primary hit: {miss_block.sum()} of {ok.sum()}, their run times sum to {dur[miss_block].sum() / 1e3:.2f} "
          f"of {dur[ok].sum() / 1e3:.2f} wave-ms (mean {dur[miss_block].mean():.2f} us against {dur[ok & ~miss_block].mean():.2f} us)")
print("waves that ended last: (block x, y of 8x8), start, run time")
for gw in np.argsort(-np.where(live, t1, 0))[:24]:
    b16, g4 = gw >> 2, gw & 3
    x8, y8 = (b16 % bx) * 2 + (g4 & 1), (b16 // bx) * 2 + (g4 >> 1)
    print(f"  gw {gw:6d} block ({x8:3d},{y8:3d})  start {t0[gw]:6.1f}  run {dur[gw]:6.1f}  end {t1[gw]:6.1f}  {('', ' early', ' quad', ' quad')[chain[gw] >> 30]}")
