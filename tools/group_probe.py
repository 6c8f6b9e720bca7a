"""Development tool: hit records of the headline frame, single frames on three streams against groups of frames in one
launch (rt_trace_primary_frames) on RT_STREAMS streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ray_tracer_amd as rt
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
xyz9 = np.load(os.path.join(ROOT, "tests/golden/car_tris.npy"))
W, H = 1920, 1080
NS = int(os.environ.get("RT_STREAMS", "2"))
G = int(os.environ.get("RT_GROUP", "16"))
core = rt.TraceCore(0, rt.default_params(cull_empty_blocks=int(os.environ.get("RT_CULL", "1"))))
core.set_scene(None, rt.triangles_from_xyz9(xyz9)); core.build_grid()
cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
streams = [torch.cuda.Stream() for _ in range(max(NS, 3))]
bufs = [(torch.empty((G * W * H, 2), dtype=torch.int32, device="cuda"), torch.empty((G * W * H, 2), dtype=torch.int32, device="cuda")) for _ in range(max(NS, 3))]
def singles(n):
    for k in range(n):
        core.set_stream(streams[k % 3].cuda_stream)
        core.trace_primary(cam, W, H, out_primary=bufs[k % 3][0], out_shadow=bufs[k % 3][1])
def groups(n):
    for k in range(n // G):
        core.set_stream(streams[k % NS].cuda_stream)
        core.trace_primary_frames([cam] * G, W, H, out_primary=bufs[k % NS][0], out_shadow=bufs[k % NS][1])
for name, fn in (("single frames, 3 streams", singles), (f"groups of {G}, {NS} streams", groups)) * 2:
    fn(64); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(320); torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 320 * 1e6:.1f} us per frame   mask {core.cull_stats()}")
