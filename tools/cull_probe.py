"""Development tool: cost and effect of the block mask (rt_params.cull_empty_blocks) on the headline frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ray_tracer_amd as rt
xyz9 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/car_tris.npy"))
W, H = 1920, 1080
NS = int(os.environ.get("RT_STREAMS", "3"))
streams = [torch.cuda.Stream() for _ in range(NS)]  # the same streams (hardware queues) for every mode
for cull in (0, 1, 2):
    core = rt.TraceCore(0, rt.default_params(cull_empty_blocks=cull))
    core.set_scene(None, rt.triangles_from_xyz9(xyz9)); core.build_grid()
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
    bufs = [(torch.empty((W*H, 2), dtype=torch.int32, device="cuda"), torch.empty((W*H, 2), dtype=torch.int32, device="cuda")) for _ in range(NS)]
    def frames(n):
        for k in range(n):
            core.set_stream(streams[k % NS].cuda_stream)
            core.trace_primary(cam, W, H, out_primary=bufs[k % NS][0], out_shadow=bufs[k % NS][1])
    frames(12); torch.cuda.synchronize()
    print("cull", cull, "stats", core.cull_stats())
    for rep in range(3):
        t0 = time.perf_counter()
        frames(300)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("  %d streams: host submit %.1f us/frame, total %.1f us/frame" % (NS, (t1-t0)/300*1e6, (t2-t0)/300*1e6))
    core.close()
