"""Development aid: what one rank of an N-GPU run spends per 1080p frame on its own GPU (no collective): its tiles of
GROUP frames per launch, groups in flight on separate streams, plus rank 0's assemble. Gives the compute-side bound of
the N-GPU frame rate; the link-side bound is (N-1)/N * 8.3 MB per frame into rank 0 (RT_SPARSE=1: tiles that show nothing but sky are neither
rendered nor sent - rt_tile_flags, RT_OWNER_SKY, rt_assemble_frames_sky; every line also says how many bytes of a frame
the other ranks send to rank 0)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import sharding  # noqa: E402

W, H = int(os.environ.get("RT_W", "1920")), int(os.environ.get("RT_H", "1080"))
TILE = int(os.environ.get("RT_TILE", "32"))
TILE_H = int(os.environ.get("RT_TILE_H", str(TILE)))
RGB8 = os.environ.get("RT_WIRE", "rgb") == "rgb"
SPARSE = os.environ.get("RT_SPARSE") == "1"


def main():
    dev = torch.device("cuda", 0)
    core = rt.TraceCore(0)
    if os.environ.get("RT_SCENE") == "massive":  # the reference's benchmark scene, 1 M triangles, its camera
        import math
        from ray_tracer_amd import workloads
        core.set_scene(None, workloads.generate_test_triangles(1000000))
        cam = rt.Camera((0, 0, 150), math.pi, 0.0).to_uniform(0.0, float(H))
    else:
        core.set_scene(None, rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests", "golden", "car_tris.npy"))))
        cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
    core.build_grid()
    out = []
    for world in [int(x) for x in os.environ.get('RT_WORLDS', '1,2,4,8').split(',')]:
        for group, in_flight in ((int(os.environ['RT_GROUP']), int(os.environ.get('RT_INFLIGHT', '2'))),) if 'RT_GROUP' in os.environ else ((1, 4), (4, 3), (8, 3), (16, 2)) if 'RT_QUICK' not in os.environ else (((8, 2),) if os.environ['RT_QUICK'] == '2' else ((8, 3), (16, 2))):
            per_rank = []
            for rank in (range(world - 1, -1, -1) if os.environ.get('RT_REVERSE') else range(world)):
                sh = sharding.FrameSharder(core, W, H, TILE, TILE_H, rank=rank, world=1, device=dev, shadow=True, mode="rgba", max_bounces=1)
                weights = ([int(k) for k in os.environ['RT_WEIGHTS'].split(',')] if os.environ.get('RT_WEIGHTS') else sharding.gatherer_weights(world)) if os.environ.get('RT_WEIGHTED') else None
                sky_owner = None
                if SPARSE:  # the kept tiles dealt in the pattern of the weights; sky tiles belong to nobody
                    kept = np.flatnonzero(core.tile_flags(cam, W, H, TILE, TILE_H))
                    table = np.full(sharding.n_tiles(W, H, TILE, TILE_H), rt.RT_OWNER_SKY, np.uint32)
                    table[kept] = np.resize(np.asarray(sharding.owner_table(weights or [1] * world), np.uint32), len(kept))
                    sky_owner = [int(k) for k in table]
                    shards = [rt.RtShard(TILE, TILE_H, r, owner=sky_owner) for r in range(world)]
                    sh.shard = shards[rank]
                    sh.per_rank = max(1, max(rt.shard_pixels(W, H, s) for s in shards))
                    sent = sum(rt.shard_pixels(W, H, s) for s in shards[1:])
                elif weights:
                    owner = sharding.owner_table(weights)
                    shards = [rt.RtShard(TILE, TILE_H, r, owner=owner) for r in range(world)]
                    sh.shard = shards[rank]
                    sh.per_rank = max(rt.shard_pixels(W, H, s) for s in shards)
                    sent = sum(rt.shard_pixels(W, H, s) for s in shards[1:])
                else:
                    sh.shard = rt.RtShard(TILE, TILE_H, rank, world)  # this rank's tiles of a `world`-GPU run, no gather
                    sh.per_rank = sharding.per_rank_pixels(W, H, TILE, TILE_H, world)
                    sent = sum(rt.shard_pixels(W, H, rt.RtShard(TILE, TILE_H, r, world)) for r in range(1, world))
                # the gathering rank's extra work: one assemble per group, on the group's stream
                packed = images = None
                if rank == 0 and os.environ.get('RT_WITH_ASSEMBLE'):
                    packed = torch.zeros((world, group, sh.per_rank, 3 if RGB8 else 4), dtype=torch.uint8, device=dev)
                    images = torch.empty((in_flight, group, W * H, 4), dtype=torch.uint8, device=dev)
                sh.enable_groups(group, in_flight, rgb8=RGB8)
                cams = [cam] * group
                slots = sh.gslots
                if sky_owner is not None:
                    import ctypes as C
                    sky_owner_c = (C.c_uint32 * len(sky_owner))(*sky_owner)
                    cams_c = (rt.CameraUniform * group)()
                    for j in range(group):
                        C.memmove(C.byref(cams_c, j * C.sizeof(rt.CameraUniform)), C.byref(cam), C.sizeof(rt.CameraUniform))
                def run(n):
                    for k in range(n):
                        slot = slots[k % len(slots)]
                        core.set_stream(slot.stream.cuda_stream)
                        core.render_frames(cams, W, H, shard=sh.shard, max_bounces=1, shadows=True, out=slot.local, rgb8=RGB8,
                                           frame_stride=sh.per_rank)
                        if packed is not None and sky_owner is not None:
                            core.assemble_frames_sky(packed, RGB8, W, H, TILE, TILE_H, world, sky_owner_c, cams_c, group * sh.per_rank, sh.per_rank,
                                                     images[k % len(slots)])
                        elif packed is not None:
                            core.assemble_frames(packed, RGB8, W, H, TILE, TILE_H, world, sharding.owner_table(weights) if weights else None, group, group * sh.per_rank,
                                                 sh.per_rank, images[k % len(slots)])
                    core.sync()
                t_pre = time.perf_counter()  # untimed, until the clocks have settled (the first rank measured read 2-3 us high)
                while time.perf_counter() - t_pre < 0.2:
                    run(10)
                n = max(10, int(os.environ.get('RT_FRAMES', '400')) // group)
                t0 = time.perf_counter()
                run(n)
                per_rank.append((time.perf_counter() - t0) / (n * group) * 1e6)
                core.set_stream(None)
            out.append({"world": world, "group": group, "sparse": SPARSE, "bytes_per_frame_into_rank0": sent * (3 if RGB8 else 4),
                        "pixels_per_rank_block": sh.per_rank, "groups_in_flight": in_flight, "us_per_frame_by_rank": [round(x, 1) for x in per_rank],
                        "slowest_rank_us": round(max(per_rank), 1), "speedup_bound": round(out[0]["slowest_rank_us"] / max(per_rank), 2) if out else 1.0})
            print(json.dumps(out[-1]), flush=True)
    # rank 0's assemble of one group of 8 frames
    g = 8
    packed = torch.zeros((8, g, sharding.per_rank_pixels(W, H, TILE, TILE, 8), 4), dtype=torch.uint8, device=dev)
    images = torch.empty((g, W * H, 4), dtype=torch.uint8, device=dev)
    for _ in range(3):
        core.assemble_frames_rgba8(packed, W, H, TILE, TILE, 8, g, g * packed.shape[2], packed.shape[2], images)
    core.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        core.assemble_frames_rgba8(packed, W, H, TILE, TILE, 8, g, g * packed.shape[2], packed.shape[2], images)
    core.sync()
    print(json.dumps({"assemble_us_per_frame": round((time.perf_counter() - t0) / 50 / g * 1e6, 2)}))


if __name__ == "__main__":
    main()
