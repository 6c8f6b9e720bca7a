"""The N > 1 path on CPU: two processes, gloo backend. The sharding logic (tile dealing, one gather of
[primary | shadow] blocks into rank 0, frame assembly, scene broadcast) is the product's; the tracer and the
assemble kernel are replaced by CPU stand-ins (the oracle and a numpy scatter) because there is no GPU here.
The same FrameSharder drives librt_mi355.so on the GPU box (tests/test_gpu_parity.py covers the kernels)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, TW, TH = 150, 85, 32, 16


class OracleCore:
    """Stands in for ray_tracer_amd.TraceCore on CPU tensors."""

    def __init__(self, ob, scene):
        self.ob, self.scene = ob, scene

    def trace_primary(self, cam, width, height, shard=None, shadow=True, out_primary=None, out_shadow=None):
        osh = self.ob.Shard(shard.tile_w, shard.tile_h, shard.first, shard.stride, shard.owner_list()) if shard is not None else None
        p, s, _ = self.ob.trace_primary(self.scene, cam, width, height, shard=osh, shadow=shadow, n_threads=2)
        out_primary[:len(p)] = torch.from_numpy(p.view(np.int32).reshape(-1, 2))
        if shadow:
            out_shadow[:len(s)] = torch.from_numpy(s.view(np.int32).reshape(-1, 2))

    def render(self, cam, width, height, shard=None, max_bounces=8, shadows=False, out=None):
        osh = self.ob.Shard(shard.tile_w, shard.tile_h, shard.first, shard.stride, shard.owner_list()) if shard is not None else None
        img, _ = self.ob.render(self.scene, cam, width, height, shard=osh, max_bounces=max_bounces, shadows=shadows,
                                n_threads=2)
        out[:len(img)] = torch.from_numpy(img)

    def render_frames(self, cams, width, height, shard=None, max_bounces=8, shadows=False, out=None, rgb8=False,
                      frame_stride=0):
        assert frame_stride in (0, out.shape[1])  # frames are frame_stride pixels apart in `out`
        for j, cam in enumerate(cams):
            if rgb8:
                tmp = torch.zeros((out.shape[1], 4), dtype=torch.uint8)
                self.render(cam, width, height, shard=shard, max_bounces=max_bounces, shadows=shadows, out=tmp)
                out[j] = tmp[:, :3]
            else:
                self.render(cam, width, height, shard=shard, max_bounces=max_bounces, shadows=shadows, out=out[j])

    def assemble_frames_rgb8(self, gathered, width, height, tile_w, tile_h, n_ranks, n_frames, rank_stride, frame_stride,
                             images):
        flat = gathered.reshape(-1, 3)
        rgb = torch.empty((width * height, 3), dtype=torch.uint8)
        for j in range(n_frames):
            self._assemble(flat[j * frame_stride:], width, height, tile_w, tile_h, n_ranks, rgb, rank_stride, torch.uint8, 3)
            images[j, :, :3] = rgb
            images[j, :, 3] = 255

    def assemble_frames_rgba8(self, gathered, width, height, tile_w, tile_h, n_ranks, n_frames, rank_stride, frame_stride,
                              images):
        flat = gathered.reshape(-1, 4)
        for j in range(n_frames):
            self._assemble(flat[j * frame_stride:], width, height, tile_w, tile_h, n_ranks, images[j], rank_stride, torch.uint8, 4)

    def assemble_frames(self, gathered, rgb8, width, height, tile_w, tile_h, n_ranks, owner, n_frames, rank_stride,
                        frame_stride, images):
        per = 3 if rgb8 else 4
        flat = gathered.reshape(-1, per)
        if images.dim() == 2:
            images = images.unsqueeze(0)
        pic = torch.empty((width * height, per), dtype=torch.uint8)
        for j in range(n_frames):
            self._assemble(flat[j * frame_stride:], width, height, tile_w, tile_h, n_ranks, pic, rank_stride, torch.uint8, per,
                           owner)
            images[j, :, :per] = pic
            images[j, :, 3] = 255 if rgb8 else images[j, :, 3]

    def set_stream(self, stream):
        pass

    def tile_flags(self, cam, width, height, tile_w, tile_h):
        """Stand-in for rt_tile_flags: a tile without a primary hit shows nothing but sky (the GPU's mask is more
        cautious, never less)."""
        p, _, _ = self.ob.trace_primary(self.scene, cam, width, height, shadow=False, n_threads=2)
        hit = (p["id"] != self.ob.MISS).reshape(height, width)
        tiles_x, tiles_y = -(-width // tile_w), -(-height // tile_h)
        flags = np.zeros(tiles_x * tiles_y, np.uint8)
        for ty in range(tiles_y):
            for tx in range(tiles_x):
                flags[ty * tiles_x + tx] = hit[ty * tile_h:(ty + 1) * tile_h, tx * tile_w:(tx + 1) * tile_w].any()
        return flags

    def assemble_frames_sky(self, gathered, rgb8, width, height, tile_w, tile_h, n_ranks, owner, cams, rank_stride, frame_stride,
                            images):
        SKY = 0xFFFFFFFF
        per = 3 if rgb8 else 4
        flat = gathered.reshape(-1, per)
        tiles_x = -(-width // tile_w)
        for j, cam in enumerate(cams):
            sky, _ = self.ob.render(self.scene, cam, width, height, max_bounces=1, n_threads=2)  # its missed pixels are the sky
            count = {}
            for t, r in enumerate(owner):
                x0, y0 = (t % tiles_x) * tile_w, (t // tiles_x) * tile_h
                if r != SKY:
                    lt = count.get(r, 0)
                    count[r] = lt + 1
                for y in range(y0, min(height, y0 + tile_h)):
                    for x in range(x0, min(width, x0 + tile_w)):
                        if r == SKY:
                            images[j, y * width + x] = torch.from_numpy(sky[y * width + x].copy())
                        else:
                            src = j * frame_stride + r * rank_stride + lt * tile_w * tile_h + (y - y0) * tile_w + (x - x0)
                            images[j, y * width + x, :per] = flat[src]
                            if rgb8:
                                images[j, y * width + x, 3] = 255

    def assemble_tiles_rgba8(self, gathered, width, height, tile_w, tile_h, n_ranks, image, rank_stride=0):
        self._assemble(gathered, width, height, tile_w, tile_h, n_ranks, image, rank_stride, torch.uint8, 4)

    def assemble_tiles(self, gathered, width, height, tile_w, tile_h, n_ranks, image, rank_stride=0):
        self._assemble(gathered, width, height, tile_w, tile_h, n_ranks, image, rank_stride, torch.int32, 2)

    def _assemble(self, gathered, width, height, tile_w, tile_h, n_ranks, image, rank_stride, dtype, per, owner=None):
        # `gathered` is a view starting at rank 0's block; ranks are rank_stride records apart
        tiles_x = -(-width // tile_w)
        base = gathered.untyped_storage()
        flat = torch.empty(0, dtype=dtype).set_(base).view(-1, per)
        off = gathered.storage_offset() // per
        for y in range(height):
            for x in range(width):
                tile = (y // tile_h) * tiles_x + x // tile_w
                r, lt = tile % n_ranks, tile // n_ranks
                if owner is not None:  # owner[o]: the rank of tile o of every round; its tiles are packed in tile order
                    q, o = divmod(tile, len(owner))
                    r = owner[o]
                    lt = q * owner.count(r) + owner[:o].count(r)
                src = off + r * rank_stride + lt * tile_w * tile_h + (y % tile_h) * tile_w + x % tile_w
                image[y * width + x] = flat[src]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding as ob
        from ray_tracer_amd import sharding
        xyz9 = np.load(os.path.join(ROOT, "tests", "golden", "car_tris.npy")) if rank == 0 else None
        tris0 = ob.triangles_from_xyz9(xyz9) if rank == 0 else None
        boxes0 = np.zeros(0, ob.BOX_DTYPE) if rank == 0 else None
        boxes, tris = sharding.broadcast_scene(boxes0, tris0, "cpu")
        assert len(tris) == 3132 and len(boxes) == 0
        scene = ob.Scene(None, tris.view(ob.TRIANGLE_DTYPE))
        cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, float(H))
        sh = sharding.FrameSharder(OracleCore(ob, scene), W, H, TW, TH, device="cpu", shadow=True)
        assert sh.per_rank == sharding.per_rank_pixels(W, H, TW, TH, world)
        out = sh.trace(cam)
        if rank == 0:
            want_p, want_s, _ = ob.trace_primary(scene, cam, W, H, n_threads=2)
            got_p = out[0].numpy().view(ob.HIT_DTYPE).reshape(-1)
            got_s = out[1].numpy().view(ob.HIT_DTYPE).reshape(-1)
            ok = (np.array_equal(got_p["id"], want_p["id"]) and np.array_equal(got_p["t"].view(np.uint32), want_p["t"].view(np.uint32))
                  and np.array_equal(got_s["id"], want_s["id"]) and np.array_equal(got_s["t"].view(np.uint32), want_s["t"].view(np.uint32)))
            hits = int((want_p["id"] != ob.MISS).sum())
            open(os.path.join(tmp, "result"), "w").write(f"{int(ok)} {hits}")
        else:
            assert out is None
        # the shaded picture, two frames in flight (buffers alternate), second frame from another pose
        shr = sharding.FrameSharder(OracleCore(ob, scene), W, H, TW, TH, device="cpu", shadow=True, mode="rgba",
                                    frames_in_flight=2, max_bounces=2)
        cam2 = ob.camera_to_uniform((0, 150, 50), 1.0, -0.4, 0.0, float(H))
        pics = [shr.trace(c) for c in (cam, cam2, cam)]
        if rank == 0:
            ok2 = True
            for c, pic in ((cam2, pics[1]), (cam, pics[2])):  # pics[0]'s buffer has been reused by the third frame
                want, _ = ob.render(scene, c, W, H, max_bounces=2, shadows=True, n_threads=2)
                ok2 = ok2 and np.array_equal(pic.numpy(), want)
            ok2 = ok2 and pics[0] is pics[2] and pics[0] is not pics[1]
            open(os.path.join(tmp, "result_rgba"), "w").write(str(int(ok2)))
        # groups of frames: one launch, one gather, one assemble per group; a full group, then a partial one
        # unequal shares (the gathering rank renders less); the 30 tiles are not a whole number of rounds of 3 or 8
        weights = [1, 2] if world == 2 else [2, 3, 3]
        shg = sharding.FrameSharder(OracleCore(ob, scene), W, H, TW, TH, device="cpu", shadow=False, mode="rgba", max_bounces=1,
                                    weights=weights)
        assert shg.shard.first == rank and shg.shard.stride == sum(weights) and shg.owner.count(rank) == weights[rank]
        shg.enable_groups(group=3, groups_in_flight=2, rgb8=(world == 3))  # one world size per wire format
        cams = [cam, cam2, ob.camera_to_uniform((150, 120, 250), 3.4, -0.3, 0.0, float(H)), cam2, cam]
        first, second = shg.trace_group(cams[:3]), shg.trace_group(cams[3:])
        if rank == 0:
            ok3 = first.shape == (3, W * H, 4) and second.shape == (2, W * H, 4)
            for c, pic in zip(cams, list(first) + list(second)):
                want, _ = ob.render(scene, c, W, H, max_bounces=1, n_threads=2)
                ok3 = ok3 and np.array_equal(pic.numpy(), want)
            open(os.path.join(tmp, "result_groups"), "w").write(str(int(ok3)))
        else:
            assert first is None and second is None
        # sky tiles stay at home: announced cameras travel without them, an unannounced one travels whole
        far = ob.camera_to_uniform((900, 500, 1200), 3.7, -0.3, 0.0, float(H))  # the model small in the frame: mostly sky
        shs = sharding.FrameSharder(OracleCore(ob, scene), W, H, TW, TH, device="cpu", shadow=False, mode="rgba", max_bounces=1,
                                    weights=weights)
        shs.enable_groups(group=2, groups_in_flight=2, rgb8=(world == 2))
        shs.enable_sparse()
        shs.prepare_cameras([cam, far])
        dense_block = shs.group * shs.per_rank * shs.wire_bytes
        a = shs.trace_group([far, cam])
        sparse_block = shs.last_wire_bytes
        b = shs.trace_group([far, cam2])  # cam2 was not announced
        whole_block = shs.last_wire_bytes
        c = shs.trace_group([far])
        if rank == 0:
            ok4 = sparse_block < dense_block and whole_block == dense_block
            for cm, pic in ((far, a[0]), (cam, a[1]), (far, b[0]), (cam2, b[1]), (far, c[0])):
                want, _ = ob.render(scene, cm, W, H, max_bounces=1, n_threads=2)
                ok4 = ok4 and np.array_equal(pic.numpy(), want)
            open(os.path.join(tmp, "result_sparse"), "w").write(f"{int(ok4)} {sparse_block} {dense_block}")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_frame_sharding_two_ranks_gloo(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ok, hits = open(tmp_path / "result").read().split()
    assert ok == "1" and int(hits) > 1000
    assert open(tmp_path / "result_rgba").read() == "1"
    assert open(tmp_path / "result_groups").read() == "1"
    ok, sparse_block, dense_block = open(tmp_path / "result_sparse").read().split()
    assert ok == "1" and int(sparse_block) < int(dense_block), (sparse_block, dense_block)


def test_tile_bookkeeping():
    sys.path.insert(0, ROOT)
    from ray_tracer_amd import RtShard, shard_pixels, sharding
    for (w, h, tw, th) in ((1920, 1080, 64, 64), (150, 85, 32, 16), (7680, 4320, 64, 64), (10, 10, 64, 64)):
        nt = sharding.n_tiles(w, h, tw, th)
        for world in (1, 2, 3, 4, 8):
            per = sharding.per_rank_pixels(w, h, tw, th, world)
            mine = [shard_pixels(w, h, RtShard(tw, th, r, world)) for r in range(world)]
            assert max(mine) == per or nt < world  # the padded block holds the busiest rank's tiles
            assert sum(mine) == nt * tw * th       # every tile is traced exactly once


def test_bench_starts_its_own_ranks_when_run_plainly():
    """`python bench.py --gpus 2` without a launcher (WORLD_SIZE unset) must start its ranks itself and still print
    exactly one JSON line on stdout; under torch.distributed.run it must not start a second set. RT_BENCH_DRY_RUN stops
    each rank after the rendezvous and one collective, so this runs without a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["RT_BENCH_DRY_RUN"] = "1"
    bench = os.path.join(ROOT, "bench.py")
    plain = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "7", "--warmup", "3"], env=env, capture_output=True,
                           text=True, timeout=300)
    assert plain.returncode == 0, plain.stderr[-3000:]
    lines = [ln for ln in plain.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, plain.stdout
    got = json.loads(lines[0])
    passes = (got.pop("preheat_passes_min"), got.pop("preheat_passes_max"))
    assert got == {"dry_run": True, "n_gpus": 2, "ranks_seen": 2, "steps": 7, "warmup": 3}
    # the untimed pre-heat passes (each holds a collective) are counted alike on every rank, however fast a rank is
    assert passes[0] == passes[1] >= 2, passes
    launched = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                               "127.0.0.1", "--master-port", str(_free_port()), bench, "--gpus", "2", "--steps", "5", "--warmup", "1"],
                              env=env, capture_output=True, text=True, timeout=300)
    assert launched.returncode == 0, launched.stderr[-3000:]
    lines = [ln for ln in launched.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["ranks_seen"] == 2, launched.stdout
    # a world size that disagrees with --gpus is refused, loudly
    wrong = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True,
                           text=True, timeout=300)
    assert wrong.returncode != 0 and "WORLD_SIZE" in wrong.stderr


class _CommCore:
    """Stands in for TraceCore's communicator calls: rank `bad` cannot bind RCCL."""

    def __init__(self, rank, bad):
        self.rank, self.bad, self.inited = rank, bad, None

    def comm_unique_id(self):
        if self.rank == self.bad:
            raise RuntimeError("RCCL not found")
        return bytes([self.rank + 1]) * 128

    def comm_init(self, comm_id, rank, world):
        self.inited = (bytes(comm_id), rank, world)


def _comm_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ray_tracer_amd import sharding
        good = _CommCore(rank, bad=-1)
        sharding.init_comm(good, rank, world)
        assert good.inited == (bytes([1]) * 128, rank, world)  # rank 0's id on every rank
        broken = _CommCore(rank, bad=1)
        try:
            sharding.init_comm(broken, rank, world)
            raised = ""
        except RuntimeError as e:
            raised = str(e)
        assert "rank 1" in raised and broken.inited is None  # EVERY rank raises, none is left waiting in a collective
        dist.barrier()
        open(os.path.join(tmp, f"ok{rank}"), "w").write("1")
    finally:
        dist.destroy_process_group()


def test_a_rank_without_rccl_stops_every_rank_together(tmp_path):
    mp.spawn(_comm_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
