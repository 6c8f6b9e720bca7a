"""rt_params.repack_candidates: the (ray, candidate) pairs of a wave dealt out again to its lanes. The reference tests a
cell's list in order, one ray per invocation (raytracer_unified.wgsl:531-556); here the pairs of 64 rays meet per ray
through a minimum over (t, list position), which must give the in-order scan's answer bit for bit. Checked against the
oracle and against the plain kernel: scenes whose neighbouring rays sit in different cells (random triangles on a fine
grid), scenes whose rays share their cells (the bundled models), both constant sets, with and without shadow rays, frames
off the 8-pixel raster, shards, a step limit that bites, origins on the far bounds, many equal t (the benchmark scene's
repeated triangles with duplicate removal off), and the entry point that allocates nothing."""
import math

import numpy as np
import pytest

from test_gpu_parity import assert_hits_equal, ocam, oparams

pytestmark = pytest.mark.gpu


def pair(rt, tris, cell_size=0.0, **kw):
    cores = []
    for repack in (1, 0):
        c = rt.TraceCore(0, rt.default_params(repack_candidates=repack, **kw))
        c.set_scene(None, tris)
        c.build_grid(cell_size=cell_size)
        cores.append(c)
    return cores


def same(on, off, cam, w, h, what, shard=None, shadow=True):
    p1, s1 = on.trace_primary(cam, w, h, shard=shard, shadow=shadow)
    p0, s0 = off.trace_primary(cam, w, h, shard=shard, shadow=shadow)
    assert_hits_equal(p1, p0, f"{what}: primary, pairs dealt out against the plain loop")
    if shadow:
        assert_hits_equal(s1, s0, f"{what}: shadow, pairs dealt out against the plain loop")
    return p1, s1


def random_soup(n, seed, spread=50.0, size=2.5):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (n, 1, 3)).astype(np.float32)
    return (c + rng.normal(0, size, (n, 3, 3)).astype(np.float32)).astype(np.float32).reshape(n, 9)


@pytest.mark.parametrize("wgsl", [0, 1])
def test_rays_in_different_cells_equal_the_oracle(rt, ob, wgsl):
    xyz9 = random_soup(20000, 11)
    on, off = pair(rt, rt.triangles_from_xyz9(xyz9), cell_size=4.0, wgsl_compat=wgsl)
    try:
        scene = ob.Scene(None, ob.triangles_from_xyz9(xyz9), cell_size=4.0)
        for name, pos, yaw, pitch, (w, h) in (("inside", (0, 0, 0), 0.4, 0.1, (203, 117)),
                                               ("outside", (0, 20, 140), math.pi, -0.1, (256, 144)),
                                               ("far, small in the frame", (300, 200, 400), 3.8, -0.35, (160, 96))):
            cam = rt.Camera(pos, yaw, pitch).to_uniform(0.0, float(h))
            for shadow in (False, True):
                p, s = same(on, off, cam, w, h, f"{name} shadow={shadow}", shadow=shadow)
            assert (p["id"] != rt.RT_MISS).sum() > 100, name
            op, os_, _ = ob.trace_primary(scene, ocam(ob, cam), w, h, params=oparams(ob, on.params), n_threads=8)
            assert_hits_equal(p, op, f"{name}: primary against the oracle")
            assert_hits_equal(s, os_, f"{name}: shadow against the oracle")
    finally:
        on.close(), off.close()


def test_counting_build_of_the_dealing_kernel(rt):
    """collect_stats = 2 with repack_candidates on runs the counting build of k_trace_primary_repack: the records are
    the uncounted kernel's, rays and hits are the plain counting kernel's, and the fetch counters describe the dealing
    kernel (no more reference records than the plain loop with its look-ahead asks for, no fewer cell records than the
    plain loop behind the frame's block mask)."""
    xyz9 = random_soup(20000, 5)
    on, off = pair(rt, rt.triangles_from_xyz9(xyz9), cell_size=4.0)
    try:
        w, h = 256, 144
        for pos, yaw, pitch in (((0, 0, 0), 0.4, 0.1), ((0, 20, 140), math.pi, -0.1)):
            cam = rt.Camera(pos, yaw, pitch).to_uniform(0.0, float(h))
            p, s = same(on, off, cam, w, h, "uncounted")
            got = {}
            for name, core in (("on", on), ("off", off)):
                core.set_collect_stats(2)
                pc, sc = core.trace_primary(cam, w, h)
                got[name] = core.stats()
                core.set_collect_stats(0)
                assert_hits_equal(pc, p, f"{name}: primary records of the counting build")
                assert_hits_equal(sc, s, f"{name}: shadow records of the counting build")
            a, b = got["on"], got["off"]
            for k in ("rays", "primary_hits", "shadow_rays", "shadow_hits"):
                assert a[k] == b[k], (k, a[k], b[k])
            assert a["rays"] == w * h + int((p["id"] != rt.RT_MISS).sum())
            assert a["cell_records_fetched"] >= b["cell_records_fetched"] > 0
            assert 0 < a["ref_records_fetched"] + a["ref_records_staged"] <= b["ref_records_fetched"] + b["ref_records_staged"] * 64
    finally:
        on.close(), off.close()


def test_bundled_models_shards_and_step_limit(rt, ob, car_xyz9, phoenix_xyz9):
    for xyz9, pos, yaw, pitch in ((car_xyz9, (200, 200, 300), 3.35, -0.28), (phoenix_xyz9, (-300, 250, 600), 2.7, -0.2)):
        for kw in ({}, {"max_dda_steps": 12}, {"wgsl_compat": 1}):
            on, off = pair(rt, rt.triangles_from_xyz9(xyz9), **kw)
            try:
                w, h = 333, 187
                cam = rt.Camera(pos, yaw, pitch).to_uniform(0.0, float(h))
                p, s = same(on, off, cam, w, h, f"whole frame {kw}")
                assert (p["id"] != rt.RT_MISS).sum() > 1000
                for (tw, th, first, stride) in ((32, 32, 1, 3), (17, 9, 0, 2)):
                    same(on, off, cam, w, h, f"tiles {tw}x{th} {kw}", shard=rt.RtShard(tw, th, first, stride))
                if not kw:  # and against the oracle
                    scene = ob.Scene(None, ob.triangles_from_xyz9(xyz9))
                    op, os_, _ = ob.trace_primary(scene, ocam(ob, cam), w, h, params=oparams(ob, on.params), n_threads=8)
                    assert_hits_equal(p, op, "primary against the oracle")
                    assert_hits_equal(s, os_, "shadow against the oracle")
            finally:
                on.close(), off.close()


def test_origins_on_the_far_bounds_and_equal_t(rt):
    # a camera exactly on the grid's upper bounds starts outside the grid's cells (the traversal's literal path)
    xyz9 = random_soup(4000, 5, spread=20.0, size=2.0)
    on, off = pair(rt, rt.triangles_from_xyz9(xyz9), cell_size=2.0)
    try:
        meta = on.grid_info()[0]
        top = tuple(float(v) for v in meta.bounds_max[:3])
        for pos in (top, (top[0], 0.0, 0.0), (0.0, top[1], top[2])):
            for yaw, pitch in ((3.9, -0.6), (0.7, 0.2)):
                cam = rt.Camera(pos, yaw, pitch).to_uniform(0.0, 90.0)
                same(on, off, cam, 160, 90, f"origin {pos}")
    finally:
        on.close(), off.close()
    # the benchmark generator repeats its positions: with duplicate removal off a cell's list holds the same triangle
    # many times, every copy with the same t - the earliest list position must win
    from ray_tracer_amd import workloads
    on, off = pair(rt, workloads.generate_test_triangles(30000), cell_size=4.0, skip_duplicate_triangles=0)
    try:
        cam = rt.Camera((0, 0, 75), math.pi, 0.0).to_uniform(0.0, 135.0)
        p, _ = same(on, off, cam, 240, 135, "repeated triangles")
        assert (p["id"] != rt.RT_MISS).sum() > 1000
    finally:
        on.close(), off.close()


def test_the_entry_point_that_allocates_nothing_deals_pairs_out_too(rt):
    import torch
    xyz9 = random_soup(8000, 3)
    on, off = pair(rt, rt.triangles_from_xyz9(xyz9), cell_size=4.0)
    try:
        w, h = 200, 120
        cam = rt.Camera((0, 0, 0), 1.0, 0.0).to_uniform(0.0, float(h))
        p0, s0 = off.trace_primary(cam, w, h)
        out = [torch.empty((w * h, 2), dtype=torch.int32, device="cuda") for _ in range(2)]
        on.trace_primary_noalloc(cam, w, h, out[0], out[1])
        on.sync()
        p1 = out[0].cpu().numpy().view(rt.HIT_DTYPE).reshape(-1)
        s1 = out[1].cpu().numpy().view(rt.HIT_DTYPE).reshape(-1)
        assert_hits_equal(p1, p0, "primary")
        assert_hits_equal(s1, s0, "shadow")
    finally:
        on.close(), off.close()


def test_pictures_whose_bounce_rays_share_their_lists_equal_the_plain_ones(rt, ob, car_xyz9, phoenix_xyz9):
    """rt_render with reflection bounces (k_render_deal) against rt_render_opts.plain_bounces = 1 (k_render) and the
    oracle: the bundled models with their materials and textures (texture-decided alpha included), the same with every
    material opaque, and without materials; 2 and 8 bounces, with and without the shadow term, shards."""
    from render_scenes import bundled_model
    for name, xyz9, pos, yaw, pitch in (("car", car_xyz9, (200, 200, 300), 3.35, -0.28), ("phoenix", phoenix_xyz9, (-300, 250, 600), 2.7, -0.2)):
        tris, mats, textures = bundled_model(ob, name, xyz9)
        opaque = mats.copy()
        opaque["alpha_mode"] = 0
        mirror = mats.copy()  # every surface reflects: long paths
        mirror["metallic"] = 0.6
        for what, m, tex in (("as loaded", mats, textures), ("opaque", opaque, textures), ("mirrors", mirror, textures), ("no materials", None, None)):
            core = rt.TraceCore(0)
            try:
                core.set_scene(None, tris)
                core.set_materials(m)
                if tex is not None:
                    core.set_textures(tex)
                core.build_grid()
                w, h = 240, 135
                cam = rt.Camera(pos, yaw, pitch).to_uniform(0.0, float(h))
                for bounces in (2, 8):
                    for shadows in (False, True):
                        a = core.render(cam, w, h, max_bounces=bounces, shadows=shadows)
                        b = core.render(cam, w, h, max_bounces=bounces, shadows=shadows, plain_bounces=True)
                        assert np.array_equal(a, b), f"{name} {what}: {bounces} bounces, shadows {shadows}"
                shard = rt.RtShard(32, 32, 1, 3)
                assert np.array_equal(core.render(cam, w, h, shard=shard), core.render(cam, w, h, shard=shard, plain_bounces=True)), f"{name} {what}: shard"
                if what in ("mirrors", "no materials"):  # ("as loaded" against the oracle: tests/test_gpu_textures.py)
                    scene = ob.Scene(None, tris, materials=m, textures=tex) if m is not None else ob.Scene(None, tris)
                    want, _ = ob.render(scene, ocam(ob, cam), w, h, params=oparams(ob, core.params), max_bounces=8, shadows=True)
                    got = core.render(cam, w, h, max_bounces=8, shadows=True)
                    assert np.array_equal(got, want), f"{name} {what}: against the oracle"
            finally:
                core.close()


@pytest.mark.parametrize("any_hit", [False, True])
def test_ray_lists_in_no_order(rt, ob, any_hit):
    """rt_trace_rays with repack_candidates: random origins and directions (a list no caller sorted), rays that miss the
    grid, axis-parallel and zero directions, a count that is no multiple of 64 - against the plain kernel and the
    oracle."""
    xyz9 = random_soup(15000, 21)
    on, off = pair(rt, rt.triangles_from_xyz9(xyz9), cell_size=4.0)
    try:
        rng = np.random.default_rng(4)
        n = 20000 + 37
        o = rng.uniform(-70, 70, (n, 3)).astype(np.float32)
        d = rng.normal(0, 1, (n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[:6] = [(1, 0, 0), (0, -1, 0), (0, 0, 1), (0, 0, 0), (1, 1, 0), (0, 1e-30, 1)]
        o[6:12] = [(200, 200, 200), (-200, 0, 0), (0, 0, 0), (55, 0, 0), (0, 300, 0), (10, 10, -300)]
        a = on.trace_rays(o, d, any_hit=any_hit)
        b = off.trace_rays(o, d, any_hit=any_hit)
        assert_hits_equal(a, b, "pairs dealt out against the plain loop")
        assert (a["id"] != rt.RT_MISS).sum() > 1000
        scene = ob.Scene(None, ob.triangles_from_xyz9(xyz9), cell_size=4.0)
        want = ob.trace_rays(scene, o, d, any_hit=any_hit, n_threads=8)
        want = want[0] if isinstance(want, tuple) else want
        assert_hits_equal(a, want, "against the oracle")
    finally:
        on.close(), off.close()
