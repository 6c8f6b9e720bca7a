"""The oracle's shaded picture (orc_render): committed goldens and properties that follow from the shader text."""
import os

import numpy as np

from render_scenes import demo_boxes, demo_materials

HERE = os.path.dirname(os.path.abspath(__file__))


def golden():
    return np.load(os.path.join(HERE, "golden", "render_golden.npz"))


def test_oracle_reproduces_committed_pictures(ob, car_xyz9):
    g = golden()
    tris = ob.pyramid_triangles()
    tris["material_id"] = [0, 1, 2, 7, 0, 1]
    cam = ob.camera_to_uniform((0, 8, 20), np.pi, -0.5, 0.0, 36.0)
    img, _ = ob.render(ob.Scene(None, tris, cell_size=1.0), cam, 64, 36, materials=demo_materials(ob))
    assert np.array_equal(img, g["pyramid_rgba"])
    car = ob.Scene(None, ob.triangles_from_xyz9(car_xyz9))
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, 36.0)
    assert np.array_equal(ob.render(car, cam, 64, 36)[0], g["car_rgba"])
    assert np.array_equal(ob.render(car, cam, 64, 36, shadows=True)[0], g["car_rgba_shadows"])
    cam = ob.camera_to_uniform((4, 12, 40), np.pi, -0.25, 0.4, 54.0)
    assert np.array_equal(ob.render(ob.Scene(demo_boxes(ob), None, cell_size=4.0), cam, 96, 54)[0], g["boxes_rgba"])


def test_sky_lighting_and_shadows(ob, car_xyz9):
    car = ob.Scene(None, ob.triangles_from_xyz9(car_xyz9))
    w, h = 96, 54
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, float(h))
    img, st = ob.render(car, cam, w, h)
    prim, _, _ = ob.trace_primary(car, cam, w, h, shadow=False)
    miss = prim["id"] == ob.MISS
    assert st["rays"] == w * h and st["primary_hits"] == (~miss).sum()  # grey, non-reflective: one ray per pixel
    # sky: mix((.3,.5,.7), (.5,.7,1), (d.y+1)/2) (raytracer_unified.wgsl:591-596)
    _, d = ob.primary_rays(cam, w, h)
    t = (d[:, 1] + np.float32(1.0)) * np.float32(0.5)
    for k, (a, b) in enumerate(((0.3, 0.5), (0.5, 0.7), (0.7, 1.0))):
        want = np.float32(a) * (np.float32(1.0) - t) + np.float32(b) * t
        assert np.array_equal(img[miss, k], np.rint(np.clip(want[miss], 0, 1) * np.float32(255)).astype(np.uint8))
    assert np.all(img[:, 3] == 255)
    # lit grey: 0.7 * (0.3 + 0.7 * diffuse) lies in [0.21, 0.7] -> bytes 54..178, all channels equal
    hit = img[~miss]
    assert np.all(hit[:, 0] == hit[:, 1]) and np.all(hit[:, 1] == hit[:, 2]) and hit[:, 0].min() >= 54 and hit[:, 0].max() <= 179
    dark, st2 = ob.render(car, cam, w, h, shadows=True)
    assert st2["shadow_rays"] == st["primary_hits"] and 0 < st2["shadow_hits"] < st2["shadow_rays"]
    assert np.all(dark <= img) and (dark < img).any(axis=1).sum() > 0
    assert np.array_equal(dark[miss], img[miss])


def test_bounces_debug_pixel_grid_overlay_and_shards(ob):
    boxes = demo_boxes(ob)
    scene = ob.Scene(boxes, None, cell_size=4.0)
    w, h = 96, 54
    cam = ob.camera_to_uniform((4, 12, 40), np.pi, -0.25, 0.4, float(h))
    full, st = ob.render(scene, cam, w, h)
    one, st1 = ob.render(scene, cam, w, h, max_bounces=1)
    assert st["rays"] > st1["rays"] == w * h and (full != one).any()
    dbg, _ = ob.render(scene, cam, w, h, debug_pixel=(40, 20))
    diff = np.flatnonzero((dbg != full).any(axis=1))
    ys, xs = diff // w, diff % w
    assert tuple(dbg[20 * w + 40]) == (255, 255, 0, 255) and np.all(np.abs(xs - 40) <= 1) and np.all(np.abs(ys - 20) <= 1)
    gcam = ob.camera_to_uniform((4, 12, 40), np.pi, -0.25, 0.4, float(h), show_grid=True)
    grid, _ = ob.render(scene, gcam, w, h)
    changed = (grid != full).any(axis=1)
    assert changed.sum() > 50 and np.all(grid[changed][:, 1] >= full[changed][:, 1])  # mixed towards green
    # shards tile the frame
    tw = th = 32
    tx, ty = (w + tw - 1) // tw, (h + th - 1) // th
    canvas = np.zeros((h, w, 4), np.uint8)
    for rank in range(3):
        part, _ = ob.render(scene, cam, w, h, shard=ob.Shard(tw, th, rank, 3))
        part = part.reshape(-1, th, tw, 4)
        for lt, tile in enumerate(range(rank, tx * ty, 3)):
            y0, x0 = (tile // tx) * th, (tile % tx) * tw
            canvas[y0:y0 + th, x0:x0 + tw] = part[lt][: h - y0, : w - x0]
    assert np.array_equal(canvas.reshape(-1, 4), full)


def alpha_materials(ob):
    """0 opaque, 1 MASK below its cutoff (invisible), 2 MASK above it, 3 BLEND nearly transparent (invisible),
    4 MASK, textured, factor alpha below the cutoff: with no texture array the renderer binds a white dummy, so
    alpha = 1 * 0.1 and the material is invisible too; with textures it depends on the texel."""
    m = np.zeros(5, ob.MATERIAL_DTYPE)
    m["texture_index"] = m["normal_texture_index"] = m["emissive_texture_index"] = -1
    m["roughness"] = 1.0
    m["base_color"] = [(0.8, 0.2, 0.2, 1.0), (0.2, 0.8, 0.2, 0.3), (0.2, 0.2, 0.8, 0.7), (0.8, 0.8, 0.2, 0.005), (0.5, 0.5, 0.5, 0.1)]
    m["alpha_mode"] = [0, 1, 1, 2, 1]
    m["alpha_cutoff"] = [0.5, 0.5, 0.5, 0.5, 0.5]
    m["texture_index"][4] = 0
    return m


def test_alpha_rules_make_triangles_invisible(ob, car_xyz9):
    tris = ob.triangles_from_xyz9(car_xyz9)
    tris["material_id"] = np.arange(len(tris)) % 6  # 5: no such material
    mats = alpha_materials(ob)
    scene = ob.Scene(None, tris, materials=mats)
    plain = ob.Scene(None, tris)
    w, h = 160, 90
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, float(h))
    p, s, _ = ob.trace_primary(scene, cam, w, h)
    p0, _, _ = ob.trace_primary(plain, cam, w, h)
    hit = p["id"] != ob.MISS
    assert not np.isin(p["id"][hit] % 6, (1, 3, 4)).any() and np.isin(p0["id"][p0["id"] != ob.MISS] % 6, (1, 3, 4)).any()
    sh = s["id"] != ob.MISS
    assert not np.isin(s["id"][sh] % 6, (1, 3, 4)).any()
    # an opaque texel (alpha 255 * 0.1 is still below the cutoff; raise the factor) brings material 4 back
    mats2 = mats.copy()
    mats2["base_color"][4][3] = 0.9
    p2, _, _ = ob.trace_primary(ob.Scene(None, tris, materials=mats2, textures=[np.full((2, 2, 4), 255, np.uint8)]), cam, w, h)
    assert np.isin(p2["id"][p2["id"] != ob.MISS] % 6, (4,)).any()
    # traversal agrees with testing every triangle
    o, d = ob.primary_rays(cam, w, h)
    brute = ob.trace_rays_bruteforce(scene, o, d)
    assert np.array_equal(brute["id"], p["id"]) and np.array_equal(brute["t"].view(np.uint32), p["t"].view(np.uint32))
    # the picture sees through them too
    img, _ = ob.render(scene, cam, w, h)
    img0, _ = ob.render(plain, cam, w, h, materials=mats)
    assert (img != img0).any()
