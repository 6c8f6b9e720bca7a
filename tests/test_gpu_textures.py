"""GPU parity for the material part of the triangle test (raytracer_unified.wgsl:279-366): base-colour textures,
alpha MASK / BLEND decided per hit by a texture, normal maps, emissive textures - hit records (ids, f32 bit patterns
of t) and the shaded picture byte for byte against the oracle, through the C ABI."""
import ctypes as C
import math

import numpy as np
import pytest

from render_scenes import bundled_model, random_textured_materials, random_textures
from test_gpu_fuzz import random_scene
from test_oracle_textures import material, one_triangle

pytestmark = pytest.mark.gpu
f32 = np.float32


def same_hits(a, b):
    return np.array_equal(a["id"], b["id"]) and np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))


def ocam(ob, cam):
    oc = ob.CameraUniform()
    C.memmove(C.byref(oc), C.byref(cam), C.sizeof(oc))
    return oc


def test_alpha_from_a_texture(rt, ob, core):
    tex = np.full((1, 2, 4), 255, np.uint8)
    tex[0, 0, 3] = 0
    tris = np.concatenate([one_triangle(ob, 0, -5.0), one_triangle(ob, 1, -9.0)])
    o = f32([[-2, -3, 0], [2, -3, 0], [0, -3, 0]])
    d = f32([[0, 0, -1]] * 3)
    core.set_scene(None, tris)
    core.build_grid()
    core.set_textures([tex])
    for mode, cutoff in ((2, 0.5), (1, 0.5), (1, 0.51), (0, 0.5)):
        mats = np.concatenate([material(ob, texture_index=0, alpha_mode=mode, alpha_cutoff=cutoff), material(ob)])
        core.set_materials(mats)  # after the grid: the records are patched in place
        scene = ob.Scene(None, tris, materials=mats, textures=[tex])
        for any_hit in (False, True):
            want, _ = ob.trace_rays(scene, o, d, any_hit=any_hit)
            assert same_hits(core.trace_rays(o, d, any_hit=any_hit), want), (mode, cutoff, any_hit)
    assert list(core.trace_rays(o, d)["id"]) == [0, 0, 0]  # OPAQUE was last
    mats = np.concatenate([material(ob, texture_index=0, alpha_mode=2), material(ob)])
    core.set_materials(mats)
    assert list(core.trace_rays(o, d)["id"]) == [1, 0, 0]
    core.set_textures(None)  # back to the white dummy: alpha 1 everywhere
    assert list(core.trace_rays(o, d)["id"]) == [0, 0, 0]


@pytest.mark.parametrize("name,pose", [("car", ((200, 200, 300), 3.35, -0.28)), ("phoenix", ((0, 60, 160), math.pi, -0.2))])
def test_bundled_models_with_materials_and_textures(rt, ob, core, car_xyz9, phoenix_xyz9, name, pose):
    tris, mats, textures = bundled_model(ob, name, car_xyz9 if name == "car" else phoenix_xyz9)
    core.set_scene(None, tris)
    core.set_materials(mats)
    core.set_textures(textures)
    core.build_grid()
    scene = ob.Scene(None, tris, materials=mats, textures=textures)
    for w, h, wgsl in ((96, 54, 0), (640, 360, 0), (320, 180, 1)):
        c2 = rt.TraceCore(0, rt.default_params(wgsl_compat=wgsl))
        c2.set_scene(None, tris)
        c2.build_grid()
        c2.set_textures(textures)   # textures and materials after the grid, in the other order
        c2.set_materials(mats)
        cam = rt.Camera(*pose).to_uniform(0.0, float(h))
        params = ob.Params.default(wgsl_compat=wgsl)
        op, os_, st = ob.trace_primary(scene, ocam(ob, cam), w, h, params=params)
        p, s = c2.trace_primary(cam, w, h)
        assert same_hits(p, op) and same_hits(s, os_), (name, w, h, wgsl)
        for bounces, shadows in ((8, False), (2, True)):
            img = c2.render(cam, w, h, max_bounces=bounces, shadows=shadows)
            want, _ = ob.render(scene, ocam(ob, cam), w, h, params=params, max_bounces=bounces, shadows=shadows)
            assert np.array_equal(img, want), (name, w, h, wgsl, bounces, shadows, int((img != want).any(axis=1).sum()))
        c2.set_collect_stats(True)
        p2, s2 = c2.trace_primary(cam, w, h)
        got = c2.stats()
        assert same_hits(p2, op) and same_hits(s2, os_)
        assert {k: got[k] for k in st} == st
        c2.close()
    # the committed goldens of the oracle
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "textured_golden.npz"))
    cam = rt.Camera(*pose).to_uniform(0.0, 54.0)
    p, s = core.trace_primary(cam, 96, 54)
    assert np.array_equal(p, g[f"{name}_primary"]) and np.array_equal(s, g[f"{name}_shadow"])
    assert np.array_equal(core.render(cam, 96, 54), g[f"{name}_rgba"])


def test_random_textured_scenes(rt, ob):
    import os
    seed, trials = int(os.environ.get("RT_FUZZ_SEED", "777")), int(os.environ.get("RT_FUZZ_TRIALS", "32"))
    rng = np.random.default_rng(seed)
    rays = pixels = wide = 0
    for trial in range(trials):
        boxes, tris, ext = random_scene(ob, rng)
        if tris is None:
            continue
        n_tex = int(rng.integers(1, 5))
        textures = random_textures(rng, n_tex)
        mats = random_textured_materials(ob, rng, int(rng.integers(1, 7)), n_tex)  # ids 0..5: some out of range -> grey
        for k in ("uv0", "uv1", "uv2"):
            tris[k] = rng.uniform(-2.5, 2.5, (len(tris), 2))
        kw = dict(wgsl_compat=int(rng.integers(0, 2)), max_dda_steps=int(rng.choice([200, 200, 23])),
                  max_objects_per_cell=int(rng.choice([8192, 8192, 5])))
        cell = float(rng.choice([16.0, 4.0, 1.0])) * (ext / 40.0 if rng.random() < 0.5 else 1.0)
        core = rt.TraceCore(0, rt.default_params(**kw))
        core.set_scene(boxes, tris)
        core.set_materials(mats)
        if rng.random() < 0.1:
            textures = None  # materials name textures but none were given: the white dummy
        core.set_textures(textures)
        try:
            _, _, n_refs = core.build_grid(cell_size=cell)
        except rt.RtError:
            core.close()
            continue
        wide += (n_refs + 1) * 48 >= 2 ** 32  # beyond what a buffer descriptor addresses: the kernels' plain-pointer form
        scene = ob.Scene(boxes, tris, materials=mats, textures=textures, cell_size=cell)
        params = ob.Params.default(**kw)
        for view in range(2):
            inside = rng.random() < 0.4
            pos = rng.uniform(-ext, ext, 3) * (0.5 if inside else 3.0)
            yaw, pitch = rng.uniform(0, 2 * math.pi), rng.uniform(-1.2, 1.2)
            if not inside:
                yaw = math.atan2(-pos[0], -pos[2])
                pitch = math.atan2(-pos[1], math.hypot(pos[0], pos[2]))
            w, h = int(rng.integers(33, 130)), int(rng.integers(17, 80))
            cam = rt.Camera(tuple(pos), yaw, pitch).to_uniform(float(rng.uniform(0, 3)), float(h))
            oc = ocam(ob, cam)
            p, s = core.trace_primary(cam, w, h)
            op, os_, st = ob.trace_primary(scene, oc, w, h, params=params)
            assert same_hits(p, op) and same_hits(s, os_), f"trial {trial} view {view}: records differ ({kw}, cell {cell})"
            rays += st["rays"]
            o, d = ob.primary_rays(oc, w, h, params=params)
            for any_hit in (False, True):
                want, _ = ob.trace_rays(scene, o, d, params=params, cam=oc, any_hit=any_hit)
                assert same_hits(core.trace_rays(o, d, cam=cam, any_hit=any_hit), want), f"trial {trial}: ray list, any_hit={any_hit}"
            bounces, shadows = int(rng.choice([1, 3, 8])), bool(rng.random() < 0.5)
            img = core.render(cam, w, h, max_bounces=bounces, shadows=shadows)
            want, _ = ob.render(scene, oc, w, h, params=params, max_bounces=bounces, shadows=shadows)
            assert np.array_equal(img, want), f"trial {trial} view {view}: {int((img != want).any(axis=1).sum())} pixels differ"
            pixels += w * h
        core.close()
    print(f"random textured scenes: {rays} rays and {pixels} shaded pixels compared, 0 mismatches")
    assert rays > 100000 and (wide >= 1 or seed != 777)


def test_set_textures_rejects_bad_input(rt, core):
    L = core._L
    assert L.rt_set_textures(core._h, None, 2) == rt._lib.RT_ERR_INVALID
    arr = (rt._lib.RtTexture * 1)()
    arr[0].rgba8, arr[0].width, arr[0].height = None, 4, 4
    assert L.rt_set_textures(core._h, arr, 1) == rt._lib.RT_ERR_INVALID
    t = np.zeros((2, 2, 4), np.uint8)
    arr[0].rgba8, arr[0].width, arr[0].height = t.ctypes.data, 0, 2
    assert L.rt_set_textures(core._h, arr, 1) == rt._lib.RT_ERR_INVALID
    assert b"texture 0" in L.rt_last_error(core._h)
    with pytest.raises(ValueError):
        core.set_textures([np.zeros((2, 2, 3), np.uint8)])
