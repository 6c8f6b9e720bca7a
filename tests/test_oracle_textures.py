"""The oracle's material evaluation at a triangle hit (raytracer_unified.wgsl:279-366; texture array and sampler of
src/renderer.rs:158-259). The reference has no test or fixture for any of it (parity unpinned by the reference), so
the oracle is pinned here by answers worked out by hand from the shader text and the WebGPU sampling rules: texel
centres at integer + 0.5, linear filter, repeat addressing, layer index clamped."""
import numpy as np

from render_scenes import bundled_model

f32 = np.float32


def one_triangle(ob, material_id=0, z=-5.0):
    t = np.zeros(1, ob.TRIANGLE_DTYPE)
    t["v0"], t["v1"], t["v2"] = (-4, -4, z), (4, -4, z), (-4, 4, z)
    t["uv0"], t["uv1"], t["uv2"] = (0, 0), (1, 0), (0, 1)
    t["material_id"] = material_id
    return t


def material(ob, **kw):
    m = np.zeros(1, ob.MATERIAL_DTYPE)
    m["base_color"] = (1, 1, 1, 1)
    m["roughness"] = 1.0
    m["texture_index"] = m["normal_texture_index"] = m["emissive_texture_index"] = -1
    for k, v in kw.items():
        m[k] = v
    return m


def test_texture_array_is_built_like_the_renderers(ob):
    a = np.full((2, 3, 4), 7, np.uint8)
    b = np.full((4, 2, 4), 9, np.uint8)
    arr = ob.texture_array([a, b])
    assert arr.shape == (2, 4, 3, 4)  # layers, max height, max width
    assert np.all(arr[0, :2, :3] == 7) and np.all(arr[0, 2:] == 0)
    assert np.all(arr[1, :, :2] == 9) and np.all(arr[1, :, 2:] == 0)


def test_sampler_known_answers(ob):
    tex = np.zeros((2, 2, 4), np.uint8)
    tex[0, 0], tex[0, 1], tex[1, 0], tex[1, 1] = (255, 0, 0, 255), (0, 255, 0, 255), (0, 0, 255, 255), (255, 255, 255, 0)
    scene = ob.Scene(None, one_triangle(ob), textures=[tex, np.full((1, 1, 4), 51, np.uint8)])
    # texel centres return the texel: (0.25, 0.25) is the centre of texel (0, 0) of a 2x2 texture
    assert np.array_equal(ob.sample_texture(scene, 0.25, 0.25, 0), f32([1, 0, 0, 1]))
    assert np.array_equal(ob.sample_texture(scene, 0.75, 0.25, 0), f32([0, 1, 0, 1]))
    assert np.array_equal(ob.sample_texture(scene, 0.25, 0.75, 0), f32([0, 0, 1, 1]))
    assert np.array_equal(ob.sample_texture(scene, 0.75, 0.75, 0), f32([1, 1, 1, 0]))
    # the middle of the texture weights all four texels by 1/4
    assert np.array_equal(ob.sample_texture(scene, 0.5, 0.5, 0), f32([0.5, 0.5, 0.5, 0.75]))
    # repeat addressing: uv = 0 lies half a texel left of / above texel 0 and blends with the last texel
    assert np.array_equal(ob.sample_texture(scene, 0.0, 0.25, 0), f32([0.5, 0.5, 0, 1]))
    assert np.array_equal(ob.sample_texture(scene, 1.25, -0.75, 0), ob.sample_texture(scene, 0.25, 0.25, 0))
    assert np.array_equal(ob.sample_texture(scene, -3.5, 7.5, 0), ob.sample_texture(scene, 0.5, 0.5, 0))
    # layer 1 holds a 1x1 texture in the corner of a 2x2 layer: uv spans the layer, the rest of it is zero
    assert np.array_equal(ob.sample_texture(scene, 0.25, 0.25, 1), f32([0.2, 0.2, 0.2, 0.2]))
    assert np.array_equal(ob.sample_texture(scene, 0.75, 0.75, 1), f32([0, 0, 0, 0]))
    assert np.array_equal(ob.sample_texture(scene, 0.5, 0.25, 1), f32([0.1, 0.1, 0.1, 0.1]))
    # layer indices are clamped to the array
    assert np.array_equal(ob.sample_texture(scene, 0.25, 0.25, 9), ob.sample_texture(scene, 0.25, 0.25, 1))
    assert np.array_equal(ob.sample_texture(scene, 0.75, 0.25, -4), ob.sample_texture(scene, 0.75, 0.25, 0))
    # no textures at all: the renderer binds a 1x1 white dummy
    assert np.array_equal(ob.sample_texture(ob.Scene(None, one_triangle(ob)), 0.3, 0.9, 2), f32([1, 1, 1, 1]))


def test_alpha_from_a_texture_lets_rays_through(ob):
    """Two triangles behind each other; the front one's texture is transparent in its left half (u < 0.5)."""
    tex = np.full((1, 2, 4), 255, np.uint8)
    tex[0, 0, 3] = 0
    tris = np.concatenate([one_triangle(ob, 0, -5.0), one_triangle(ob, 1, -9.0)])
    o = f32([[-2, -3, 0], [2, -3, 0]])  # u = 0.25: centre of the transparent texel; u = 0.75: centre of the opaque one
    d = f32([[0, 0, -1], [0, 0, -1]])
    for mode, cutoff in ((2, 0.5), (1, 0.5)):
        mats = np.concatenate([material(ob, texture_index=0, alpha_mode=mode, alpha_cutoff=cutoff), material(ob)])
        hits, _ = ob.trace_rays(ob.Scene(None, tris, materials=mats, textures=[tex]), o, d)
        assert list(hits["id"]) == [1, 0] and list(hits["t"]) == [9.0, 5.0]
        occ, _ = ob.trace_rays(ob.Scene(None, tris, materials=mats, textures=[tex]), o, d, any_hit=True)
        assert list(occ["id"]) == [1, 0]
    # OPAQUE ignores the texture's alpha; MASK compares with the cutoff (alpha 0.5 at the texel border u = 0.5)
    mats = np.concatenate([material(ob, texture_index=0, alpha_mode=0), material(ob)])
    hits, _ = ob.trace_rays(ob.Scene(None, tris, materials=mats, textures=[tex]), o, d)
    assert list(hits["id"]) == [0, 0]
    o = f32([[0, -3, 0]])  # u = 0.5: halfway between the two texel centres, alpha exactly 0.5
    for cutoff, want in ((0.5, 0), (0.51, 1)):
        mats = np.concatenate([material(ob, texture_index=0, alpha_mode=1, alpha_cutoff=cutoff), material(ob)])
        hits, _ = ob.trace_rays(ob.Scene(None, tris, materials=mats, textures=[tex]), o, d[:1])
        assert hits["id"][0] == want
    # the material's own alpha multiplies the texture's (wgsl:299)
    mats = np.concatenate([material(ob, texture_index=0, alpha_mode=2, base_color=(1, 1, 1, 0.009)), material(ob)])
    hits, _ = ob.trace_rays(ob.Scene(None, tris, materials=mats, textures=[tex]), f32([[2, -3, 0]]), d[:1])
    assert hits["id"][0] == 1


def test_base_colour_normal_map_and_emissive_in_the_picture(ob):
    """One big triangle facing the camera, shaded by hand: colour = texel * factor * (0.3 + 0.7 * max(n.l, 0)) + emissive."""
    w, h = 8, 8
    cam = ob.camera_to_uniform((-1.5, -1.5, 0), np.pi, 0.0, 0.0, float(h))  # looks down -z
    tris = one_triangle(ob)
    tex = np.zeros((1, 1, 4), np.uint8)
    tex[0, 0] = (255, 128, 0, 255)
    flat = np.zeros((1, 1, 4), np.uint8)
    flat[0, 0] = (128, 128, 255, 255)  # the "no change" normal-map colour, up to 8-bit rounding
    glow = np.zeros((1, 1, 4), np.uint8)
    glow[0, 0] = (0, 51, 0, 255)
    light = -np.array([0.5, -1.0, 0.3]) / np.linalg.norm([0.5, -1.0, 0.3])
    diffuse = max(light[2], 0.0)  # geometric normal (0, 0, 1)
    centre = (h // 2) * w + w // 2

    def px(img):
        return img[centre, :3].astype(int)

    base = material(ob, texture_index=0, base_color=(0.5, 1.0, 1.0, 1.0))
    img, _ = ob.render(ob.Scene(None, tris, materials=base, textures=[tex, flat, glow]), cam, w, h)
    want = np.array([0.5, 128 / 255, 0.0]) * (0.3 + 0.7 * diffuse)
    assert np.all(np.abs(px(img) - want * 255) <= 1)
    # emissive texture scales the emissive factor
    em = material(ob, texture_index=0, base_color=(0.5, 1.0, 1.0, 1.0), emissive=(1, 1, 1), emissive_texture_index=2)
    img2, _ = ob.render(ob.Scene(None, tris, materials=em, textures=[tex, flat, glow]), cam, w, h)
    assert np.all(np.abs(px(img2) - (want + [0, 0.2, 0]) * 255) <= 1)
    # a flat normal map leaves the lighting (nearly) alone; a tilted one changes it
    nm = material(ob, texture_index=0, base_color=(0.5, 1.0, 1.0, 1.0), normal_texture_index=1)
    img3, _ = ob.render(ob.Scene(None, tris, materials=nm, textures=[tex, flat, glow]), cam, w, h)
    assert np.all(np.abs(px(img3) - px(img)) <= 1)
    tilt = np.zeros((1, 1, 4), np.uint8)
    tilt[0, 0] = (128, 255, 128, 255)  # tangent-space normal ~ (0, 1, 0): the world-space bitangent, here +y, towards the light
    img4, _ = ob.render(ob.Scene(None, tris, materials=nm, textures=[tex, tilt, glow]), cam, w, h)
    tn = np.array([128 / 255 * 2 - 1, 1.0, 128 / 255 * 2 - 1])
    tn /= np.linalg.norm(tn)
    tangent, bitangent, gn = np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0, 0, 1.0])  # cross((0,1,0), (0,0,1)) = (1,0,0)
    n = tn[0] * tangent + tn[1] * bitangent + tn[2] * gn
    n /= np.linalg.norm(n)
    want4 = np.array([0.5, 128 / 255, 0.0]) * (0.3 + 0.7 * max(float(n @ light), 0.0))
    assert np.all(np.abs(px(img4) - want4 * 255) <= 1)
    assert not np.array_equal(px(img4), px(img))


def test_bundled_models_with_their_materials(ob, car_xyz9, phoenix_xyz9):
    """The bundled models as the loader hands them over: BLEND materials whose base-colour textures have
    transparent texels open holes (fewer hits, never other than farther ones); pictures differ from the untextured
    ones. Committed as goldens so that a change shows up."""
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "textured_golden.npz"))
    for name, xyz9, pose in (("car", car_xyz9, ((200, 200, 300), 3.35, -0.28)), ("phoenix", phoenix_xyz9, ((0, 60, 160), np.pi, -0.2))):
        tris, mats, textures = bundled_model(ob, name, xyz9)
        w, h = 96, 54
        cam = ob.camera_to_uniform(*pose, 0.0, float(h))
        plain = ob.Scene(None, tris)
        full = ob.Scene(None, tris, materials=mats, textures=textures)
        p0, _, _ = ob.trace_primary(plain, cam, w, h, shadow=False)
        p1, s1, _ = ob.trace_primary(full, cam, w, h)
        both = (p0["id"] != ob.MISS) & (p1["id"] != ob.MISS)
        assert np.all(p1["t"][both] >= p0["t"][both])          # alpha can only push a hit back
        assert not np.any((p0["id"] == ob.MISS) & (p1["id"] != ob.MISS))
        img, _ = ob.render(full, cam, w, h)
        assert np.array_equal(p1, g[f"{name}_primary"]) and np.array_equal(s1, g[f"{name}_shadow"])
        assert np.array_equal(img, g[f"{name}_rgba"])
        assert (img != ob.render(plain, cam, w, h)[0]).any()
    # the phoenix's feathers are alpha-cut: holes appear
    assert (p1["id"] != p0["id"]).sum() > 0
