"""Small scenes for the shaded-picture tests (also used by tests/golden/make_render_fixtures.py)."""
import numpy as np


def demo_materials(ob):
    m = np.zeros(3, ob.MATERIAL_DTYPE)
    m["texture_index"] = m["normal_texture_index"] = m["emissive_texture_index"] = -1
    m["roughness"] = 1.0
    m["base_color"] = [(0.8, 0.2, 0.2, 1.0), (0.2, 0.6, 0.9, 1.0), (0.9, 0.9, 0.9, 1.0)]
    m["emissive"][1] = (0.1, 0.05, 0.0)
    m["metallic"][2] = 0.6
    return m


def demo_boxes(ob):
    """A floor, a mirror wall and a few coloured boxes, the last one moving (BoxData::create_moving_box)."""
    b = np.zeros(6, ob.BOX_DTYPE)
    spec = [((-30, -2, -30), (30, 0, 30), (0.6, 0.6, 0.6), 0.3), ((-30, 0, -32), (30, 25, -30), (0.9, 0.9, 0.95), 0.8),
            ((-8, 0, -8), (-2, 6, -2), (0.9, 0.2, 0.2), 0.0), ((2, 0, -6), (7, 9, -1), (0.2, 0.8, 0.3), 0.5),
            ((-3, 0, 4), (3, 3, 9), (0.2, 0.3, 0.9), 0.0)]
    for i, (lo, hi, col, refl) in enumerate(spec):
        b[i] = ob.make_box(np.array(lo, np.float32), np.array(hi, np.float32))
        b[i]["color"], b[i]["reflectivity"] = col, refl
    c0, c1, h = np.array((10, 4, 5), np.float32), np.array((14, 8, 2), np.float32), np.array((1.5, 1.5, 1.5), np.float32)
    b[5] = ob.make_box(np.minimum(c0, c1) - h - np.float32(0.5), np.maximum(c0, c1) + h + np.float32(0.5), center0=c0, center1=c1, half=h)
    b[5]["color"], b[5]["reflectivity"] = (0.9, 0.8, 0.1), 0.2
    return b


def bundled_model(ob, name, xyz9):
    """(TriangleData, MaterialData, textures) of a bundled model as the reference's loader produces them
    (src/loaders/gltf_triangles.rs): positions from <name>_tris.npy, the rest from model_materials.npz
    (textures reduced; see tests/golden/make_material_fixture.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_materials.npz"))
    tris = ob.triangles_from_xyz9(xyz9)
    uv = g[f"{name}_uv"]
    tris["uv0"], tris["uv1"], tris["uv2"] = uv[:, 0:2], uv[:, 2:4], uv[:, 4:6]
    tris["material_id"] = g[f"{name}_material_id"]
    mats = np.ascontiguousarray(g[f"{name}_materials"]).view(ob.MATERIAL_DTYPE).reshape(-1)
    textures = [g[f"{name}_texture_{i}"] for i in range(int(g[f"{name}_n_textures"]))]
    return tris, mats, textures


def random_textures(rng, n):
    """n RGBA8 textures of different sizes: smooth colour gradients plus noise, alpha in patches around the BLEND
    threshold and the usual MASK cutoffs, so that alpha rules cut holes with texel-sized detail."""
    out = []
    for _ in range(n):
        h, w = int(rng.choice([1, 3, 8, 16, 37])), int(rng.choice([1, 4, 8, 16, 29]))
        t = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
        t[..., 3] = rng.choice([0, 1, 2, 3, 90, 127, 128, 200, 255], (h, w))
        out.append(t)
    return out


def random_textured_materials(ob, rng, n, n_textures):
    m = np.zeros(n, ob.MATERIAL_DTYPE)
    m["base_color"] = rng.uniform(0.2, 1.0, (n, 4))
    m["base_color"][:, 3] = rng.choice([1.0, 1.0, 0.7, 0.004], n)
    m["emissive"] = rng.choice([0.0, 0.0, 0.5], (n, 3))
    m["metallic"] = rng.choice([0.0, 0.0, 0.5, 1.0], n)
    m["roughness"] = 1.0
    # layer indices beyond the array are clamped to its last layer by the sampler
    m["texture_index"] = rng.choice([-1, 0, 1, n_textures - 1, n_textures + 3], n)
    m["normal_texture_index"] = rng.choice([-1, -1, 0, n_textures - 1], n)
    m["emissive_texture_index"] = rng.choice([-1, -1, 0, n_textures - 1], n)
    m["alpha_mode"] = rng.choice([0, 1, 2, 2], n)
    m["alpha_cutoff"] = rng.choice([0.5, 0.3, 0.9], n)
    return m
