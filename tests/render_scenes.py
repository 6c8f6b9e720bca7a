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
