#!/usr/bin/env python3
"""Generates tests/golden/textured_golden.npz: the CPU oracle's hit records and pictures of the two bundled models
with their materials and (reduced) textures at 96x54. Outputs of the oracle, kept so that a change in the oracle
shows up as a diff against history; the reference holds no image or hit fixture.
    python tests/golden/make_textured_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from oracle import binding as ob
    from render_scenes import bundled_model
    out = {}
    for name, pose in (("car", ((200, 200, 300), 3.35, -0.28)), ("phoenix", ((0, 60, 160), np.pi, -0.2))):
        tris, mats, textures = bundled_model(ob, name, np.load(os.path.join(HERE, f"{name}_tris.npy")))
        scene = ob.Scene(None, tris, materials=mats, textures=textures)
        cam = ob.camera_to_uniform(*pose, 0.0, 54.0)
        p, s, st = ob.trace_primary(scene, cam, 96, 54)
        img, _ = ob.render(scene, cam, 96, 54)
        out[f"{name}_primary"], out[f"{name}_shadow"], out[f"{name}_rgba"] = p, s, img
        print(name, st)
    np.savez_compressed(os.path.join(HERE, "textured_golden.npz"), **out)


if __name__ == "__main__":
    main()
