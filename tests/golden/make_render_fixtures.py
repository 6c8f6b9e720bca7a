#!/usr/bin/env python3
"""Writes tests/golden/render_golden.npz: small shaded pictures from the CPU oracle (orc_render), so that later
changes to the oracle or to the GPU path are noticed. Needs only the oracle and the committed car fixture."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from render_scenes import demo_boxes, demo_materials  # noqa: E402


def main():
    from oracle import binding as ob
    gold = {}
    tris = ob.pyramid_triangles()
    tris["material_id"] = [0, 1, 2, 7, 0, 1]  # 7: no such material -> the shader's grey
    cam = ob.camera_to_uniform((0, 8, 20), np.pi, -0.5, 0.0, 36.0)
    gold["pyramid_rgba"], _ = ob.render(ob.Scene(None, tris, cell_size=1.0), cam, 64, 36, materials=demo_materials(ob))
    car = ob.Scene(None, ob.triangles_from_xyz9(np.load(os.path.join(HERE, "car_tris.npy"))))
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, 36.0)
    gold["car_rgba"], _ = ob.render(car, cam, 64, 36)
    gold["car_rgba_shadows"], _ = ob.render(car, cam, 64, 36, shadows=True)
    boxes = demo_boxes(ob)
    cam = ob.camera_to_uniform((4, 12, 40), np.pi, -0.25, 0.4, 54.0)
    gold["boxes_rgba"], st = ob.render(ob.Scene(boxes, None, cell_size=4.0), cam, 96, 54)
    print("boxes stats", st)
    np.savez_compressed(os.path.join(HERE, "render_golden.npz"), **gold)
    print("wrote", sorted(gold), {k: v.shape for k, v in gold.items()})


if __name__ == "__main__":
    main()
