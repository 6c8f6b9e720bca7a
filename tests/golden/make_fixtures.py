#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.

Run in the build container (needs /root/reference for the two bundled models):
    python tests/golden/make_fixtures.py

1. car_tris.npy / phoenix_tris.npy - the bundled glTF models flattened to world-space
   triangles, float32 (n, 9) = v0,v1,v2. This restates what the reference's loader does
   with positions (src/loaders/gltf_triangles.rs:179-283): walk the scene's node tree,
   global = parent * local (column-major 4x4), transform_point3 every POSITION, then emit
   one triangle per index triple. Every node above a mesh carries `matrix` (or nothing); only
   skeleton joints use TRS. Arithmetic is float32 in glam's scalar order; glam's SIMD Mat4 on the
   author's machine may round the last bit differently, so these vertices are INPUT data
   for both the oracle and the GPU path, not a pinned output of the reference.
   Models: "Old Rusty Car" by Renafox (CC-BY-NC-4.0), "phoenix bird" by NORBERTO-3D
   (CC-BY-4.0) - see tests/golden/ATTRIBUTION.md.
2. *_golden.npz - outputs of the CPU oracle (oracle/) on small cases, kept so that a
   change in either the oracle or the GPU path shows up as a diff against history:
   grid cell lists, 64x36 hit images, ray-list hits. The reference holds no golden vectors
   for these (it has no test for grid, traversal or ray generation).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
f32 = np.float32

_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _accessor(g, blob, idx):
    a = g["accessors"][idx]
    bv = g["bufferViews"][a["bufferView"]]
    dt = np.dtype(_COMP[a["componentType"]])
    nc = _NCOMP[a["type"]]
    off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    stride = bv.get("byteStride", dt.itemsize * nc)
    out = np.empty((a["count"], nc), dt)
    for k in range(nc):
        out[:, k] = np.ndarray((a["count"],), dt, blob, off + k * dt.itemsize, (stride,))
    return out


def _mat_mul(a, b):
    """column-major 4x4 product a*b in float32, column by column: a.x*b.x + a.y*b.y + a.z*b.z + a.w*b.w."""
    out = np.empty((4, 4), f32)  # out[c] = column c
    for c in range(4):
        r = a[0] * b[c][0]
        r = r + a[1] * b[c][1]
        r = r + a[2] * b[c][2]
        r = r + a[3] * b[c][3]
        out[c] = r
    return out


def _trs_matrix(n):
    t = np.array(n.get("translation", [0, 0, 0]), f32)
    x, y, z, w = (f32(v) for v in n.get("rotation", [0, 0, 0, 1]))
    sc = np.array(n.get("scale", [1, 1, 1]), f32)
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)],
                  [2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w)],
                  [2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)]], f32)  # rows = columns
    m = np.zeros((4, 4), f32)
    m[0, :3], m[1, :3], m[2, :3], m[3, :3], m[3, 3] = r[0] * sc[0], r[1] * sc[1], r[2] * sc[2], t, 1
    return m


def _transform_points(m, p):
    """transform_point3: ((x_axis*x + y_axis*y) + z_axis*z) + w_axis, float32."""
    r = m[0][None, :3] * p[:, 0:1]
    r = r + m[1][None, :3] * p[:, 1:2]
    r = r + m[2][None, :3] * p[:, 2:3]
    r = r + m[3][None, :3]
    return r.astype(f32)


def load_gltf_triangles(path):
    g = json.load(open(path))
    blob = open(os.path.join(os.path.dirname(path), g["buffers"][0]["uri"]), "rb").read()
    tris = []

    trs_seen = []

    def node(i, parent, under_trs=False):
        n = g["nodes"][i]
        if any(k in n for k in ("translation", "rotation", "scale")):
            local = _trs_matrix(n)  # only skeleton joints in the bundled models; no mesh hangs below one
            trs_seen.append(i)
        else:
            local = np.array(n.get("matrix", np.eye(4).reshape(-1)), f32).reshape(4, 4)  # rows here = columns
        glob = _mat_mul(parent, local)
        if "mesh" in n:
            assert not under_trs, "mesh below a TRS node: restate the gltf crate's decomposed transform first"
            for prim in g["meshes"][n["mesh"]]["primitives"]:
                assert prim.get("mode", 4) == 4
                pos = _transform_points(glob, _accessor(g, blob, prim["attributes"]["POSITION"]).astype(f32))
                if "indices" in prim:
                    idx = _accessor(g, blob, prim["indices"]).reshape(-1).astype(np.int64)
                    idx = idx[:len(idx) // 3 * 3].reshape(-1, 3)
                else:
                    idx = np.arange(len(pos) // 3 * 3).reshape(-1, 3)
                tris.append(pos[idx].reshape(-1, 9))
        under_trs = under_trs or i in trs_seen
        for c in n.get("children", []):
            node(c, glob, under_trs)

    for s in g["scenes"]:
        for r in s["nodes"]:
            node(r, np.eye(4, dtype=f32))
    return np.concatenate(tris).astype(f32)


def main():
    from oracle import binding as ob

    if os.path.isdir(REF):
        car = load_gltf_triangles(os.path.join(REF, "models/no_animation/scene.gltf"))
        phoenix = load_gltf_triangles(os.path.join(REF, "models/animated/scene.gltf"))
        np.save(os.path.join(HERE, "car_tris.npy"), car)
        np.save(os.path.join(HERE, "phoenix_tris.npy"), phoenix)
    car = np.load(os.path.join(HERE, "car_tris.npy"))
    phoenix = np.load(os.path.join(HERE, "phoenix_tris.npy"))
    print("car", car.shape, "phoenix", phoenix.shape)

    gold = {}
    # pyramid scene, its start camera (src/scenes/pyramid.rs:22-61, src/camera.rs:64)
    pyr = ob.Scene(None, ob.pyramid_triangles())
    cam = ob.camera_to_uniform((0, 8, 20), np.pi, -0.5, 0.0, 36.0)
    p, s, _ = ob.trace_primary(pyr, cam, 64, 36)
    gold["pyramid_cell_start"], gold["pyramid_indices"] = pyr.grid.cell_start, pyr.grid.indices
    gold["pyramid_primary"], gold["pyramid_shadow"] = p, s
    # a finer grid so that traversal actually steps
    pyr2 = ob.Scene(None, ob.pyramid_triangles(), cell_size=1.0)
    p, s, _ = ob.trace_primary(pyr2, cam, 64, 36)
    gold["pyramid_c1_cell_start"], gold["pyramid_c1_indices"] = pyr2.grid.cell_start, pyr2.grid.indices
    gold["pyramid_c1_primary"], gold["pyramid_c1_shadow"] = p, s

    # car, `SCENE=gltf` start camera (src/camera.rs:63)
    car_scene = ob.Scene(None, ob.triangles_from_xyz9(car))
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, 36.0)
    p, s, st = ob.trace_primary(car_scene, cam, 64, 36)
    gold["car_dims"] = np.array(car_scene.grid.meta.fine_dims, np.uint32)
    gold["car_bounds"] = np.array(list(car_scene.grid.meta.bounds_min) + list(car_scene.grid.meta.bounds_max), f32)
    gold["car_cell_start"], gold["car_indices"] = car_scene.grid.cell_start, car_scene.grid.indices
    gold["car_primary"], gold["car_shadow"] = p, s
    print("car grid", car_scene.grid.meta.fine_dims, car_scene.grid.n_refs, "64x36 stats", st)

    ph_scene = ob.Scene(None, ob.triangles_from_xyz9(phoenix))
    gold["phoenix_dims"] = np.array(ph_scene.grid.meta.fine_dims, np.uint32)
    gold["phoenix_cell_start"], gold["phoenix_indices"] = ph_scene.grid.cell_start, ph_scene.grid.indices
    print("phoenix grid", ph_scene.grid.meta.fine_dims, ph_scene.grid.n_refs)

    # reference ray generator against its triangle generator (src/core/benchmark.rs:99-140)
    tri = ob.generate_test_triangles(10000)
    o, d = ob.generate_test_rays(1000)
    sc = ob.Scene(None, tri)
    hits, _ = ob.trace_rays(sc, o, d)
    gold["bench10k_rays1000_hits"] = hits
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **gold)
    print("wrote", sorted(gold))


if __name__ == "__main__":
    main()
