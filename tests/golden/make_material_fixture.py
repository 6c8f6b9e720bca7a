#!/usr/bin/env python3
"""Generates tests/golden/model_materials.npz: uv coordinates, material ids, MaterialData records and (reduced)
textures of the two bundled models, i.e. the rest of what the reference's loader hands the renderer besides
positions (src/loaders/gltf_triangles.rs:40-154, 225-237).

Run in the build container (needs /root/reference and PIL to decode the PNG files):
    python tests/golden/make_material_fixture.py

Restated from the loader:
  - one MaterialData per glTF material: base colour factor, metallic / roughness factors (glTF defaults 1), emissive
    factor, the *texture* indices of the base-colour, normal and emissive textures (-1 when absent), alpha mode
    0/1/2 and cutoff (default 0.5)                                             (:41-108)
  - one texture per glTF *image*, as RGBA8: RGB gets alpha 255; a two-channel image (the `image` crate decodes a
    grey+alpha PNG to that) becomes (c0, c1, 0, 255)                            (:116-154)
  - per triangle: TEXCOORD_0 of its three corners and the primitive's material index (0 when out of range)
                                                                               (:225-237)
Triangles come out in the same order as tests/golden/make_fixtures.py emits their positions.

The textures are REDUCED for the repository: every image of a model is cut down by keeping every k-th texel of every
k-th row, k = (the model's largest texture side) / 128, which keeps the ratio between the sizes of different textures of one
model (the renderer's texture array scales uv by the largest texture, src/renderer.rs:196-248). They are input data
for both the oracle and the GPU path, not a pinned output of the reference.
Models: "Old Rusty Car" by Renafox (CC-BY-NC-4.0), "phoenix bird" by NORBERTO-3D (CC-BY-4.0) - ATTRIBUTION.md.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"
MAX_SIDE = 128

from make_fixtures import _accessor  # noqa: E402


def load_materials(path):
    from PIL import Image

    from oracle.binding import MATERIAL_DTYPE
    g = json.load(open(path))
    base = os.path.dirname(path)
    blob = open(os.path.join(base, g["buffers"][0]["uri"]), "rb").read()

    mats = np.zeros(len(g["materials"]), MATERIAL_DTYPE)
    for i, m in enumerate(g["materials"]):
        pbr = m.get("pbrMetallicRoughness", {})
        mats[i]["base_color"] = pbr.get("baseColorFactor", [1, 1, 1, 1])
        mats[i]["metallic"] = pbr.get("metallicFactor", 1.0)
        mats[i]["roughness"] = pbr.get("roughnessFactor", 1.0)
        mats[i]["emissive"] = m.get("emissiveFactor", [0, 0, 0])
        mats[i]["texture_index"] = pbr["baseColorTexture"]["index"] if "baseColorTexture" in pbr else -1
        mats[i]["normal_texture_index"] = m["normalTexture"]["index"] if "normalTexture" in m else -1
        mats[i]["emissive_texture_index"] = m["emissiveTexture"]["index"] if "emissiveTexture" in m else -1
        mats[i]["alpha_mode"] = {"OPAQUE": 0, "MASK": 1, "BLEND": 2}[m.get("alphaMode", "OPAQUE")]
        mats[i]["alpha_cutoff"] = m.get("alphaCutoff", 0.5)

    textures = []
    for im in g.get("images", []):
        img = Image.open(os.path.join(base, im["uri"]))
        a = np.asarray(img)
        if img.mode == "RGBA":
            rgba = a
        elif img.mode == "RGB":
            rgba = np.concatenate([a, np.full(a.shape[:2] + (1,), 255, np.uint8)], axis=2)
        elif img.mode == "LA":
            rgba = np.stack([a[..., 0], a[..., 1], np.zeros_like(a[..., 0]), np.full_like(a[..., 0], 255)], axis=2)
        else:
            raise SystemExit(f"{im['uri']}: mode {img.mode} not handled")
        textures.append(rgba)
    k = max(1, max(max(t.shape[:2]) for t in textures) // MAX_SIDE) if textures else 1
    textures = [np.ascontiguousarray(t[::k, ::k]) for t in textures]

    uvs, ids = [], []

    def node(i):
        n = g["nodes"][i]
        if "mesh" in n:
            for prim in g["meshes"][n["mesh"]]["primitives"]:
                npos = g["accessors"][prim["attributes"]["POSITION"]]["count"]
                if "TEXCOORD_0" in prim["attributes"]:
                    acc = g["accessors"][prim["attributes"]["TEXCOORD_0"]]
                    assert acc["componentType"] == 5126, "normalised integer uv: restate into_f32 first"
                    uv = _accessor(g, blob, prim["attributes"]["TEXCOORD_0"]).astype(np.float32)
                else:
                    uv = np.zeros((npos, 2), np.float32)
                if "indices" in prim:
                    idx = _accessor(g, blob, prim["indices"]).reshape(-1).astype(np.int64)
                    idx = idx[:len(idx) // 3 * 3].reshape(-1, 3)
                else:
                    idx = np.arange(npos // 3 * 3).reshape(-1, 3)
                uvs.append(uv[idx].reshape(-1, 6))
                mid = prim.get("material", 0)
                ids.append(np.full(len(idx), mid if mid < len(mats) else 0, np.uint32))
        for c in n.get("children", []):
            node(c)

    for s in g["scenes"]:
        for r in s["nodes"]:
            node(r)
    return np.concatenate(uvs).astype(np.float32), np.concatenate(ids), mats, textures


def main():
    out = {}
    for name, sub, tris in (("car", "models/no_animation", "car_tris.npy"), ("phoenix", "models/animated", "phoenix_tris.npy")):
        uv, mid, mats, tex = load_materials(os.path.join(REF, sub, "scene.gltf"))
        assert len(uv) == len(np.load(os.path.join(HERE, tris))), "triangle order differs from make_fixtures.py"
        out[f"{name}_uv"], out[f"{name}_material_id"] = uv, mid
        out[f"{name}_materials"] = mats.view(np.uint8).reshape(len(mats), 64)
        out[f"{name}_n_textures"] = np.array(len(tex))
        for i, t in enumerate(tex):
            out[f"{name}_texture_{i}"] = t
        print(name, uv.shape, np.bincount(mid), [t.shape for t in tex])
        print(mats)
    np.savez_compressed(os.path.join(HERE, "model_materials.npz"), **out)
    print("wrote model_materials.npz", os.path.getsize(os.path.join(HERE, "model_materials.npz")), "bytes")


if __name__ == "__main__":
    main()
