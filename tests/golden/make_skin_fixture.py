#!/usr/bin/env python3
"""Writes tests/golden/phoenix_skin.npz: the skin and the animation of the reference's bundled animated model
(models/animated, "phoenix bird" by NORBERTO-3D, CC-BY-4.0, see ATTRIBUTION.md) as plain arrays - vertex
positions, joint indices and weights, triangle indices, inverse bind matrices, the node tree with its default
transforms, and every animation channel's key times and values. Data only; needs /root/reference.

The reference does not skin (its loader flattens the bind pose, src/loaders/gltf_triangles.rs:179-283), so
these arrays are inputs for this build's own skinning kernel and its CPU restatement."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_fixtures import _accessor  # noqa: E402

REF = "/root/reference/models/animated/scene.gltf"


def main():
    g = json.load(open(REF))
    blob = open(os.path.join(os.path.dirname(REF), g["buffers"][0]["uri"]), "rb").read()
    nn = len(g["nodes"])
    parent = np.full(nn, -1, np.int32)
    has_trs = np.zeros(nn, np.uint8)
    local = np.tile(np.eye(4, dtype=np.float32).reshape(-1), (nn, 1))  # column-major
    trs = np.tile(np.array([0, 0, 0, 0, 0, 0, 1, 1, 1, 1], np.float32), (nn, 1))  # T, R (xyzw), S
    for i, n in enumerate(g["nodes"]):
        for c in n.get("children", []):
            parent[c] = i
        if any(k in n for k in ("translation", "rotation", "scale")):
            has_trs[i] = 1
            trs[i, 0:3] = n.get("translation", [0, 0, 0])
            trs[i, 3:7] = n.get("rotation", [0, 0, 0, 1])
            trs[i, 7:10] = n.get("scale", [1, 1, 1])
        elif "matrix" in n:
            local[i] = n["matrix"]
    skin = g["skins"][0]
    joints_nodes = np.array(skin["joints"], np.int32)
    ibm = _accessor(g, blob, skin["inverseBindMatrices"]).astype(np.float32)  # (J, 16) column-major
    pos, jnt, wgt, idx, mesh_of_vertex = [], [], [], [], []
    base = 0
    for i, n in enumerate(g["nodes"]):
        if "mesh" not in n:
            continue
        assert n.get("skin", 0) == 0
        for prim in g["meshes"][n["mesh"]]["primitives"]:
            a = prim["attributes"]
            p = _accessor(g, blob, a["POSITION"]).astype(np.float32)
            pos.append(p)
            jnt.append(_accessor(g, blob, a["JOINTS_0"]).astype(np.uint16))
            wgt.append(_accessor(g, blob, a["WEIGHTS_0"]).astype(np.float32))
            t = _accessor(g, blob, prim["indices"]).reshape(-1).astype(np.int64)
            idx.append((t[:len(t) // 3 * 3] + base).reshape(-1, 3).astype(np.uint32))
            base += len(p)
    anim = g["animations"][0]
    c_node, c_path, c_off, times, values = [], [], [0], [], []
    for ch in anim["channels"]:
        s = anim["samplers"][ch["sampler"]]
        assert s["interpolation"] == "LINEAR"
        t = _accessor(g, blob, s["input"]).reshape(-1).astype(np.float32)
        v = _accessor(g, blob, s["output"]).astype(np.float32)
        path = {"translation": 0, "rotation": 1, "scale": 2}[ch["target"]["path"]]
        assert len(t) == len(v) and v.shape[1] == (4 if path == 1 else 3)
        c_node.append(ch["target"]["node"])
        c_path.append(path)
        vv = np.zeros((len(t), 4), np.float32)
        vv[:, :v.shape[1]] = v
        times.append(t)
        values.append(vv)
        c_off.append(c_off[-1] + len(t))
    out = dict(node_parent=parent, node_has_trs=has_trs, node_local=local, node_trs=trs, skin_joints=joints_nodes,
               inverse_bind=ibm, positions=np.concatenate(pos), joints=np.concatenate(jnt), weights=np.concatenate(wgt),
               tri_indices=np.concatenate(idx), chan_node=np.array(c_node, np.int32), chan_path=np.array(c_path, np.uint8),
               chan_offset=np.array(c_off, np.int64), key_times=np.concatenate(times), key_values=np.concatenate(values))
    np.savez_compressed(os.path.join(HERE, "phoenix_skin.npz"), **out)
    print({k: v.shape for k, v in out.items()}, "duration", float(out["key_times"].max()))


if __name__ == "__main__":
    main()
