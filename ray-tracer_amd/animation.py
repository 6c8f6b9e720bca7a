"""Host side of skeletal animation for the skinning kernel (rt_set_skin / rt_skin_vertices): samples the key
frames of a glTF-style animation and turns the node tree into one 4x4 matrix per joint.

The reference has no skinning (its loader flattens the bind pose, src/loaders/gltf_triangles.rs:179-283); the
rules here are glTF 2.0's: LINEAR interpolation (spherical for rotations, along the shorter arc), local = T*R*S,
global = parent * local, joint matrix = global(joint) * inverse_bind(joint); a skinned mesh ignores its own
node's transform. The matrices are INPUT to both the GPU kernel and its CPU restatement, so evaluating them in
float64 here does not enter any parity comparison."""
import numpy as np


def _trs_matrix(t, r, s):
    x, y, z, w = r
    m = np.eye(4)
    m[:3, 0] = np.array([1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w)]) * s[0]
    m[:3, 1] = np.array([2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w)]) * s[1]
    m[:3, 2] = np.array([2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)]) * s[2]
    m[:3, 3] = t
    return m


def _slerp(a, b, u):
    d = float(np.dot(a, b))
    if d < 0.0:
        b, d = -b, -d
    if d > 0.9995:
        q = a + (b - a) * u
    else:
        th = np.arccos(d)
        q = (np.sin((1 - u) * th) * a + np.sin(u * th) * b) / np.sin(th)
    return q / np.linalg.norm(q)


class SkinnedModel:
    """Arrays of tests/golden/phoenix_skin.npz (or any file with the same keys)."""

    def __init__(self, npz):
        d = np.load(npz) if isinstance(npz, str) else npz
        self.parent = d["node_parent"]
        self.has_trs = d["node_has_trs"].astype(bool)
        self.local = d["node_local"].astype(np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)  # column-major in the file
        self.trs = d["node_trs"].astype(np.float64)
        self.joints_nodes = d["skin_joints"]
        self.inverse_bind = d["inverse_bind"].astype(np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)
        self.positions = np.ascontiguousarray(d["positions"], np.float32)
        self.joints = np.ascontiguousarray(d["joints"], np.uint16)
        self.weights = np.ascontiguousarray(d["weights"], np.float32)
        self.tri_indices = np.ascontiguousarray(d["tri_indices"], np.uint32)
        self.chan_node, self.chan_path, self.chan_offset = d["chan_node"], d["chan_path"], d["chan_offset"]
        self.key_times, self.key_values = d["key_times"].astype(np.float64), d["key_values"].astype(np.float64)
        self.duration = float(self.key_times.max())
        order, seen = [], set()
        for i in range(len(self.parent)):  # parents before children
            chain = []
            while i != -1 and i not in seen:
                chain.append(i)
                seen.add(i)
                i = int(self.parent[i])
            order.extend(reversed(chain))
        self.order = order

    def bind_pose_triangles(self):
        """(nt, 9) float32 triangles of the unskinned vertex positions (mesh space)."""
        return self.positions[self.tri_indices.astype(np.int64)].reshape(-1, 9)

    def joint_matrices(self, time):
        """(n_joints, 16) float32, column-major, for `time` seconds into the animation (clamped to its range)."""
        trs = self.trs.copy()
        for c in range(len(self.chan_node)):
            a, b = int(self.chan_offset[c]), int(self.chan_offset[c + 1])
            t, v = self.key_times[a:b], self.key_values[a:b]
            k = int(np.searchsorted(t, time, side="right")) - 1
            path = int(self.chan_path[c])
            width = 4 if path == 1 else 3
            if k < 0:
                val = v[0, :width]
            elif k >= len(t) - 1:
                val = v[-1, :width]
            else:
                u = (time - t[k]) / (t[k + 1] - t[k])
                val = _slerp(v[k], v[k + 1], u) if path == 1 else v[k, :3] + (v[k + 1, :3] - v[k, :3]) * u
            n = int(self.chan_node[c])
            if path == 0:
                trs[n, 0:3] = val
            elif path == 1:
                trs[n, 3:7] = val
            else:
                trs[n, 7:10] = val
        glob = np.empty((len(self.parent), 4, 4))
        for i in self.order:
            loc = _trs_matrix(trs[i, 0:3], trs[i, 3:7], trs[i, 7:10]) if self.has_trs[i] else self.local[i]
            p = int(self.parent[i])
            glob[i] = loc if p < 0 else glob[p] @ loc
        jm = np.einsum("jab,jbc->jac", glob[self.joints_nodes], self.inverse_bind)
        return np.ascontiguousarray(jm.transpose(0, 2, 1).reshape(-1, 16), np.float32)  # back to column-major
