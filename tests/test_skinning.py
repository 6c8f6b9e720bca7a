"""Skeletal animation: the host-side key-frame sampling, the oracle's skinning, and (gpu) the skinning kernel
feeding the binning pass, on the skin of the reference's bundled animated model (tests/golden/phoenix_skin.npz)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def model():
    from ray_tracer_amd.animation import SkinnedModel
    return SkinnedModel(os.path.join(HERE, "golden", "phoenix_skin.npz"))


def test_fixture_is_a_proper_skin(model):
    assert model.positions.shape == (3485, 3) and model.tri_indices.shape == (4064, 3)
    assert np.allclose(model.weights.sum(axis=1), 1.0, atol=1e-6) and int(model.joints.max()) < len(model.joints_nodes)
    assert 3.0 < model.duration < 4.0


def test_joint_matrices_are_rigid_and_periodic(model):
    a, b = model.joint_matrices(0.0), model.joint_matrices(model.duration)
    assert a.shape == (89, 16) and np.allclose(a, b, atol=5e-2)  # the clip loops (translations are hundreds of units)
    mid = model.joint_matrices(0.4 * model.duration)
    assert not np.allclose(a, mid, atol=1e-2)
    assert np.all(np.abs(a.reshape(-1, 4, 4)[:, :, 3] - (0, 0, 0, 1)) < 1e-6)  # affine: last row 0 0 0 1


def test_oracle_skinning_matches_float64_and_moves_the_wings(model, ob):
    for t in (0.0, 0.9, 2.1):
        jm = model.joint_matrices(t)
        got = ob.skin_vertices(model.positions, model.joints, model.weights, model.tri_indices, jm)
        m = jm.reshape(-1, 4, 4).transpose(0, 2, 1).astype(np.float64)
        s = np.einsum("vk,vkab->vab", model.weights.astype(np.float64), m[model.joints.astype(int)])
        p = np.concatenate([model.positions, np.ones((len(model.positions), 1), np.float32)], 1).astype(np.float64)
        want = np.einsum("vab,vb->va", s, p)[:, :3][model.tri_indices.astype(int)].reshape(-1, 9)
        assert np.allclose(got, want, rtol=0, atol=2e-3)
    rest = ob.skin_vertices(model.positions, model.joints, model.weights, model.tri_indices, model.joint_matrices(0.0))
    assert np.abs(got - rest).max() > 5.0


@pytest.mark.gpu
def test_gpu_skinning_feeds_binning_and_trace(model, ob, rt):
    """rt_skin_vertices == the oracle's skinning bit for bit; the grid binned from the device-resident result and the
    frame traced through it equal the oracle's on the same vertices; matrices from host and from device memory."""
    import torch
    core = rt.TraceCore(0)
    core.set_scene(None, rt.triangles_from_xyz9(model.bind_pose_triangles()))
    core.set_skin(model.positions, model.joints, model.weights, model.tri_indices)
    w, h = 240, 135
    cam = rt.Camera((-300, 250, 600), 2.7, -0.2).to_uniform(0.0, float(h))
    ocam = ob.camera_to_uniform((-300, 250, 600), 2.7, -0.2, 0.0, float(h))
    for k, t in enumerate((0.0, 0.7, 1.9, 3.1)):
        jm = model.joint_matrices(t)
        core.skin_vertices(torch.from_numpy(jm).cuda() if k % 2 else jm)
        want = ob.skin_vertices(model.positions, model.joints, model.weights, model.tri_indices, jm)
        with pytest.raises(rt.RtError):
            core.trace_primary(cam, 8, 8)  # the old grid is stale
        assert np.array_equal(core.read_vertices().view(np.uint32), want.view(np.uint32)), f"vertices at t={t}"
        core.build_grid()
        scene = ob.Scene(None, ob.triangles_from_xyz9(want))
        cs, idx = core.read_grid_csr()
        assert np.array_equal(cs, scene.grid.cell_start) and np.array_equal(idx, scene.grid.indices)
        p, s = core.trace_primary(cam, w, h)
        op, os_, _ = ob.trace_primary(scene, ocam, w, h)
        for got, exp in ((p, op), (s, os_)):
            assert np.array_equal(got["id"], exp["id"]) and np.array_equal(got["t"].view(np.uint32), exp["t"].view(np.uint32))
        assert (op["id"] != ob.MISS).sum() > 500
    with pytest.raises(rt.RtError):
        core.skin_vertices(jm[:10])  # fewer matrices than the skin's joints
    core.close()
