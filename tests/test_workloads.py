"""The product-side workload generators (ray-tracer_amd/workloads.py) against the oracle's restatements."""
import numpy as np

from ray_tracer_amd import workloads as wl


def test_triangles_bit_equal(ob):
    for n in (1, 100, 10000, 30000):
        a, b = wl.generate_test_triangles(n), ob.generate_test_triangles(n)
        assert a.tobytes() == b.tobytes()


def test_particle_boxes_bit_equal(ob):
    for n in (100, 10000):
        mn, mx = wl.particle_aabbs(n)
        omn, omx = ob.particle_aabbs(n)
        assert np.array_equal(mn, omn) and np.array_equal(mx, omx)


def test_rays_and_unit_vectors_close(ob):
    """cos/sin come from numpy instead of libm: equal to a few ulp, not bit for bit."""
    o, d = wl.generate_test_rays(5000)
    oo, od = ob.generate_test_rays(5000)
    assert np.allclose(o, oo, rtol=0, atol=2e-5) and np.allclose(d, od, rtol=0, atol=1e-6)
    u = wl.random_unit_vectors(100)
    ou = np.stack([ob.random_unit_vector(s) for s in range(100)])
    assert np.allclose(u, ou, rtol=0, atol=1e-6)
    assert np.allclose(np.linalg.norm(u, axis=1), 1.0, atol=1e-6)


def test_rotation_is_rigid():
    rng = np.random.default_rng(0)
    v = rng.uniform(-100, 100, (50, 9)).astype(np.float32)
    w = wl.rotate_about_y(v, 17)
    e = lambda a: np.linalg.norm(a.reshape(-1, 3, 3)[:, 1] - a.reshape(-1, 3, 3)[:, 0], axis=1)
    assert np.allclose(e(v), e(w), rtol=1e-5)
    assert np.allclose(wl.rotate_about_y(v, 0)[:, [0, 2]], v[:, [0, 2]], atol=1e-4)
