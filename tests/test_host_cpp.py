"""include/rt_host.hpp (the compiled host-side mirror of RayTracer::new / ::render) through its demo program.
The binary is built by __graft_entry__.build() (tests/host_cpp/Makefile), never inside a test."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host_cpp", "host_demo")


def test_host_demo_reports_missing_gpu_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr  # an error, not a silent CPU fallback


@pytest.mark.gpu
def test_host_demo_matches_golden(tmp_path, golden, rt, ob):
    out = tmp_path / "frame.bin"
    r = subprocess.run([EXE, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    n = 64 * 36
    rec = raw[:2 * n * 8].view(rt.HIT_DTYPE)
    p, s = rec[:n], rec[n:]
    rgba = raw[2 * n * 8:].reshape(-1, 4)
    want, _ = ob.render(ob.Scene(None, ob.pyramid_triangles()), ob.camera_to_uniform((0, 8, 20), np.pi, -0.5, 0.0, 36.0), 64, 36)
    assert np.array_equal(rgba, want)
    for got, want in ((p, golden["pyramid_primary"]), (s, golden["pyramid_shadow"])):
        assert np.array_equal(got["id"], want["id"]) and np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
    assert r.stdout.strip() == "hits 26"
