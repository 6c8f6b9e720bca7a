"""The random scenes once more through the CHECKED build of the library (make -C ray-tracer_amd/csrc checked): its trace
kernels compare every reference-record, cell and triangle index they form with the array it addresses and count
violations instead of faulting. The library is bound once per process, hence the child process (one, run once)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECKED = os.path.join(ROOT, "ray-tracer_amd", "librt_mi355_checked.so")


def test_random_scenes_through_the_checked_kernels():
    assert os.path.exists(CHECKED), "build it first: make -C ray-tracer_amd/csrc checked (__graft_entry__.build() does)"
    env = dict(os.environ, RT_LIB_PATH=CHECKED, RT_FUZZ_TRIALS="20", RT_FUZZ_SEED="77", RT_EXPECT_CHECKED="1")
    out = subprocess.run([sys.executable, "-u", "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"), "-x", "-q", "-s",
                          "-m", "gpu", "--timeout", "300", "--timeout-method", "thread", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])
    assert "checked build: 0 index violations" in out.stdout, out.stdout[-2000:]
