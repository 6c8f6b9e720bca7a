"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`): a driver walks every
routine family on small inputs that reach the index-forming corners; a finding is a non-zero exit. Host code only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "oracle sanitizer run: clean" in out.stdout
