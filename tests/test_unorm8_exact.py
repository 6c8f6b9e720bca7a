"""The texture filter turns a texel byte into a / 255 (Rgba8Unorm, src/renderer.rs:196-248). The GPU code replaces the
f32 division by q = a * r, q + fma(-q, 255, a) * r with r = RN(1/255); this checks in exact rational arithmetic that the
result is the correctly rounded quotient - what the division (and the oracle) gives - for all 256 bytes."""
import os
import re
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rn(x):
    """Fraction -> nearest float32, ties to even."""
    f = np.float32(float(x))
    near = [np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))]
    return np.float32(min(near, key=lambda c: (abs(Fraction(float(c)) - x), int(np.float32(c).view(np.uint32)) & 1)))


def test_three_instruction_quotient_is_the_division():
    src = open(os.path.join(ROOT, "ray-tracer_amd", "csrc", "rt_texture.hpp")).read()
    bits = int(re.search(r"RCP_255_BITS = (0x[0-9A-Fa-f]+)u", src).group(1), 16)
    r = np.array([bits], np.uint32).view(np.float32)[0]
    assert r == rn(Fraction(1, 255))
    for a in range(256):
        want = np.float32(a) / np.float32(255.0)
        assert want == rn(Fraction(a, 255))
        q = rn(Fraction(a) * Fraction(float(r)))
        e = rn(Fraction(a) - Fraction(float(q)) * 255)            # fma(-q, 255, a): one rounding
        got = rn(Fraction(float(e)) * Fraction(float(r)) + Fraction(float(q)))  # fma(e, r, q): one rounding
        assert got == want, a
