"""Synthetic workloads of the reference's own benchmarks, as numpy arrays ready for TraceCore.

These are the generators the reference's example and criterion benches feed their CPU loops with; the
measurement tools use them so the GPU path is timed on the very scenes the reference quotes numbers for.
All arithmetic is float32 in the reference's order."""
import numpy as np

from .host import TRIANGLE_DTYPE

f32 = np.float32


def generate_test_triangles(count):
    """src/core/benchmark.rs:99-123: unit right triangles on a pseudo-random lattice in [-50, 51)^3.
    Positions repeat with period 10 000, so larger scenes stack exact duplicates."""
    i = np.arange(count, dtype=np.uint64)
    x = ((i * 7919) % 10000).astype(f32) / f32(100.0) - f32(50.0)
    y = ((i * 6547) % 10000).astype(f32) / f32(100.0) - f32(50.0)
    z = ((i * 4231) % 10000).astype(f32) / f32(100.0) - f32(50.0)
    t = np.zeros(count, TRIANGLE_DTYPE)
    t["v0"] = np.stack([x, y, z], 1)
    t["v1"] = np.stack([x + f32(1.0), y, z], 1)
    t["v2"] = np.stack([x, y + f32(1.0), z], 1)
    t["uv1"] = (1.0, 0.0)
    t["uv2"] = (0.0, 1.0)
    return t


def generate_test_rays(count):
    """src/core/benchmark.rs:126-146: origins on a radius-100 circle, all aimed at the scene centre.
    (cos/sin come from numpy here, so the last bit may differ from the reference's libm.)"""
    i = np.arange(count, dtype=np.uint64)
    angle = (i.astype(f32) / f32(count)) * f32(2.0) * f32(np.pi)
    height = ((i * 997) % count).astype(f32) / f32(count) * f32(100.0) - f32(50.0)
    o = np.stack([f32(100.0) * np.cos(angle, dtype=f32), height, f32(100.0) * np.sin(angle, dtype=f32)], 1).astype(f32)
    d = -o
    n = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=f32)
    return o, (d * (f32(1.0) / n)[:, None]).astype(f32)


def random_unit_vectors(count):
    """benches/bounding_volumes.rs:39-47 for seeds 0..count-1."""
    s = np.arange(count, dtype=f32)
    theta = np.fmod(s * f32(0.123456), f32(2.0) * f32(np.pi))
    phi = np.fmod(s * f32(0.789012), f32(np.pi))
    return np.stack([np.sin(phi, dtype=f32) * np.cos(theta, dtype=f32), np.sin(phi, dtype=f32) * np.sin(theta, dtype=f32),
                     np.cos(phi, dtype=f32)], 1).astype(f32)


def particle_aabbs(count, half=0.5):
    """benches/bounding_volumes.rs:117-131: the particle-system boxes of the criterion groups."""
    i = np.arange(count, dtype=f32)
    c = np.stack([np.fmod(i * f32(0.1), f32(20.0)) - f32(10.0), np.fmod(i * f32(0.2), f32(20.0)) - f32(10.0),
                  -np.fmod(i * f32(0.3), f32(50.0)) - f32(10.0)], 1).astype(f32)
    return (c - f32(half)).astype(f32), (c + f32(half)).astype(f32)


def rotate_about_y(xyz9, frame, frames_per_turn=120):
    """Deformation used for the animated-model measurement: rigid rotation about +Y through the model's centroid
    by 2*pi*frame/frames_per_turn plus a vertical offset 10*(sin(2t)+1)/2 with t = frame/60 (the moving-box
    interpolation of raytracer_unified.wgsl:201). The reference has no skinning; this is a synthetic motion."""
    v = np.asarray(xyz9, f32).reshape(-1, 3)
    c = v.mean(axis=0, dtype=np.float64).astype(f32)
    a = 2.0 * np.pi * frame / frames_per_turn
    ca, sa = f32(np.cos(a)), f32(np.sin(a))
    p = v - c
    out = np.empty_like(v)
    out[:, 0] = ca * p[:, 0] + sa * p[:, 2] + c[0]
    out[:, 1] = p[:, 1] + c[1] + f32(10.0 * (np.sin(2.0 * frame / 60.0) + 1.0) / 2.0)
    out[:, 2] = -sa * p[:, 0] + ca * p[:, 2] + c[2]
    return out.reshape(-1, 9)


def primary_rays(camera_uniform, width, height, fov=0.785398):
    """The shader's ray generation (raytracer_unified.wgsl:625-653) in numpy float32, for feeding ray-list entry
    points with a frame's primary rays. Returns (origins, dirs), each (width*height, 3)."""
    px, py = np.meshgrid(np.arange(width, dtype=f32), np.arange(height, dtype=f32))
    u = (px.reshape(-1) + f32(0.5)) / f32(width)
    v = (py.reshape(-1) + f32(0.5)) / f32(height)
    ndc_x, ndc_y = u * f32(2.0) - f32(1.0), v * f32(2.0) - f32(1.0)
    aspect, scale = f32(width) / f32(height), f32(np.tan(f32(fov)))
    fwd, right, up = (np.array(list(getattr(camera_uniform, k)), f32) for k in ("forward", "right", "up"))
    d = fwd[None, :] + right[None, :] * (ndc_x * aspect * scale)[:, None] + up[None, :] * (-ndc_y * scale)[:, None]
    d = (d / np.sqrt((d * d).sum(axis=1, dtype=f32))[:, None]).astype(f32)
    o = np.tile(np.array(list(camera_uniform.position), f32), (width * height, 1))
    return o, d
