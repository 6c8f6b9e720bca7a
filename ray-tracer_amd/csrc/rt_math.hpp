// Device-side f32 vector helpers and the two primitive intersection routines.
//
// Numerics contract: every routine evaluates the same IEEE-754 binary32 expression tree
// as the reference's scalar code, in the same operand order - glam 0.29 scalar Vec3
// (dot = (x*x' + y*y') + z*z', cross = (y*z' - y'*z, ...), normalize = v * (1/sqrt(dot))) -
// so results are bit-identical to a CPU evaluation. That needs: no FMA contraction
// (-ffp-contract=off), no fast-math, correctly rounded 1/x and sqrt (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt) and f32 denormals preserved (gfx950 default).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct V3 {
    float x, y, z;
};

__host__ __device__ inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__host__ __device__ inline V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
__host__ __device__ inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__host__ __device__ inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__host__ __device__ inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__host__ __device__ inline V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__host__ __device__ inline float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
__host__ __device__ inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
__host__ __device__ inline float length(V3 a) { return sqrtf(dot(a, a)); }
__host__ __device__ inline V3 normalize(V3 a) { return a * (1.0f / length(a)); }
__host__ __device__ inline V3 vmin(V3 a, V3 b) { return V3{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
__host__ __device__ inline V3 vmax(V3 a, V3 b) { return V3{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
__host__ __device__ inline V3 vabs(V3 a) { return V3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
__host__ __device__ inline float sign0(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
// a*(1-t) + b*t
__host__ __device__ inline V3 mix(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }

// float -> u32, saturating, NaN -> 0 (Rust `as u32`, WGSL u32())
__host__ __device__ inline uint32_t f2u_sat(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_cvt_u32_f32 has exactly these semantics (NaN and negatives -> 0, too large -> 0xFFFFFFFF); a plain cast
    // is undefined for out-of-range input, so the instruction is named explicitly
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
#endif
}
// float -> i32, saturating, NaN -> 0 (Rust `as i32`)
__host__ __device__ inline int32_t f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)x;
}

// 1/d with |d| < 1e-8 replaced: reference src/math/ray.rs:8-12 (copysign form, WGSL = false)
// and raytracer_unified.wgsl:162-166,509-513 (sign form, WGSL = true).
template <bool WGSL>
__host__ __device__ inline float safe_inv(float d) {
    const float eps = 1e-8f;
    if (fabsf(d) < eps) {
        if (WGSL) return 1.0f / (eps * sign0(d));
        return 1.0f / copysignf(eps, d);
    }
    return 1.0f / d;
}

// Slab test with a precomputed safe inverse direction.
// reference: src/math/ray.rs:14-35 == raytracer_unified.wgsl:168-189
__host__ __device__ inline float slab(V3 o, V3 inv, V3 bmin, V3 bmax) {
    V3 t_min = (bmin - o) * inv;
    V3 t_max = (bmax - o) * inv;
    V3 t1 = vmin(t_min, t_max);
    V3 t2 = vmax(t_min, t_max);
    float t_near = fmaxf(fmaxf(t1.x, t1.y), t1.z);
    float t_far = fminf(fminf(t2.x, t2.y), t2.z);
    if (t_near > t_far || t_far < 0.0f) return -1.0f;
    if (t_near < 0.0f) return t_far > 0.001f ? t_far : -1.0f;
    return t_near;
}

// Moller-Trumbore against a triangle stored as (v0, e1 = v1 - v0, e2 = v2 - v0).
// reference: src/core/triangle_intersection.rs:31-82 (WGSL = false: eps 1e-6, reject t < eps)
// and raytracer_unified.wgsl:238-277 (WGSL = true: eps 1e-5, accept t > eps).
// Returns t, or -1 when the triangle is rejected.
// The reference rejects step by step (determinant, u, v, t). Every value here is produced by the same
// operations in the same order, so accepted hits carry identical bits; only the control flow is flatter:
// the reciprocal is taken before the determinant test (its value is unused when that test fails) and there
// is one branch, after the u range test, which most candidates fail.
template <bool WGSL>
__host__ __device__ inline float moller_trumbore(V3 o, V3 d, V3 v0, V3 e1, V3 e2) {
    const float eps = WGSL ? 0.00001f : 1e-6f;
    const V3 h = cross(d, e2);
    const float a = dot(e1, h);
    const float f = 1.0f / a;
    const V3 s = o - v0;
    const float u = f * dot(s, h);
    float t = -1.0f;
    if (!(fabsf(a) < eps) && !(u < 0.0f || u > 1.0f)) {
        const V3 q = cross(s, e1);
        const float v = f * dot(d, q);
        const float tt = f * dot(e2, q);
        const bool t_ok = WGSL ? (tt > eps) : !(tt < eps);
        if (!(v < 0.0f || u + v > 1.0f) && t_ok) t = tt;
    }
    return t;
}

}  // namespace rt
