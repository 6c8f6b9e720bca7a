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

// The reference's second triangle routine, the "watertight" test (src/core/triangle_intersection.rs:101-181):
// shear the triangle into the ray's frame along its dominant axis, take the three signed edge functions, accept
// when they do not disagree in sign. Returns t, or -1.
__host__ __device__ inline float watertight(V3 o, V3 d, V3 v0, V3 v1, V3 v2) {
    const float eps = 1e-6f;
    const V3 a = v0 - o, b = v1 - o, c = v2 - o;
    const V3 ad = vabs(d);
    const int kz = (ad.x > ad.y && ad.x > ad.z) ? 0 : (ad.y > ad.z ? 1 : 2);
    const int kx = (kz + 1) % 3, ky = (kx + 1) % 3;
    auto at = [](V3 p, int k) { return k == 0 ? p.x : (k == 1 ? p.y : p.z); };
    const float dx = at(d, kx), dy = at(d, ky), dz = at(d, kz);
    const float sx = dx / dz, sy = dy / dz, sz = 1.0f / dz;
    const float ax = at(a, kx) - sx * at(a, kz), ay = at(a, ky) - sy * at(a, kz);
    const float bx = at(b, kx) - sx * at(b, kz), by = at(b, ky) - sy * at(b, kz);
    const float cx = at(c, kx) - sx * at(c, kz), cy = at(c, ky) - sy * at(c, kz);
    const float u = cx * by - cy * bx;
    const float v = ax * cy - ay * cx;
    const float w = bx * ay - by * ax;
    if ((u < 0.0f || v < 0.0f || w < 0.0f) && (u > 0.0f || v > 0.0f || w > 0.0f)) return -1.0f;
    const float det = u + v + w;
    if (fabsf(det) < eps) return -1.0f;
    const float az = sz * at(a, kz), bz = sz * at(b, kz), cz = sz * at(c, kz);
    const float t = (u * az + v * bz + w * cz) * (1.0f / det);
    if (t < eps) return -1.0f;
    return t;
}

#if defined(__HIPCC__)
// The same test written for the packed two-lane f32 instructions of gfx950 (v_pk_mul_f32, v_pk_add_f32), which run
// at the rate of the one-lane forms. Every product and sum below is one of the reference's, with the same operands
// and the same single rounding; only (a) operands of commutative products are swapped, and (b) the middle
// component of each cross product is carried negated (hy_n = -h.y, qy_n = -q.y) and subtracted where the reference
// adds it - x - y and x + (-y), and a - b and -(b - a), are the same IEEE operation - so the results carry identical
// bits. Pairs are (x, y) of each record quad, which is where a 16-byte load puts them.
typedef float f2 __attribute__((ext_vector_type(2)));

struct RayPk {
    f2 dxy, oxy;
    float dz, oz;
};
__device__ inline RayPk make_ray_pk(V3 o, V3 d) { return RayPk{f2{d.x, d.y}, f2{o.x, o.y}, d.z, o.z}; }

template <bool WGSL>
__device__ inline float moller_trumbore_pk(const RayPk& r, float4 c0, float4 c1, float4 c2) {
    const float eps = WGSL ? 0.00001f : 1e-6f;
    const f2 v0xy = {c0.x, c0.y}, e1xy = {c1.x, c1.y}, e2xy = {c2.x, c2.y};
    const f2 dyx = __builtin_shufflevector(r.dxy, r.dxy, 1, 0);
    // h = d x e2: (h.x, -h.y) = e2.z * (d.y, d.x) - d.z * (e2.y, e2.x);  h.z = d.x*e2.y - e2.x*d.y
    const f2 hA = dyx * c2.z;
    const f2 hB = __builtin_shufflevector(e2xy, e2xy, 1, 0) * r.dz;
    const f2 hxny = hA - hB;
    const f2 hC = e2xy * dyx;
    const float hz = hC.y - hC.x;
    // a = (e1.x*h.x + e1.y*h.y) + e1.z*h.z
    const f2 aD = e1xy * hxny;
    const float a = (aD.x - aD.y) + c1.z * hz;
    const float f = 1.0f / a;
    // s = o - v0;  u = f * ((s.x*h.x + s.y*h.y) + s.z*h.z)
    const f2 sxy = r.oxy - v0xy;
    const float sz = r.oz - c0.z;
    const f2 uE = sxy * hxny;
    const float u = f * ((uE.x - uE.y) + sz * hz);
    float t = -1.0f;
    if (!(fabsf(a) < eps) && !(u < 0.0f || u > 1.0f)) {
        // q = s x e1: (q.x, -q.y) = e1.z * (s.y, s.x) - s.z * (e1.y, e1.x);  q.z = s.x*e1.y - e1.x*s.y
        const f2 syx = __builtin_shufflevector(sxy, sxy, 1, 0);
        const f2 qA = syx * c1.z;
        const f2 qB = __builtin_shufflevector(e1xy, e1xy, 1, 0) * sz;
        const f2 qxny = qA - qB;
        const f2 qC = e1xy * syx;
        const float qz = qC.y - qC.x;
        const f2 vG = r.dxy * qxny;
        const float v = f * ((vG.x - vG.y) + r.dz * qz);
        const f2 tH = e2xy * qxny;
        const float tt = f * ((tH.x - tH.y) + c2.z * qz);
        const bool t_ok = WGSL ? (tt > eps) : !(tt < eps);
        if (!(v < 0.0f || u + v > 1.0f) && t_ok) t = tt;
    }
    return t;
}
#endif

}  // namespace rt
