// CPU restatement of the reference's trace path. TEST INFRASTRUCTURE ONLY - see oracle.h.
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
// Every f32 expression below keeps the operand order of the cited reference line; nothing
// is re-associated, fused or vectorised by hand.
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace {

// f32::min / f32::max of the reference (and fminf / fmaxf of C): the other operand when one is NaN, the second
// operand on a tie. Written out so the compiler inlines them (a libm call per min/max made this port several times
// slower than the compiled Rust it stands in for when it is timed as the CPU baseline).
static inline float fmin_(float a, float b) { return (b != b) ? a : ((a < b) ? a : b); }
static inline float fmax_(float a, float b) { return (b != b) ? a : ((a > b) ? a : b); }

struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 ld(const float* p) { return V3{p[0], p[1], p[2]}; }
inline void st(float* p, V3 v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
// glam scalar Vec3::dot: (x*x') + (y*y') + (z*z'), left to right
inline float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
// glam scalar Vec3::cross
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline float length(V3 a) { return sqrtf(dot(a, a)); }
// glam Vec3::normalize = self * self.length_recip(), length_recip = 1.0 / length
inline V3 normalize(V3 a) { return a * (1.0f / length(a)); }
inline V3 vmin(V3 a, V3 b) { return V3{fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return V3{fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
inline V3 vabs(V3 a) { return V3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
inline float wgsl_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// Rust `as u32` / WGSL u32(): saturating, NaN -> 0
inline uint32_t f2u_sat(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
// Rust `as i32`: saturating, NaN -> 0
inline int32_t f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
// Rust `as usize`: saturating, NaN -> 0
inline uint64_t f2usize_sat(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)x;
}

constexpr float AABB_EPS = 1e-8f;

// src/math/ray.rs:8-12 (copysign form) / raytracer_unified.wgsl:162-166,509-513 (sign form)
inline float safe_inv(float d, int wgsl) {
    if (fabsf(d) < AABB_EPS) {
        if (wgsl) return 1.0f / (AABB_EPS * wgsl_sign(d));
        return 1.0f / copysignf(AABB_EPS, d);
    }
    return 1.0f / d;
}

// src/math/ray.rs:3-36 == raytracer_unified.wgsl:158-190
inline float intersect_aabb(V3 o, V3 d, V3 bmin, V3 bmax, int wgsl) {
    V3 inv = v3(safe_inv(d.x, wgsl), safe_inv(d.y, wgsl), safe_inv(d.z, wgsl));
    V3 t_min = (bmin - o) * inv;
    V3 t_max = (bmax - o) * inv;
    V3 t1 = vmin(t_min, t_max);
    V3 t2 = vmax(t_min, t_max);
    float t_near = fmax_(fmax_(t1.x, t1.y), t1.z);
    float t_far = fmin_(fmin_(t2.x, t2.y), t2.z);
    if (t_near > t_far || t_far < 0.0f) return -1.0f;
    if (t_near < 0.0f) {
        if (t_far > 0.001f) return t_far;
        return -1.0f;
    }
    return t_near;
}

struct TriHit {
    float t, u, v;
    V3 n;
};

// src/core/triangle_intersection.rs:31-82 (wgsl=0) / raytracer_unified.wgsl:238-277 (wgsl=1)
inline bool moller_trumbore(V3 o, V3 d, V3 v0, V3 v1, V3 v2, int wgsl, TriHit* out) {
    const float eps = wgsl ? 0.00001f : 1e-6f;
    V3 edge1 = v1 - v0;
    V3 edge2 = v2 - v0;
    V3 h = cross(d, edge2);
    float a = dot(edge1, h);
    if (fabsf(a) < eps) return false;
    float f = 1.0f / a;
    V3 s = o - v0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = cross(s, edge1);
    float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = f * dot(edge2, q);
    if (wgsl) {
        if (!(t > eps)) return false;
    } else {
        if (t < eps) return false;
    }
    out->t = t;
    out->u = u;
    out->v = v;
    out->n = normalize(cross(edge1, edge2));
    return true;
}

// src/core/triangle_intersection.rs:101-181
inline bool watertight(V3 o, V3 d, V3 v0, V3 v1, V3 v2, TriHit* out) {
    const float eps = 1e-6f;
    V3 a = v0 - o, b = v1 - o, c = v2 - o;
    V3 ad = vabs(d);
    int kz = (ad.x > ad.y && ad.x > ad.z) ? 0 : (ad.y > ad.z ? 1 : 2);
    int kx = (kz + 1) % 3, ky = (kx + 1) % 3;
    auto at = [](V3 p, int k) { return k == 0 ? p.x : (k == 1 ? p.y : p.z); };
    V3 dd = v3(at(d, kx), at(d, ky), at(d, kz));
    float sx = dd.x / dd.z, sy = dd.y / dd.z, sz = 1.0f / dd.z;
    float ax = at(a, kx) - sx * at(a, kz), ay = at(a, ky) - sy * at(a, kz);
    float bx = at(b, kx) - sx * at(b, kz), by = at(b, ky) - sy * at(b, kz);
    float cx = at(c, kx) - sx * at(c, kz), cy = at(c, ky) - sy * at(c, kz);
    float u = cx * by - cy * bx;
    float v = ax * cy - ay * cx;
    float w = bx * ay - by * ax;
    if ((u < 0.0f || v < 0.0f || w < 0.0f) && (u > 0.0f || v > 0.0f || w > 0.0f)) return false;
    float det = u + v + w;
    if (fabsf(det) < eps) return false;
    float az = sz * at(a, kz), bz = sz * at(b, kz), cz = sz * at(c, kz);
    float t = u * az + v * bz + w * cz;
    float inv_det = 1.0f / det;
    t = t * inv_det;
    u = u * inv_det;
    v = v * inv_det;
    if (t < eps) return false;
    out->t = t;
    out->u = u;
    out->v = v;
    out->n = normalize(cross(v1 - v0, v2 - v0));
    return true;
}

// src/types.rs:235-244
inline void tri_bounds(const TriangleData& t, V3* mn, V3* mx) {
    V3 a = ld(t.v0), b = ld(t.v1), c = ld(t.v2);
    *mn = vmin(vmin(a, b), c);
    *mx = vmax(vmax(a, b), c);
}

}  // namespace

// =============================== primitive routine exports ===============================

extern "C" float orc_intersect_aabb(const float o[3], const float d[3], const float bmin[3], const float bmax[3],
                                    int wgsl_compat) {
    return intersect_aabb(ld(o), ld(d), ld(bmin), ld(bmax), wgsl_compat);
}

static void put6(const TriHit& h, float out6[6]) {
    out6[0] = h.t;
    out6[1] = h.u;
    out6[2] = h.v;
    out6[3] = h.n.x;
    out6[4] = h.n.y;
    out6[5] = h.n.z;
}

extern "C" int orc_moller_trumbore(const float o[3], const float d[3], const float v0[3], const float v1[3],
                                   const float v2[3], int wgsl_compat, float out6[6]) {
    TriHit h;
    if (!moller_trumbore(ld(o), ld(d), ld(v0), ld(v1), ld(v2), wgsl_compat, &h)) return 0;
    put6(h, out6);
    return 1;
}

extern "C" int orc_watertight(const float o[3], const float d[3], const float v0[3], const float v1[3],
                              const float v2[3], float out6[6]) {
    TriHit h;
    if (!watertight(ld(o), ld(d), ld(v0), ld(v1), ld(v2), &h)) return 0;
    put6(h, out6);
    return 1;
}

// src/core/triangle_intersection.rs:184-204: list order, strict <
extern "C" int64_t orc_batch_intersect_triangles(const float o[3], const float d[3], const TriangleData* tris,
                                                 const uint32_t* indices, size_t n_indices, float out6[6]) {
    int64_t best = -1;
    float closest_t = INFINITY;
    TriHit bh{};
    for (size_t i = 0; i < n_indices; ++i) {
        const TriangleData& tr = tris[indices[i]];
        TriHit h;
        if (moller_trumbore(ld(o), ld(d), ld(tr.v0), ld(tr.v1), ld(tr.v2), 0, &h)) {
            if (h.t < closest_t) {
                closest_t = h.t;
                best = (int64_t)indices[i];
                bh = h;
            }
        }
    }
    if (best >= 0 && out6) put6(bh, out6);
    return best;
}

// src/math/grid.rs:3-15 (multiplication by the inverse cell size)
extern "C" void orc_world_to_cell(const float pos[3], const float bounds_min[3], float cell_size, int32_t out[3]) {
    V3 rel = ld(pos) - ld(bounds_min);
    float inv = 1.0f / cell_size;
    out[0] = f2i_sat(floorf(rel.x * inv));
    out[1] = f2i_sat(floorf(rel.y * inv));
    out[2] = f2i_sat(floorf(rel.z * inv));
}

// src/math/aabb.rs:14-29
extern "C" void orc_aabb_union(const float amin[3], const float amax[3], const float bmin[3], const float bmax[3],
                               float omin[3], float omax[3]) {
    st(omin, vmin(ld(amin), ld(bmin)));
    st(omax, vmax(ld(amax), ld(bmax)));
}
extern "C" void orc_aabb_center(const float amin[3], const float amax[3], float out[3]) {
    st(out, (ld(amin) + ld(amax)) * 0.5f);
}
extern "C" float orc_aabb_surface_area(const float amin[3], const float amax[3]) {
    V3 d = ld(amax) - ld(amin);
    return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}
extern "C" void orc_triangle_bounds(const TriangleData* t, float omin[3], float omax[3]) {
    V3 mn, mx;
    tri_bounds(*t, &mn, &mx);
    st(omin, mn);
    st(omax, mx);
}

extern "C" void orc_intersect_aabb_pairs(const float* origins, const float* dirs, uint32_t n_rays,
                                         const float* box_min, const float* box_max, uint32_t n_prims,
                                         int wgsl_compat, float* out_t) {
    for (uint32_t r = 0; r < n_rays; ++r)
        for (uint32_t p = 0; p < n_prims; ++p)
            out_t[(size_t)r * n_prims + p] = intersect_aabb(ld(origins + 3 * r), ld(dirs + 3 * r), ld(box_min + 3 * p),
                                                            ld(box_max + 3 * p), wgsl_compat);
}

extern "C" void orc_watertight_pairs(const float* origins, const float* dirs, uint32_t n_rays, const float* v012,
                                     uint32_t n_prims, float* out_t) {
    for (uint32_t r = 0; r < n_rays; ++r)
        for (uint32_t p = 0; p < n_prims; ++p) {
            const float* v = v012 + 9 * (size_t)p;
            TriHit h;
            out_t[(size_t)r * n_prims + p] = watertight(ld(origins + 3 * r), ld(dirs + 3 * r), ld(v), ld(v + 3), ld(v + 6), &h) ? h.t : -1.0f;
        }
}

extern "C" void orc_intersect_tri_pairs(const float* origins, const float* dirs, uint32_t n_rays, const float* v012,
                                        uint32_t n_prims, int wgsl_compat, float* out_t) {
    for (uint32_t r = 0; r < n_rays; ++r)
        for (uint32_t p = 0; p < n_prims; ++p) {
            TriHit h;
            const float* v = v012 + 9 * (size_t)p;
            bool hit = moller_trumbore(ld(origins + 3 * r), ld(dirs + 3 * r), ld(v), ld(v + 3), ld(v + 6),
                                       wgsl_compat, &h);
            out_t[(size_t)r * n_prims + p] = hit ? h.t : -1.0f;
        }
}

// ======================================== camera ========================================

// src/camera.rs:51-54,76-122
extern "C" void orc_camera_to_uniform(const float position[3], float yaw, float pitch, float time,
                                      float screen_height, float fov, int show_grid, CameraUniform* out) {
    V3 fwd = normalize(v3(sinf(yaw) * cosf(pitch), sinf(pitch), cosf(yaw) * cosf(pitch)));
    V3 right = normalize(cross(fwd, v3(0.0f, 1.0f, 0.0f)));
    memset(out, 0, sizeof(*out));
    st(out->position, ld(position));
    st(out->forward, fwd);
    st(out->right, right);
    st(out->up, v3(0.0f, 1.0f, 0.0f));
    out->time = time;
    out->lod_factor = screen_height / (2.0f * tanf(fov / 2.0f));
    out->min_pixel_size = 2.0f;
    out->show_grid = show_grid ? 1.0f : 0.0f;
}

// ========================================= grids =========================================

struct orc_grid {
    GridMetadata meta;
    std::vector<uint32_t> cell_start;  // n_cells + 1
    std::vector<uint32_t> indices;
    std::vector<uint8_t> coarse;
    uint64_t n_cells = 0;
    uint64_t dropped = 0;
    uint32_t max_per_cell = 0;
};

namespace {

struct Dim3 {
    uint64_t x, y, z;
};

// src/grid.rs:9-16
Dim3 hier_dims(V3 bmin, V3 bmax, float cell) {
    V3 e = bmax - bmin;
    return Dim3{f2usize_sat(ceilf(e.x / cell)), f2usize_sat(ceilf(e.y / cell)), f2usize_sat(ceilf(e.z / cell))};
}
// src/grid_triangles.rs:126-139
Dim3 tri_dims(V3 bmin, V3 bmax, float cell) {
    float safe = fmax_(cell, RT_TRI_MIN_CELL_SIZE);
    V3 e = bmax - bmin;
    auto cl = [](uint64_t v) { return std::min<uint64_t>(std::max<uint64_t>(v, 1), RT_TRI_MAX_GRID_DIM); };
    return Dim3{cl(f2usize_sat(ceilf(e.x / safe))), cl(f2usize_sat(ceilf(e.y / safe))),
                cl(f2usize_sat(ceilf(e.z / safe)))};
}

struct Range3 {
    int64_t lo[3], hi[3];  // inclusive; empty when lo > hi
};

// src/grid.rs:218-237,292-299: division, floor, clamp at 0, `as u32`; cells outside the grid filtered
Range3 hier_range(V3 omin, V3 omax, V3 bmin, float cell, Dim3 d) {
    V3 rmin = omin - bmin, rmax = omax - bmin;
    uint32_t lo[3] = {f2u_sat(fmax_(floorf(rmin.x / cell), 0.0f)), f2u_sat(fmax_(floorf(rmin.y / cell), 0.0f)),
                      f2u_sat(fmax_(floorf(rmin.z / cell), 0.0f))};
    uint32_t hi[3] = {f2u_sat(fmax_(floorf(rmax.x / cell), 0.0f)), f2u_sat(fmax_(floorf(rmax.y / cell), 0.0f)),
                      f2u_sat(fmax_(floorf(rmax.z / cell), 0.0f))};
    uint64_t dims[3] = {d.x, d.y, d.z};
    Range3 r;
    for (int k = 0; k < 3; ++k) {
        r.lo[k] = lo[k];
        r.hi[k] = std::min<int64_t>((int64_t)hi[k], (int64_t)dims[k] - 1);
    }
    return r;
}
// src/grid_triangles.rs:171-209: division, floor, `as i32`, loops clamped to the grid
Range3 tri_range(V3 omin, V3 omax, V3 bmin, float cell, Dim3 d) {
    V3 rmin = omin - bmin, rmax = omax - bmin;
    int32_t lo[3] = {f2i_sat(floorf(rmin.x / cell)), f2i_sat(floorf(rmin.y / cell)), f2i_sat(floorf(rmin.z / cell))};
    int32_t hi[3] = {f2i_sat(floorf(rmax.x / cell)), f2i_sat(floorf(rmax.y / cell)), f2i_sat(floorf(rmax.z / cell))};
    uint64_t dims[3] = {d.x, d.y, d.z};
    Range3 r;
    for (int k = 0; k < 3; ++k) {
        r.lo[k] = std::max<int64_t>(lo[k], 0);
        r.hi[k] = std::min<int64_t>(hi[k], (int64_t)dims[k] - 1);
    }
    return r;
}

template <class F>
void for_cells(const Range3& r, Dim3 d, F&& f) {
    for (int64_t x = r.lo[0]; x <= r.hi[0]; ++x)
        for (int64_t y = r.lo[1]; y <= r.hi[1]; ++y)
            for (int64_t z = r.lo[2]; z <= r.hi[2]; ++z) f((uint64_t)x + (uint64_t)y * d.x + (uint64_t)z * d.x * d.y);
}

}  // namespace

extern "C" orc_grid* orc_grid_build(const BoxData* boxes, uint32_t nb, const TriangleData* tris, uint32_t nt,
                                    float cell_size, uint32_t max_per_cell, int mode) {
    auto g = std::make_unique<orc_grid>();
    const bool tri_mode = (mode == RT_GRID_TRIANGLES);
    if (tri_mode) {
        boxes = nullptr;
        nb = 0;
    }
    g->max_per_cell = max_per_cell;

    // bounds: src/grid.rs:112-134 / src/grid_triangles.rs:30-55
    V3 bmin = v3(-1.0f, -1.0f, -1.0f), bmax = v3(1.0f, 1.0f, 1.0f);
    bool tri_empty_special = tri_mode && nt == 0;
    if (nb > 0) {
        bmin = ld(boxes[0].min);
        bmax = ld(boxes[0].max);
        for (uint32_t i = 1; i < nb; ++i) {
            bmin = vmin(bmin, ld(boxes[i].min));
            bmax = vmax(bmax, ld(boxes[i].max));
        }
    } else if (nt > 0) {
        tri_bounds(tris[0], &bmin, &bmax);
    }
    for (uint32_t i = 0; i < nt; ++i) {
        V3 mn, mx;
        tri_bounds(tris[i], &mn, &mx);
        bmin = vmin(bmin, mn);
        bmax = vmax(bmax, mx);
    }
    if (!tri_empty_special) {
        bmin = bmin - v3(1.0f, 1.0f, 1.0f);
        bmax = bmax + v3(1.0f, 1.0f, 1.0f);
    }

    memset(&g->meta, 0, sizeof(g->meta));
    st(g->meta.bounds_min, bmin);
    st(g->meta.bounds_max, bmax);
    g->meta.num_levels = RT_GRID_LEVELS;
    g->meta.finest_cell_size = cell_size;

    Dim3 fd = tri_empty_special ? Dim3{1, 1, 1} : (tri_mode ? tri_dims(bmin, bmax, cell_size) : hier_dims(bmin, bmax, cell_size));
    g->meta.grid_sizes[RT_GRID_LEVELS - 1][0] = (uint32_t)fd.x;
    g->meta.grid_sizes[RT_GRID_LEVELS - 1][1] = (uint32_t)fd.y;
    g->meta.grid_sizes[RT_GRID_LEVELS - 1][2] = (uint32_t)fd.z;
    g->n_cells = fd.x * fd.y * fd.z;
    if (g->n_cells > (1ull << 31)) return nullptr;

    // coarse levels: src/grid.rs:138-150,244-253 / src/grid_triangles.rs:60-84 (never read by the shader)
    if (!tri_empty_special) {
        for (uint32_t level = 0; level + 1 < RT_GRID_LEVELS; ++level) {
            float cs = cell_size * (float)(1u << (RT_GRID_LEVELS - 1 - level));
            Dim3 cd = tri_mode ? tri_dims(bmin, bmax, cs) : hier_dims(bmin, bmax, cs);
            g->meta.grid_sizes[level][0] = (uint32_t)cd.x;
            g->meta.grid_sizes[level][1] = (uint32_t)cd.y;
            g->meta.grid_sizes[level][2] = (uint32_t)cd.z;
            size_t base = g->coarse.size();
            g->coarse.resize(base + cd.x * cd.y * cd.z, 0);
            uint8_t* cnt = g->coarse.data() + base;
            auto bump = [&](uint64_t idx) {
                if (cnt[idx] < 255) cnt[idx]++;
            };
            for (uint32_t i = 0; i < nb; ++i)
                for_cells(hier_range(ld(boxes[i].min), ld(boxes[i].max), bmin, cs, cd), cd, bump);
            for (uint32_t i = 0; i < nt; ++i) {
                V3 mn, mx;
                tri_bounds(tris[i], &mn, &mx);
                for_cells(tri_mode ? tri_range(mn, mx, bmin, cs, cd) : hier_range(mn, mx, bmin, cs, cd), cd, bump);
            }
        }
    }

    // fine level: ids in insertion (= ascending id) order, inserts beyond the cap dropped
    // (src/grid.rs:92-100,168-176; src/grid_triangles.rs:92-102,194-196)
    std::vector<uint32_t> count(g->n_cells, 0);
    auto range_of = [&](uint32_t id) {
        if (id < nb) return hier_range(ld(boxes[id].min), ld(boxes[id].max), bmin, cell_size, fd);
        V3 mn, mx;
        tri_bounds(tris[id - nb], &mn, &mx);
        return tri_mode ? tri_range(mn, mx, bmin, cell_size, fd) : hier_range(mn, mx, bmin, cell_size, fd);
    };
    const uint32_t total = nb + nt;
    for (uint32_t id = 0; id < total; ++id)
        for_cells(range_of(id), fd, [&](uint64_t c) {
            if (count[c] < max_per_cell)
                count[c]++;
            else
                g->dropped++;
        });
    g->cell_start.resize(g->n_cells + 1);
    uint64_t acc = 0;
    for (uint64_t c = 0; c < g->n_cells; ++c) {
        g->cell_start[c] = (uint32_t)acc;
        acc += count[c];
    }
    if (acc > 0xFFFFFFF0ull) return nullptr;
    g->cell_start[g->n_cells] = (uint32_t)acc;
    g->indices.resize(acc);
    std::fill(count.begin(), count.end(), 0);
    for (uint32_t id = 0; id < total; ++id)
        for_cells(range_of(id), fd, [&](uint64_t c) {
            if (count[c] < max_per_cell) g->indices[g->cell_start[c] + count[c]++] = id;
        });
    return g.release();
}

extern "C" void orc_grid_free(orc_grid* g) { delete g; }

extern "C" void orc_grid_info(const orc_grid* g, GridMetadata* meta, uint64_t* n_cells, uint64_t* n_refs,
                              uint64_t* dropped_refs) {
    if (meta) *meta = g->meta;
    if (n_cells) *n_cells = g->n_cells;
    if (n_refs) *n_refs = g->indices.size();
    if (dropped_refs) *dropped_refs = g->dropped;
}

extern "C" void orc_grid_copy_csr(const orc_grid* g, uint32_t* cell_start, uint32_t* indices) {
    memcpy(cell_start, g->cell_start.data(), g->cell_start.size() * sizeof(uint32_t));
    if (!g->indices.empty()) memcpy(indices, g->indices.data(), g->indices.size() * sizeof(uint32_t));
}

// src/grid.rs:338-355
extern "C" void orc_grid_copy_dense(const orc_grid* g, FineCellData* out) {
    for (uint64_t c = 0; c < g->n_cells; ++c) {
        memset(&out[c], 0, sizeof(FineCellData));
        uint32_t n = g->cell_start[c + 1] - g->cell_start[c];
        uint32_t m = std::min<uint32_t>(n, RT_MAX_OBJECTS_PER_CELL);
        memcpy(out[c].object_indices, g->indices.data() + g->cell_start[c], m * sizeof(uint32_t));
        out[c].count = n;
    }
}
// src/grid_triangles.rs:246-258
extern "C" void orc_grid_copy_dense_tri(const orc_grid* g, TriFineCellData* out) {
    for (uint64_t c = 0; c < g->n_cells; ++c) {
        memset(&out[c], 0, sizeof(TriFineCellData));
        uint32_t n = g->cell_start[c + 1] - g->cell_start[c];
        uint32_t m = std::min<uint32_t>(n, RT_TRI_MAX_TRIANGLES_PER_CELL);
        memcpy(out[c].object_indices, g->indices.data() + g->cell_start[c], m * sizeof(uint32_t));
        out[c].count = m;
    }
}
extern "C" uint64_t orc_grid_coarse_counts(const orc_grid* g, uint8_t* out, uint64_t cap) {
    if (out) memcpy(out, g->coarse.data(), std::min<uint64_t>(cap, g->coarse.size()));
    return g->coarse.size();
}

// ======================================== tracing ========================================

namespace {

struct Hit {
    bool hit = false;
    float t = 1e10f;
    uint32_t id = RT_MISS;
    V3 pos{0, 0, 0};
    V3 n{0, 0, 0};        // geometric normal
    float u = 0, v = 0;   // barycentrics of a triangle hit
};

struct FrameConsts {
    float t_lerp;      // (sin(time*2)+1)*0.5            raytracer_unified.wgsl:201,449
    float fov_scale;   // tan(DEFAULT_FOV)               raytracer_unified.wgsl:643
    V3 shadow_dir;     // -normalize(LIGHT_DIRECTION)    raytracer_unified.wgsl:599-600
};

FrameConsts frame_consts(const rt_params& p, const CameraUniform* cam) {
    FrameConsts fc;
    float time = cam ? cam->time : 0.0f;
    fc.t_lerp = (sinf(time * 2.0f) + 1.0f) * 0.5f;
    fc.fov_scale = tanf(p.fov);
    fc.shadow_dir = neg(normalize(ld(p.light_direction)));
    return fc;
}

struct Counters {
    uint64_t cells = 0, tri = 0, box = 0;
    uint32_t steps = 0;  // num_steps of the last traced ray (raytracer_unified.wgsl:432,448,521)
};

// raytracer_unified.wgsl:147-155
inline bool should_cull_lod(const CameraUniform& cam, V3 center, V3 size) {
    float distance = length(ld(cam.position) - center);
    if (distance > 200.0f) return true;
    float max_size = fmax_(fmax_(size.x, size.y), size.z);
    float apparent = (max_size / distance) * cam.lod_factor;
    return apparent < cam.min_pixel_size;
}

// WGSL mix(a,b,t) = a*(1-t) + b*t
inline V3 mix(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }

// raytracer_unified.wgsl:193-235
inline Hit intersect_box(V3 o, V3 d, const BoxData& b, float t_lerp, uint32_t idx, int wgsl) {
    Hit h;
    h.t = 1e10f;
    h.id = idx;
    V3 center = mix(ld(b.center0), ld(b.center1), t_lerp);
    V3 half = ld(b.half_size);
    V3 bmin = center - half, bmax = center + half;
    float t = intersect_aabb(o, d, bmin, bmax, wgsl);
    if (t < 0.0f) return h;
    h.hit = true;
    h.t = t;
    h.pos = o + d * t;
    V3 p = h.pos - center;
    V3 dd = vabs(p) - half;
    if (dd.x > dd.y && dd.x > dd.z)
        h.n = v3(wgsl_sign(p.x), 0.0f, 0.0f);
    else if (dd.y > dd.z)
        h.n = v3(0.0f, wgsl_sign(p.y), 0.0f);
    else
        h.n = v3(0.0f, 0.0f, wgsl_sign(p.z));
    return h;
}

// raytracer_unified.wgsl:381-390
inline void world_to_cell_u(V3 pos, V3 bmin, float cell, uint32_t out[3]) {
    V3 rel = pos - bmin;
    float inv = 1.0f / fmax_(cell, 1e-8f);
    out[0] = f2u_sat(fmax_(0.0f, floorf(rel.x * inv)));
    out[1] = f2u_sat(fmax_(0.0f, floorf(rel.y * inv)));
    out[2] = f2u_sat(fmax_(0.0f, floorf(rel.z * inv)));
}

inline V3 clamp3(V3 v, V3 lo, V3 hi) { return vmin(vmax(v, lo), hi); }

// ---- materials at a triangle hit: raytracer_unified.wgsl:279-366 ---------------------------------------------

inline uint32_t wrap_texel(float f, uint32_t n) {
    const int i = f2i_sat(f);
    const int m = i % (int)n;
    return (uint32_t)(m < 0 ? m + (int)n : m);
}

// textureSampleLevel on the renderer's array (src/renderer.rs:196-259): repeat addressing, linear filter, mip 0,
// layer index clamped to the array. f32 weights: texel / 255, texel centres at integer + 0.5, horizontal blends
// first.
inline void sample_texture(const orc_scene& s, float u, float v, int32_t index, float out[4]) {
    if (s.tex_layers == 0 || !s.texels) {
        out[0] = out[1] = out[2] = out[3] = 1.0f;  // the 1x1 white dummy (src/renderer.rs:159-193)
        return;
    }
    const uint32_t layer = index < 0 ? 0u : ((uint32_t)index < s.tex_layers ? (uint32_t)index : s.tex_layers - 1u);
    const float x = u * (float)s.tex_w - 0.5f, y = v * (float)s.tex_h - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const uint32_t ix0 = wrap_texel(x0, s.tex_w), ix1 = wrap_texel(x0 + 1.0f, s.tex_w);
    const uint32_t iy0 = wrap_texel(y0, s.tex_h), iy1 = wrap_texel(y0 + 1.0f, s.tex_h);
    const uint8_t* base = s.texels + (size_t)layer * s.tex_w * s.tex_h * 4;
    const uint8_t* c00 = base + ((size_t)iy0 * s.tex_w + ix0) * 4;
    const uint8_t* c10 = base + ((size_t)iy0 * s.tex_w + ix1) * 4;
    const uint8_t* c01 = base + ((size_t)iy1 * s.tex_w + ix0) * 4;
    const uint8_t* c11 = base + ((size_t)iy1 * s.tex_w + ix1) * 4;
    for (int k = 0; k < 4; ++k) {
        const float top = ((float)c00[k] / 255.0f) * (1.0f - fx) + ((float)c10[k] / 255.0f) * fx;
        const float bot = ((float)c01[k] / 255.0f) * (1.0f - fx) + ((float)c11[k] / 255.0f) * fx;
        out[k] = top * (1.0f - fy) + bot * fy;
    }
}

// :283-285: uv = uv0*w + uv1*u + uv2*v, w = 1 - u - v
inline void hit_uv(const TriangleData& tr, float u, float v, float uv[2]) {
    const float w = 1.0f - u - v;
    uv[0] = (tr.uv0[0] * w + tr.uv1[0] * u) + tr.uv2[0] * v;
    uv[1] = (tr.uv0[1] * w + tr.uv1[1] * u) + tr.uv2[1] * v;
}

// :288-301
inline void base_color(const orc_scene& s, const MaterialData& m, const float uv[2], float out[4]) {
    if (m.texture_index >= 0) {
        float t[4];
        sample_texture(s, uv[0], uv[1], m.texture_index, t);
        for (int k = 0; k < 4; ++k) out[k] = t[k] * m.base_color[k];
    } else {
        for (int k = 0; k < 4; ++k) out[k] = m.base_color[k];
    }
}

// Alpha rules inside the shader's triangle test (:303-316): MASK discards the hit when base_color.a < alpha_cutoff,
// BLEND when base_color.a < 0.01. (u, v): barycentrics of the hit.
inline bool alpha_discards(const orc_scene& s, const TriangleData& tr, float u, float v) {
    if (!s.mats) return false;
    uint32_t mat = f2u_sat(tr.material_id);
    if (mat >= s.n_mats) return false;
    const MaterialData& m = s.mats[mat];
    if (m.alpha_mode != 1u && m.alpha_mode != 2u) return false;
    float uv[2], bc[4];
    hit_uv(tr, u, v, uv);
    base_color(s, m, uv, bc);
    if (m.alpha_mode == 1u) return bc[3] < m.alpha_cutoff;
    return bc[3] < 0.01f;
}

// What the triangle test leaves in the hit besides t (:317-366): colour, normal (normal-mapped), emissive.
struct Surface {
    V3 color, emissive, n;
};
inline Surface triangle_surface(const orc_scene& s, const MaterialData& m, const TriangleData& tr, float u, float v, V3 gn) {
    Surface sf;
    float uv[2], bc[4];
    hit_uv(tr, u, v, uv);
    base_color(s, m, uv, bc);
    sf.color = v3(bc[0], bc[1], bc[2]);
    sf.n = gn;
    if (m.normal_texture_index >= 0) {  // :319-348
        float ns[4];
        sample_texture(s, uv[0], uv[1], m.normal_texture_index, ns);
        const V3 tn = normalize(v3(ns[0] * 2.0f - 1.0f, ns[1] * 2.0f - 1.0f, ns[2] * 2.0f - 1.0f));
        const V3 up = fabsf(gn.y) > 0.9f ? v3(0.0f, 0.0f, 1.0f) : v3(0.0f, 1.0f, 0.0f);
        const V3 tangent = normalize(cross(up, gn));
        const V3 bitangent = cross(gn, tangent);
        sf.n = normalize((tangent * tn.x + bitangent * tn.y) + gn * tn.z);
    }
    sf.emissive = ld(m.emissive);
    if (m.emissive_texture_index >= 0) {  // :350-362
        float es[4];
        sample_texture(s, uv[0], uv[1], m.emissive_texture_index, es);
        sf.emissive = v3(sf.emissive.x * es[0], sf.emissive.y * es[1], sf.emissive.z * es[2]);
    }
    return sf;
}

// raytracer_unified.wgsl:430-588, geometry only. any_hit: stop at the first accepted intersection.
Hit trace_ray(const orc_scene& s, const rt_params& p, const CameraUniform& cam, const FrameConsts& fc, V3 o, V3 d,
              bool any_hit, Counters& ctr) {
    const int wgsl = (int)p.wgsl_compat;
    Hit closest;
    closest.hit = false;
    closest.t = 1e10f;
    ctr.steps = 0;

    auto consider = [&](const Hit& h) -> bool {  // returns true when traversal may stop (any-hit)
        if (h.hit && h.t < closest.t) {
            closest = h;
            if (any_hit) return true;
        }
        return false;
    };

    // moving boxes tested directly: :442-461
    if (s.nb > 0) {
        uint32_t start = s.nb >= p.moving_box_count ? s.nb - p.moving_box_count : 0u;
        for (uint32_t i = start; i < s.nb; ++i) {
            ctr.steps++;
            const BoxData& b = s.boxes[i];
            V3 center = mix(ld(b.center0), ld(b.center1), fc.t_lerp);
            V3 size = ld(b.half_size) * 2.0f;
            if (!(p.lod_cull && should_cull_lod(cam, center, size))) {
                ctr.box++;
                if (consider(intersect_box(o, d, b, fc.t_lerp, i, wgsl))) return closest;
            }
        }
    }

    const float cell_size = s.meta.finest_cell_size;
    const uint32_t gx = s.meta.grid_sizes[RT_GRID_LEVELS - 1][0];
    const uint32_t gy = s.meta.grid_sizes[RT_GRID_LEVELS - 1][1];
    const uint32_t gz = s.meta.grid_sizes[RT_GRID_LEVELS - 1][2];
    const V3 bmin = ld(s.meta.bounds_min), bmax = ld(s.meta.bounds_max);

    // grid entry: :467-489. A ray that misses the grid returns "no hit" even if a moving box was hit.
    V3 ray_pos = o;
    float t_offset = 0.0f;
    if (ray_pos.x < bmin.x || ray_pos.x > bmax.x || ray_pos.y < bmin.y || ray_pos.y > bmax.y || ray_pos.z < bmin.z ||
        ray_pos.z > bmax.z) {
        float t_entry = intersect_aabb(o, d, bmin, bmax, wgsl);
        if (t_entry < 0.0f) return Hit{};
        t_offset = t_entry + 0.001f;
        ray_pos = o + d * t_offset;
        const float e = 0.0001f;
        ray_pos = clamp3(ray_pos, bmin + v3(e, e, e), bmax - v3(e, e, e));
    }

    // DDA setup: :492-517
    uint32_t cell[3];
    world_to_cell_u(ray_pos, bmin, cell_size, cell);
    const int32_t step[3] = {d.x >= 0.0f ? 1 : -1, d.y >= 0.0f ? 1 : -1, d.z >= 0.0f ? 1 : -1};
    V3 cell_pos_world = bmin + v3((float)cell[0], (float)cell[1], (float)cell[2]) * cell_size;
    V3 next_boundary = cell_pos_world + v3(step[0] > 0 ? cell_size : 0.0f, step[1] > 0 ? cell_size : 0.0f,
                                           step[2] > 0 ? cell_size : 0.0f);
    V3 inv = v3(safe_inv(d.x, wgsl), safe_inv(d.y, wgsl), safe_inv(d.z, wgsl));
    V3 t_delta = vabs(inv * cell_size);  // abs(cell_size * safe_inv_dir)
    V3 t_max = v3(t_offset, t_offset, t_offset) + (next_boundary - ray_pos) * inv;
    float floor_t = t_offset + 0.00001f;
    t_max = vmax(t_max, v3(floor_t, floor_t, floor_t));

    // traversal: :520-588
    for (uint32_t i = 0; i < p.max_dda_steps; ++i) {
        ctr.steps++;
        ctr.cells++;
        // get_fine_index (:406-409) in wrapping u32 arithmetic; an index past the array reads as an empty cell
        uint32_t fine_idx = cell[0] + cell[1] * gx + cell[2] * gx * gy;
        uint32_t begin = 0, count = 0;
        if ((uint64_t)fine_idx < s.n_cells) {
            begin = s.cell_start[fine_idx];
            count = s.cell_start[fine_idx + 1] - begin;
        }
        for (uint32_t j = 0; j < count && j < p.max_objects_per_cell; ++j) {
            uint32_t obj = s.indices[begin + j];
            if (obj < s.nb) {
                const BoxData& b = s.boxes[obj];
                V3 center = (ld(b.min) + ld(b.max)) * 0.5f;
                V3 size = ld(b.max) - ld(b.min);
                if (!(p.lod_cull && should_cull_lod(cam, center, size))) {
                    ctr.box++;
                    if (consider(intersect_box(o, d, b, fc.t_lerp, obj, wgsl))) return closest;
                }
            } else {
                uint32_t ti = obj - s.nb;
                if (ti < s.nt) {
                    ctr.tri++;
                    const TriangleData& tr = s.tris[ti];
                    TriHit th;
                    if (moller_trumbore(o, d, ld(tr.v0), ld(tr.v1), ld(tr.v2), wgsl, &th) && !alpha_discards(s, tr, th.u, th.v)) {
                        Hit h;
                        h.hit = true;
                        h.t = th.t;
                        h.id = obj;
                        h.pos = o + d * th.t;
                        h.n = th.n;
                        h.u = th.u;
                        h.v = th.v;
                        if (consider(h)) return closest;
                    }
                }
            }
        }
        if (closest.hit && closest.t < fmin_(fmin_(t_max.x, t_max.y), t_max.z)) break;

        if (t_max.x < t_max.y && t_max.x < t_max.z) {
            int32_t nx = (int32_t)cell[0] + step[0];
            if (nx < 0 || nx >= (int32_t)gx) break;
            cell[0] = (uint32_t)nx;
            t_max.x += t_delta.x;
        } else if (t_max.y < t_max.z) {
            int32_t ny = (int32_t)cell[1] + step[1];
            if (ny < 0 || ny >= (int32_t)gy) break;
            cell[1] = (uint32_t)ny;
            t_max.y += t_delta.y;
        } else {
            int32_t nz = (int32_t)cell[2] + step[2];
            if (nz < 0 || nz >= (int32_t)gz) break;
            cell[2] = (uint32_t)nz;
            t_max.z += t_delta.z;
        }
    }
    if (!closest.hit) return Hit{};
    return closest;
}

// raytracer_unified.wgsl:634-653
inline V3 primary_dir(const CameraUniform& cam, const FrameConsts& fc, uint32_t px, uint32_t py, uint32_t w,
                      uint32_t h) {
    float u = ((float)px + 0.5f) / (float)w;
    float v = ((float)py + 0.5f) / (float)h;
    float ndc_x = u * 2.0f - 1.0f;
    float ndc_y = v * 2.0f - 1.0f;
    float aspect = (float)w / (float)h;
    V3 a = ld(cam.forward);
    V3 b = ((ld(cam.right) * ndc_x) * aspect) * fc.fov_scale;
    V3 c = (ld(cam.up) * (-ndc_y)) * fc.fov_scale;
    return normalize((a + b) + c);
}

struct ShardMap {
    uint32_t tile_w, tile_h, first, stride, tiles_x, tiles_y, n_tiles, n_local;
    std::vector<uint32_t> tiles;  // the frame tiles of this shard, in increasing order
    uint32_t tile(uint32_t lt) const { return tiles[lt]; }
};
ShardMap shard_map(uint32_t w, uint32_t h, const rt_shard* sh) {
    ShardMap m;
    m.tile_w = sh ? sh->tile_w : w;
    m.tile_h = sh ? sh->tile_h : h;
    m.first = sh ? sh->first : 0;
    m.stride = sh ? sh->stride : 1;
    m.tiles_x = (w + m.tile_w - 1) / m.tile_w;
    m.tiles_y = (h + m.tile_h - 1) / m.tile_h;
    m.n_tiles = m.tiles_x * m.tiles_y;
    // tile t belongs to the shard when it is tile `first` of its round of `stride` tiles, or, with an owner table,
    // when the table names `first` at t's place in the round
    for (uint32_t t = 0; t < m.n_tiles; ++t) {
        const uint32_t o = t % m.stride;
        if (sh && sh->owner ? sh->owner[o] == m.first : (t >= m.first && (t - m.first) % m.stride == 0)) m.tiles.push_back(t);
    }
    m.n_local = (uint32_t)m.tiles.size();
    return m;
}

template <class F>
void parallel_for(uint64_t n, int n_threads, F&& f) {
    if (n_threads <= 1 || n < 2) {
        f(0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    uint64_t chunk = (n + n_threads - 1) / n_threads;
    for (int k = 0; k < n_threads; ++k) {
        uint64_t lo = std::min<uint64_t>(n, k * chunk), hi = std::min<uint64_t>(n, lo + chunk);
        if (lo < hi) th.emplace_back([=, &f] { f(lo, hi, k); });
    }
    for (auto& t : th) t.join();
}

}  // namespace

extern "C" uint64_t orc_shard_pixels(uint32_t width, uint32_t height, const rt_shard* shard) {
    ShardMap m = shard_map(width, height, shard);
    return (uint64_t)m.n_local * m.tile_w * m.tile_h;
}

extern "C" void orc_trace_primary(const orc_scene* s, const rt_params* p, const CameraUniform* cam, uint32_t width,
                                  uint32_t height, const rt_shard* shard, rt_hit* out_primary, rt_hit* out_shadow,
                                  float* shadow_o, float* shadow_d, rt_stats* stats, int n_threads) {
    const ShardMap m = shard_map(width, height, shard);
    const FrameConsts fc = frame_consts(*p, cam);
    const uint64_t per_tile = (uint64_t)m.tile_w * m.tile_h;
    const uint64_t rows = (uint64_t)m.n_local * m.tile_h;  // parallelise over tile rows
    std::vector<rt_stats> part((size_t)std::max(1, n_threads));
    for (auto& q : part) memset(&q, 0, sizeof(q));
    parallel_for(rows, n_threads, [&](uint64_t lo, uint64_t hi, int k) {
        rt_stats& st_ = part[(size_t)k];
        for (uint64_t r = lo; r < hi; ++r) {
            uint32_t lt = (uint32_t)(r / m.tile_h), ly = (uint32_t)(r % m.tile_h);
            uint32_t tile = m.tile(lt);
            uint32_t ty = tile / m.tiles_x, tx = tile % m.tiles_x;
            uint32_t py = ty * m.tile_h + ly;
            for (uint32_t lx = 0; lx < m.tile_w; ++lx) {
                uint32_t px = tx * m.tile_w + lx;
                uint64_t oi = (uint64_t)lt * per_tile + (uint64_t)ly * m.tile_w + lx;
                rt_hit ph{RT_MISS, 0.0f}, sh{RT_MISS, 0.0f};
                V3 so = v3(0, 0, 0), sd = v3(0, 0, 0);
                if (px < width && py < height) {
                    V3 o = ld(cam->position);
                    V3 d = primary_dir(*cam, fc, px, py, width, height);
                    Counters c;
                    Hit h = trace_ray(*s, *p, *cam, fc, o, d, false, c);
                    st_.rays++;
                    if (h.hit) {
                        st_.primary_hits++;
                        ph.id = h.id;
                        ph.t = h.t;
                        if (out_shadow || shadow_o) {
                            V3 n = h.n;
                            if (dot(n, d) > 0.0f) n = neg(n);
                            so = h.pos + n * p->shadow_offset;
                            sd = fc.shadow_dir;
                            Hit oh = trace_ray(*s, *p, *cam, fc, so, sd, true, c);
                            st_.rays++;
                            st_.shadow_rays++;
                            if (oh.hit) {
                                st_.shadow_hits++;
                                sh.id = oh.id;
                                sh.t = oh.t;
                            }
                        }
                    }
                    st_.cells_visited += c.cells;
                    st_.tri_tests += c.tri;
                    st_.box_tests += c.box;
                }
                out_primary[oi] = ph;
                if (out_shadow) out_shadow[oi] = sh;
                if (shadow_o) st(shadow_o + 3 * oi, so);
                if (shadow_d) st(shadow_d + 3 * oi, sd);
            }
        }
    });
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (auto& q : part) {
            stats->rays += q.rays;
            stats->primary_hits += q.primary_hits;
            stats->shadow_rays += q.shadow_rays;
            stats->shadow_hits += q.shadow_hits;
            stats->cells_visited += q.cells_visited;
            stats->tri_tests += q.tri_tests;
            stats->box_tests += q.box_tests;
        }
    }
}

// ---- linear-blend skinning (glTF 2.0; the reference has none). Same operation order as k_skin in the GPU library:
// S = ((w0*M[j0] + w1*M[j1]) + w2*M[j2]) + w3*M[j3] element by element (column-major 4x4 per joint), then
// ((S.x*p.x + S.y*p.y) + S.z*p.z) + S.w.
extern "C" void orc_skin_vertices(const float* positions, const uint16_t* joints, const float* weights,
                                  const uint32_t* tri_indices, uint32_t n_triangles, const float* joint_matrices,
                                  float* out_xyz9) {
    for (uint64_t k = 0; k < (uint64_t)n_triangles * 3; ++k) {
        const uint32_t v = tri_indices[k];
        const V3 p = ld(positions + 3 * (uint64_t)v);
        const uint16_t* j = joints + 4 * (uint64_t)v;
        const float* w = weights + 4 * (uint64_t)v;
        V3 col[4];
        for (int c = 0; c < 4; ++c) {
            V3 m[4];
            for (int q = 0; q < 4; ++q) m[q] = ld(joint_matrices + 16 * (uint64_t)j[q] + 4 * c);
            col[c] = ((m[0] * w[0] + m[1] * w[1]) + m[2] * w[2]) + m[3] * w[3];
        }
        st(out_xyz9 + 3 * k, ((col[0] * p.x + col[1] * p.y) + col[2] * p.z) + col[3]);
    }
}

// ---- shaded picture: raytracer_unified.wgsl:590-621 (sky, lighting), :655-720 (bounce loop, debug pixel,
// texture store), :230-233 (box colour), :279-366 (triangle material: base-colour texture, normal map, emissive
// texture; the alpha rules are part of trace_ray above).
namespace {

inline uint32_t unorm8(float c) {
    float x = fmin_(fmax_(c, 0.0f), 1.0f);
    return (uint32_t)rintf(x * 255.0f);  // round to nearest even
}

// :419-428 (WGSL % on f32 = truncated remainder)
inline bool near_grid_boundary(const orc_scene& s, V3 pos, float threshold) {
    V3 rel = pos - ld(s.meta.bounds_min);
    float cs = s.meta.finest_cell_size;
    V3 local = v3(fmodf(rel.x, cs) / cs, fmodf(rel.y, cs) / cs, fmodf(rel.z, cs) / cs);
    float dx = fmin_(local.x, 1.0f - local.x), dy = fmin_(local.y, 1.0f - local.y), dz = fmin_(local.z, 1.0f - local.z);
    return dx < threshold || dy < threshold || dz < threshold;
}

}  // namespace

extern "C" void orc_sample_texture(const orc_scene* s, float u, float v, int32_t layer, float out_rgba[4]) {
    sample_texture(*s, u, v, layer, out_rgba);
}

extern "C" void orc_render(const orc_scene* s, const rt_params* p, const CameraUniform* cam, uint32_t width, uint32_t height,
                           const rt_shard* shard, const MaterialData* mats, uint32_t n_mats, const rt_render_opts* opts,
                           uint8_t* out_rgba8, rt_stats* stats, int n_threads) {
    const ShardMap m = shard_map(width, height, shard);
    const FrameConsts fc = frame_consts(*p, cam);
    const uint64_t per_tile = (uint64_t)m.tile_w * m.tile_h;
    const uint64_t rows = (uint64_t)m.n_local * m.tile_h;
    rt_render_opts o;
    memset(&o, 0, sizeof(o));
    o.max_bounces = 8;
    if (opts) o = *opts;
    std::vector<rt_stats> part((size_t)std::max(1, n_threads));
    for (auto& q : part) memset(&q, 0, sizeof(q));
    uint32_t* out = reinterpret_cast<uint32_t*>(out_rgba8);
    parallel_for(rows, n_threads, [&](uint64_t lo, uint64_t hi, int k) {
        rt_stats& st_ = part[(size_t)k];
        for (uint64_t r = lo; r < hi; ++r) {
            uint32_t lt = (uint32_t)(r / m.tile_h), ly = (uint32_t)(r % m.tile_h);
            uint32_t tile = m.tile(lt);
            uint32_t ty = tile / m.tiles_x, tx = tile % m.tiles_x;
            uint32_t py = ty * m.tile_h + ly;
            for (uint32_t lx = 0; lx < m.tile_w; ++lx) {
                uint32_t px = tx * m.tile_w + lx;
                uint64_t oi = (uint64_t)lt * per_tile + (uint64_t)ly * m.tile_w + lx;
                uint32_t rgba = 0;
                if (px < width && py < height) {
                    V3 ro = ld(cam->position);
                    V3 rd = primary_dir(*cam, fc, px, py, width, height);
                    V3 acc = v3(0, 0, 0);
                    float mult = 1.0f;
                    Counters c;
                    for (uint32_t bounce = 0; bounce < o.max_bounces; ++bounce) {
                        Hit h = trace_ray(*s, *p, *cam, fc, ro, rd, false, c);
                        st_.rays++;
                        if (!h.hit) {
                            float t = (rd.y + 1.0f) * 0.5f;
                            acc = acc + mix(v3(0.3f, 0.5f, 0.7f), v3(0.5f, 0.7f, 1.0f), t) * mult;
                            break;
                        }
                        if (bounce == 0) st_.primary_hits++;
                        V3 color = v3(0.7f, 0.7f, 0.7f), emissive = v3(0, 0, 0);
                        float refl = 0.0f;
                        if (h.id < s->nb) {
                            color = ld(s->boxes[h.id].color);
                            refl = s->boxes[h.id].reflectivity;
                        } else {
                            const TriangleData& tr = s->tris[h.id - s->nb];
                            uint32_t mat = f2u_sat(tr.material_id);
                            if (mat < n_mats) {
                                const Surface sf = triangle_surface(*s, mats[mat], tr, h.u, h.v, h.n);
                                color = sf.color;
                                emissive = sf.emissive;
                                h.n = sf.n;  // the normal-mapped normal is the hit's normal from here on (:347)
                                refl = mats[mat].metallic;
                            }
                        }
                        float diffuse = fmax_(dot(h.n, fc.shadow_dir), 0.0f);
                        if (o.shadows) {
                            V3 nf = h.n;
                            if (dot(nf, rd) > 0.0f) nf = neg(nf);
                            Hit oh = trace_ray(*s, *p, *cam, fc, h.pos + nf * p->shadow_offset, fc.shadow_dir, true, c);
                            st_.rays++;
                            st_.shadow_rays++;
                            if (oh.hit) {
                                st_.shadow_hits++;
                                diffuse = 0.0f;
                            }
                        }
                        V3 lit = color * (0.3f + diffuse * 0.7f) + emissive;
                        if (cam->show_grid > 0.5f && near_grid_boundary(*s, h.pos, 0.02f)) lit = mix(lit, v3(0, 1, 0), 0.6f);
                        acc = acc + (lit * (1.0f - refl)) * mult;
                        if (refl < 0.01f) break;
                        mult *= refl;
                        rd = rd - h.n * (2.0f * dot(h.n, rd));
                        ro = h.pos + h.n * 0.001f;
                    }
                    if (o.debug_enabled) {
                        int dx = (int)px - (int)o.debug_pixel[0], dy = (int)py - (int)o.debug_pixel[1];
                        if (dx == 0 && dy == 0)
                            acc = v3(1, 1, 0);
                        else if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1)
                            acc = mix(acc, v3(1, 1, 0), 0.5f);
                    }
                    rgba = unorm8(acc.x) | (unorm8(acc.y) << 8) | (unorm8(acc.z) << 16) | 0xFF000000u;
                }
                out[oi] = rgba;
            }
        }
    });
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (auto& q : part) {
            stats->rays += q.rays;
            stats->primary_hits += q.primary_hits;
            stats->shadow_rays += q.shadow_rays;
            stats->shadow_hits += q.shadow_hits;
        }
    }
}

extern "C" void orc_primary_rays(const rt_params* p, const CameraUniform* cam, uint32_t width, uint32_t height,
                                 float* origins, float* dirs) {
    const FrameConsts fc = frame_consts(*p, cam);
    for (uint32_t py = 0; py < height; ++py)
        for (uint32_t px = 0; px < width; ++px) {
            uint64_t i = (uint64_t)py * width + px;
            st(origins + 3 * i, ld(cam->position));
            st(dirs + 3 * i, primary_dir(*cam, fc, px, py, width, height));
        }
}

static CameraUniform null_camera() {
    CameraUniform c;
    memset(&c, 0, sizeof(c));
    return c;
}

extern "C" void orc_trace_rays(const orc_scene* s, const rt_params* p, const CameraUniform* cam, const float* origins,
                               const float* dirs, uint32_t n, rt_hit* out, int any_hit, rt_stats* stats,
                               int n_threads) {
    CameraUniform nc = null_camera();
    const CameraUniform& c = cam ? *cam : nc;
    const FrameConsts fc = frame_consts(*p, cam);
    std::vector<rt_stats> part((size_t)std::max(1, n_threads));
    for (auto& q : part) memset(&q, 0, sizeof(q));
    parallel_for(n, n_threads, [&](uint64_t lo, uint64_t hi, int k) {
        rt_stats& st_ = part[(size_t)k];
        for (uint64_t i = lo; i < hi; ++i) {
            Counters ctr;
            Hit h = trace_ray(*s, *p, c, fc, ld(origins + 3 * i), ld(dirs + 3 * i), any_hit != 0, ctr);
            out[i].id = h.hit ? h.id : RT_MISS;
            out[i].t = h.hit ? h.t : 0.0f;
            st_.rays++;
            st_.primary_hits += h.hit ? 1 : 0;
            st_.cells_visited += ctr.cells;
            st_.tri_tests += ctr.tri;
            st_.box_tests += ctr.box;
        }
    });
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (auto& q : part) {
            stats->rays += q.rays;
            stats->primary_hits += q.primary_hits;
            stats->cells_visited += q.cells_visited;
            stats->tri_tests += q.tri_tests;
            stats->box_tests += q.box_tests;
        }
    }
}

// Closest hit over ALL objects in ascending id order with the same accept rules (strict <, initial 1e10):
// what the grid traversal must return whenever the 200-step cap, the per-cell cap and the LOD cull do not bite
// and the winner is not a later-found equal-t duplicate.
extern "C" void orc_trace_rays_bruteforce(const orc_scene* s, const rt_params* p, const CameraUniform* cam,
                                          const float* origins, const float* dirs, uint32_t n, rt_hit* out,
                                          int n_threads) {
    CameraUniform nc = null_camera();
    const CameraUniform& c = cam ? *cam : nc;
    const FrameConsts fc = frame_consts(*p, cam);
    const int wgsl = (int)p->wgsl_compat;
    parallel_for(n, n_threads, [&](uint64_t lo, uint64_t hi, int) {
        for (uint64_t i = lo; i < hi; ++i) {
            V3 o = ld(origins + 3 * i), d = ld(dirs + 3 * i);
            float best = 1e10f;
            uint32_t id = RT_MISS;
            for (uint32_t b = 0; b < s->nb; ++b) {
                const BoxData& bx = s->boxes[b];
                V3 center = (ld(bx.min) + ld(bx.max)) * 0.5f, size = ld(bx.max) - ld(bx.min);
                if (p->lod_cull && should_cull_lod(c, center, size)) continue;
                Hit h = intersect_box(o, d, bx, fc.t_lerp, b, wgsl);
                if (h.hit && h.t < best) {
                    best = h.t;
                    id = b;
                }
            }
            for (uint32_t k = 0; k < s->nt; ++k) {
                const TriangleData& tr = s->tris[k];
                TriHit th;
                if (moller_trumbore(o, d, ld(tr.v0), ld(tr.v1), ld(tr.v2), wgsl, &th) && th.t < best && !alpha_discards(*s, tr, th.u, th.v)) {
                    best = th.t;
                    id = s->nb + k;
                }
            }
            out[i].id = id;
            out[i].t = id == RT_MISS ? 0.0f : best;
        }
    });
}

// raytracer_unified.wgsl:695-709 (first bounce only; colour fields are outside the trace core)
extern "C" void orc_debug_pixel(const orc_scene* s, const rt_params* p, const CameraUniform* cam, uint32_t width,
                                uint32_t height, uint32_t px, uint32_t py, RayDebugInfo* info) {
    const FrameConsts fc = frame_consts(*p, cam);
    memset(info, 0, sizeof(*info));
    V3 o = ld(cam->position);
    V3 d = primary_dir(*cam, fc, px, py, width, height);
    Counters c;
    Hit h = trace_ray(*s, *p, *cam, fc, o, d, false, c);
    st(info->ray_origin, o);
    st(info->ray_direction, d);
    info->hit = h.hit ? 1.0f : 0.0f;
    info->distance = h.hit ? h.t : 1e10f;
    st(info->hit_position, h.hit ? h.pos : v3(0, 0, 0));
    st(info->hit_normal, h.hit ? h.n : v3(0, 0, 0));
    info->object_id = h.hit ? (float)h.id : -1.0f;
    info->num_steps = (float)c.steps;
}

// ====================== reference CPU trace path: BVH (src/core/bvh.rs) ======================

namespace {
struct Box3 {
    V3 mn, mx;
};
inline Box3 bunion(Box3 a, Box3 b) { return Box3{vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; }
inline float barea(Box3 b) {
    V3 d = b.mx - b.mn;
    return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}
inline float comp(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

struct BvhNode {
    Box3 bounds;
    int32_t left = -1, right = -1;   // children (internal)
    uint32_t first = 0, count = 0;   // leaf: range in prim_indices
};
}  // namespace

struct orc_bvh {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> prim_indices;
    std::vector<Box3> pb;  // primitive bounds
    std::vector<V3> pc;    // primitive centroids = bounds.center()
    uint64_t n_leaves = 0, max_depth = 0, total_prims = 0;
};

namespace {
constexpr size_t MAX_LEAF_SIZE = 4;  // src/core/bvh.rs:5
constexpr int SAH_BUCKETS = 12;      // src/core/bvh.rs:8

// src/core/bvh.rs:118-178
void evaluate_sah_axis(const orc_bvh& b, const std::vector<uint32_t>& idx, Box3 bounds, int axis, float* cost,
                       float* pos) {
    bool has[SAH_BUCKETS] = {};
    Box3 bb[SAH_BUCKETS];
    size_t cnt[SAH_BUCKETS] = {};
    V3 extent = bounds.mx - bounds.mn;
    float axis_extent = comp(extent, axis);
    if (axis_extent < 1e-6f) {
        *cost = INFINITY;
        *pos = 0.0f;
        return;
    }
    for (uint32_t i : idx) {
        float offset = (comp(b.pc[i], axis) - comp(bounds.mn, axis)) / axis_extent;
        size_t bi = std::min<size_t>((size_t)f2usize_sat(offset * (float)SAH_BUCKETS), SAH_BUCKETS - 1);
        cnt[bi]++;
        bb[bi] = has[bi] ? bunion(bb[bi], b.pb[i]) : b.pb[i];
        has[bi] = true;
    }
    float best_cost = INFINITY;
    int best_split = 0;
    auto accumulate = [&](int s, int e, bool* any, Box3* out, size_t* total) {
        *any = false;
        *total = 0;
        for (int i = s; i < e; ++i)
            if (has[i]) {
                *out = *any ? bunion(*out, bb[i]) : bb[i];
                *any = true;
                *total += cnt[i];
            }
    };
    for (int split = 1; split < SAH_BUCKETS; ++split) {
        bool la, ra;
        Box3 lb{}, rb{};
        size_t lc, rc;
        accumulate(0, split, &la, &lb, &lc);
        accumulate(split, SAH_BUCKETS, &ra, &rb, &rc);
        if (la && ra) {
            // sah_cost: src/core/bvh.rs:203-209
            float c = 0.125f + 1.0f * (barea(lb) * (float)lc + barea(rb) * (float)rc);
            if (c < best_cost) {
                best_cost = c;
                best_split = split;
            }
        }
    }
    *cost = best_cost;
    *pos = comp(bounds.mn, axis) + ((float)best_split / (float)SAH_BUCKETS) * axis_extent;
}

// src/core/bvh.rs:49-92
int32_t build_recursive(orc_bvh& b, std::vector<uint32_t> idx, size_t depth) {
    Box3 bounds = b.pb[idx[0]];
    for (uint32_t i : idx) bounds = bunion(bounds, b.pb[i]);
    auto make_leaf = [&]() {
        BvhNode n;
        n.bounds = bounds;
        n.first = (uint32_t)b.prim_indices.size();
        n.count = (uint32_t)idx.size();
        b.prim_indices.insert(b.prim_indices.end(), idx.begin(), idx.end());
        b.nodes.push_back(n);
        b.n_leaves++;
        b.total_prims += idx.size();
        b.max_depth = std::max<uint64_t>(b.max_depth, depth);
        return (int32_t)b.nodes.size() - 1;
    };
    if (idx.size() <= MAX_LEAF_SIZE) return make_leaf();
    float best_cost = INFINITY, best_pos = 0.0f;
    int best_axis = 0;
    for (int axis = 0; axis < 3; ++axis) {
        float c, p;
        evaluate_sah_axis(b, idx, bounds, axis, &c, &p);
        if (c < best_cost) {
            best_cost = c;
            best_axis = axis;
            best_pos = p;
        }
    }
    // partition_primitives: src/core/bvh.rs:212-232
    size_t left = 0, right = idx.size();
    while (left < right) {
        if (comp(b.pc[idx[left]], best_axis) < best_pos) {
            left++;
        } else {
            right--;
            std::swap(idx[left], idx[right]);
        }
    }
    size_t mid = left;
    if (mid == 0 || mid == idx.size()) return make_leaf();
    std::vector<uint32_t> ridx(idx.begin() + (ptrdiff_t)mid, idx.end());
    idx.resize(mid);
    int32_t self = (int32_t)b.nodes.size();
    b.nodes.push_back(BvhNode{});
    b.nodes[(size_t)self].bounds = bounds;
    int32_t l = build_recursive(b, std::move(idx), depth + 1);
    int32_t r = build_recursive(b, std::move(ridx), depth + 1);
    b.nodes[(size_t)self].left = l;
    b.nodes[(size_t)self].right = r;
    return self;
}

// examples/massive_benchmark.rs:311-320: plain 1/dir, no epsilon
inline bool bvh_aabb(Box3 b, V3 o, V3 d) {
    V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    V3 t1 = (b.mn - o) * inv, t2 = (b.mx - o) * inv;
    V3 lo = vmin(t1, t2), hi = vmax(t1, t2);
    float tmin = fmax_(fmax_(lo.x, lo.y), lo.z);  // glam max_element: x.max(y.max(z)); same value
    float tmax = fmin_(fmin_(hi.x, hi.y), hi.z);
    return tmax >= tmin && tmax >= 0.0f;
}

// examples/massive_benchmark.rs:226-270: both children always visited, no ordering, no pruning
bool traverse(const orc_bvh& b, int32_t ni, const TriangleData* tris, V3 o, V3 d, float* out_t) {
    const BvhNode& n = b.nodes[(size_t)ni];
    if (!bvh_aabb(n.bounds, o, d)) return false;
    if (n.left < 0) {
        bool any = false;
        float closest_t = INFINITY;
        for (uint32_t k = 0; k < n.count; ++k) {
            const TriangleData& tr = tris[b.prim_indices[n.first + k]];
            TriHit h;
            if (moller_trumbore(o, d, ld(tr.v0), ld(tr.v1), ld(tr.v2), 0, &h) && h.t < closest_t) {
                closest_t = h.t;
                any = true;
            }
        }
        *out_t = closest_t;
        return any;
    }
    float tl, tr_;
    bool hl = traverse(b, n.left, tris, o, d, &tl);
    bool hr = traverse(b, n.right, tris, o, d, &tr_);
    if (hl && hr) {
        *out_t = fmin_(tl, tr_);
        return true;
    }
    if (hl) {
        *out_t = tl;
        return true;
    }
    if (hr) {
        *out_t = tr_;
        return true;
    }
    return false;
}
}  // namespace

extern "C" orc_bvh* orc_bvh_build(const TriangleData* tris, uint32_t nt) {
    if (nt == 0) return nullptr;
    auto b = std::make_unique<orc_bvh>();
    b->pb.resize(nt);
    b->pc.resize(nt);
    for (uint32_t i = 0; i < nt; ++i) {
        tri_bounds(tris[i], &b->pb[i].mn, &b->pb[i].mx);
        b->pc[i] = (b->pb[i].mn + b->pb[i].mx) * 0.5f;
    }
    std::vector<uint32_t> idx(nt);
    for (uint32_t i = 0; i < nt; ++i) idx[i] = i;
    build_recursive(*b, std::move(idx), 0);
    return b.release();
}
extern "C" void orc_bvh_free(orc_bvh* b) { delete b; }
extern "C" void orc_bvh_stats(const orc_bvh* b, uint64_t* nodes, uint64_t* leaves, uint64_t* max_depth,
                              uint64_t* prims) {
    if (nodes) *nodes = b->nodes.size();
    if (leaves) *leaves = b->n_leaves;
    if (max_depth) *max_depth = b->max_depth;
    if (prims) *prims = b->total_prims;
}

extern "C" uint64_t orc_bvh_trace(const orc_bvh* b, const TriangleData* tris, const float* origins, const float* dirs,
                                  uint32_t n, float* out_t, int n_threads) {
    std::atomic<uint64_t> hits{0};
    parallel_for(n, n_threads, [&](uint64_t lo, uint64_t hi, int) {
        uint64_t local = 0;
        for (uint64_t i = lo; i < hi; ++i) {
            float t;
            bool h = traverse(*b, 0, tris, ld(origins + 3 * i), ld(dirs + 3 * i), &t);
            if (out_t) out_t[i] = h ? t : -1.0f;
            local += h ? 1 : 0;
        }
        hits += local;
    });
    return hits.load();
}

// examples/massive_benchmark.rs:322-342
extern "C" uint64_t orc_linear_scan(const TriangleData* tris, uint32_t nt, const float* origins, const float* dirs,
                                    uint32_t n, float* out_t, int n_threads) {
    std::atomic<uint64_t> hits{0};
    parallel_for(n, n_threads, [&](uint64_t lo, uint64_t hi, int) {
        uint64_t local = 0;
        for (uint64_t i = lo; i < hi; ++i) {
            V3 o = ld(origins + 3 * i), d = ld(dirs + 3 * i);
            float closest_t = INFINITY;
            bool any = false;
            for (uint32_t k = 0; k < nt; ++k) {
                TriHit h;
                if (moller_trumbore(o, d, ld(tris[k].v0), ld(tris[k].v1), ld(tris[k].v2), 0, &h) && h.t < closest_t) {
                    closest_t = h.t;
                    any = true;
                }
            }
            if (out_t) out_t[i] = any ? closest_t : -1.0f;
            local += any ? 1 : 0;
        }
        hits += local;
    });
    return hits.load();
}

// =================================== workload generators ===================================

static void make_tri(TriangleData* t, V3 v0, V3 v1, V3 v2, float u0x, float u0y, float u1x, float u1y, float u2x,
                     float u2y, uint32_t material) {
    memset(t, 0, sizeof(*t));
    st(t->v0, v0);
    st(t->v1, v1);
    st(t->v2, v2);
    t->material_id = (float)material;
    t->uv0[0] = u0x;
    t->uv0[1] = u0y;
    t->uv1[0] = u1x;
    t->uv1[1] = u1y;
    t->uv2[0] = u2x;
    t->uv2[1] = u2y;
}

// src/core/benchmark.rs:99-123 (usize arithmetic, then `as f32 / 100.0 - 50.0`)
extern "C" void orc_generate_test_triangles(uint64_t count, TriangleData* out) {
    for (uint64_t i = 0; i < count; ++i) {
        float x = (float)((i * 7919) % 10000) / 100.0f - 50.0f;
        float y = (float)((i * 6547) % 10000) / 100.0f - 50.0f;
        float z = (float)((i * 4231) % 10000) / 100.0f - 50.0f;
        make_tri(&out[i], v3(x, y, z), v3(x + 1.0f, y, z), v3(x, y + 1.0f, z), 0, 0, 1, 0, 0, 1, 0);
    }
}

// src/core/benchmark.rs:126-140
extern "C" void orc_generate_test_rays(uint64_t count, float* origins, float* dirs) {
    const float pi = 3.14159265358979323846f;
    for (uint64_t i = 0; i < count; ++i) {
        float angle = ((float)i / (float)count) * 2.0f * pi;
        float height = (float)((i * 997) % count) / (float)count * 100.0f - 50.0f;
        V3 o = v3(100.0f * cosf(angle), height, 100.0f * sinf(angle));
        V3 d = normalize(v3(0, 0, 0) - o);
        st(origins + 3 * i, o);
        st(dirs + 3 * i, d);
    }
}

// benches/bounding_volumes.rs:39-47
extern "C" void orc_random_unit_vector(uint32_t seed, float out[3]) {
    const float pi = 3.14159265358979323846f;
    float theta = fmodf((float)seed * 0.123456f, 2.0f * pi);
    float phi = fmodf((float)seed * 0.789012f, pi);
    out[0] = sinf(phi) * cosf(theta);
    out[1] = sinf(phi) * sinf(theta);
    out[2] = cosf(phi);
}

// benches/bounding_volumes.rs:151-160
extern "C" void orc_particle_aabbs(uint32_t count, float half, float* box_min, float* box_max) {
    for (uint32_t i = 0; i < count; ++i) {
        float x = fmodf((float)i * 0.1f, 20.0f) - 10.0f;
        float y = fmodf((float)i * 0.2f, 20.0f) - 10.0f;
        float z = -fmodf((float)i * 0.3f, 50.0f) - 10.0f;
        V3 c = v3(x, y, z), h = v3(half, half, half);
        st(box_min + 3 * i, c - h);
        st(box_max + 3 * i, c + h);
    }
}

// src/scenes/pyramid.rs:22-61
extern "C" void orc_pyramid_triangles(TriangleData out[6]) {
    V3 apex = v3(0.0f, 5.0f, 0.0f);
    float b = 4.0f;
    V3 p0 = v3(-b, 0, -b), p1 = v3(b, 0, -b), p2 = v3(b, 0, b), p3 = v3(-b, 0, b);
    make_tri(&out[0], p0, p1, apex, 0, 0, 1, 0, 0.5f, 1, 0);
    make_tri(&out[1], p1, p2, apex, 0, 0, 1, 0, 0.5f, 1, 1);
    make_tri(&out[2], p2, p3, apex, 0, 0, 1, 0, 0.5f, 1, 2);
    make_tri(&out[3], p3, p0, apex, 0, 0, 1, 0, 0.5f, 1, 3);
    make_tri(&out[4], p0, p2, p1, 0, 0, 0, 0, 0, 0, 4);
    make_tri(&out[5], p0, p3, p2, 0, 0, 0, 0, 0, 0, 4);
}
