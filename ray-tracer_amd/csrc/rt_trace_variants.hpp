// DEVELOPMENT ONLY (built with -DRT_EXPERIMENTS into librt_mi355_exp.so, never shipped):
// alternative shapes of the traversal loop, timed against each other in one process by tools/variants_bench.py.
#pragma once
#include "rt_trace.hpp"

namespace rt {


// VARIANT 1: exit detection by per-axis remaining-step counters instead of cell coordinates
template <bool ANY>
__device__ inline bool trace_ray_c(const DevScene& s, V3 o, const RayDir& rd, HitRec& out) {
    float closest = 1e10f;
    uint32_t best_id = MISS, best_ref = MISS;
    const V3 d = rd.d, inv = rd.inv;
    const V3 bmin = v3(s.bmin[0], s.bmin[1], s.bmin[2]), bmax = v3(s.bmax[0], s.bmax[1], s.bmax[2]);
    const float cs = s.cell_size;
    V3 pos = o;
    float t_off = 0.0f;
    if (pos.x < bmin.x || pos.x > bmax.x || pos.y < bmin.y || pos.y > bmax.y || pos.z < bmin.z || pos.z > bmax.z) {
        const float t_entry = slab(o, inv, bmin, bmax);
        if (t_entry < 0.0f) return false;
        t_off = t_entry + 0.001f;
        pos = o + d * t_off;
        const float e = 0.0001f;
        pos = vmin(vmax(pos, bmin + v3(e, e, e)), bmax - v3(e, e, e));
    }
    uint32_t cx, cy, cz;
    {
        const V3 rel = pos - bmin;
        const float ics = s.inv_cell_size;
        cx = f2u_sat(fmaxf(0.0f, floorf(rel.x * ics)));
        cy = f2u_sat(fmaxf(0.0f, floorf(rel.y * ics)));
        cz = f2u_sat(fmaxf(0.0f, floorf(rel.z * ics)));
    }
    const uint32_t sx = rd.sx, sy = rd.sy, sz = rd.sz, dix = rd.dix, diy = rd.diy, diz = rd.diz;
    const V3 cell_world = bmin + v3((float)cx, (float)cy, (float)cz) * cs;
    const V3 next_b = cell_world + v3(sx == 1u ? cs : 0.0f, sy == 1u ? cs : 0.0f, sz == 1u ? cs : 0.0f);
    const V3 t_delta = rd.t_delta;
    V3 t_max = v3(t_off, t_off, t_off) + (next_b - pos) * inv;
    {
        const float fl = t_off + 0.00001f;
        t_max = vmax(t_max, v3(fl, fl, fl));
    }
    uint32_t idx = cx + cy * s.gx + cz * s.gx * s.gy;
    // steps that can still be taken along each axis before the ray leaves the grid; a start cell outside the
    // grid (coordinate >= g) can take none in the + direction and "c" in the - direction, as the shader's
    // signed test (next < 0 || next >= g) decides
    uint32_t rx = sx == 1u ? (cx < s.gx ? s.gx - 1u - cx : 0u) : cx;
    uint32_t ry = sy == 1u ? (cy < s.gy ? s.gy - 1u - cy : 0u) : cy;
    uint32_t rz = sz == 1u ? (cz < s.gz ? s.gz - 1u - cz : 0u) : cz;
    // the - direction from a start cell beyond the grid: the shader steps to c-1 >= g and stops
    if (sx != 1u && cx > s.gx) rx = 0u;
    if (sy != 1u && cy > s.gy) ry = 0u;
    if (sz != 1u && cz > s.gz) rz = 0u;
    for (uint32_t i = 0; i < s.max_steps; ++i) {
        uint2 cell = make_uint2(0u, 0u);
        if (idx < s.n_cells) cell = s.cells[idx];
        const uint32_t n = cell.y < s.max_per_cell ? cell.y : s.max_per_cell;
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t r = cell.x + j;
            const float4 q0 = s.refs[3 * r], q1 = s.refs[3 * r + 1], q2 = s.refs[3 * r + 2];
            const uint32_t id = __float_as_uint(q0.w);
            const float t = moller_trumbore<false>(o, d, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z));
            if (t > 0.0f && t < closest && id != REF_INVALID) {
                closest = t;
                best_id = id;
                best_ref = r;
                if (ANY) break;
            }
        }
        if (ANY && best_id != MISS) break;
        if (best_id != MISS && closest < fminf(fminf(t_max.x, t_max.y), t_max.z)) break;
        const bool ax = t_max.x < t_max.y && t_max.x < t_max.z;
        const bool ay = !ax && t_max.y < t_max.z;
        const bool az = !ax && !ay;
        const uint32_t rem = ax ? rx : (ay ? ry : rz);
        if (rem == 0u) break;
        rx -= ax ? 1u : 0u;
        ry -= ay ? 1u : 0u;
        rz -= az ? 1u : 0u;
        idx += ax ? dix : (ay ? diy : diz);
        t_max.x += ax ? t_delta.x : 0.0f;
        t_max.y += ay ? t_delta.y : 0.0f;
        t_max.z += az ? t_delta.z : 0.0f;
    }
    out = HitRec{closest, best_id, best_ref};
    return best_id != MISS;
}


// VARIANT 2: VARIANT 1 plus per-wave trip counters (how many times a wave runs each loop body)
__device__ unsigned long long g_trip[8];
__device__ unsigned int g_wave_trips[4 * 65536];  // per wave: primary DDA, primary MT, shadow DDA, shadow MT trips
__device__ inline void wave_count(int k) {
    const unsigned long long m = __ballot(1);
    if ((threadIdx.x & 63) == (uint32_t)(__builtin_ffsll((long long)m) - 1)) {
        atomicAdd(&g_trip[k], 1ull);
        const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        const int slot = k == 2 ? 0 : (k == 3 ? 1 : (k == 6 ? 2 : (k == 7 ? 3 : -1)));
        if (slot >= 0 && gw < 65536) atomicAdd(&g_wave_trips[4 * gw + slot], 1u);
    }
}
template <bool ANY>
__device__ inline bool trace_ray_t(const DevScene& s, V3 o, const RayDir& rd, HitRec& out) {
    float closest = 1e10f;
    uint32_t best_id = MISS, best_ref = MISS;
    const V3 d = rd.d, inv = rd.inv;
    const V3 bmin = v3(s.bmin[0], s.bmin[1], s.bmin[2]), bmax = v3(s.bmax[0], s.bmax[1], s.bmax[2]);
    const float cs = s.cell_size;
    V3 pos = o;
    float t_off = 0.0f;
    wave_count(ANY ? 4 : 0);
    if (pos.x < bmin.x || pos.x > bmax.x || pos.y < bmin.y || pos.y > bmax.y || pos.z < bmin.z || pos.z > bmax.z) {
        const float t_entry = slab(o, inv, bmin, bmax);
        if (t_entry < 0.0f) return false;
        t_off = t_entry + 0.001f;
        pos = o + d * t_off;
        const float e = 0.0001f;
        pos = vmin(vmax(pos, bmin + v3(e, e, e)), bmax - v3(e, e, e));
    }
    wave_count(ANY ? 5 : 1);
    uint32_t cx, cy, cz;
    {
        const V3 rel = pos - bmin;
        const float ics = s.inv_cell_size;
        cx = f2u_sat(fmaxf(0.0f, floorf(rel.x * ics)));
        cy = f2u_sat(fmaxf(0.0f, floorf(rel.y * ics)));
        cz = f2u_sat(fmaxf(0.0f, floorf(rel.z * ics)));
    }
    const uint32_t sx = rd.sx, sy = rd.sy, sz = rd.sz, dix = rd.dix, diy = rd.diy, diz = rd.diz;
    const V3 cell_world = bmin + v3((float)cx, (float)cy, (float)cz) * cs;
    const V3 next_b = cell_world + v3(sx == 1u ? cs : 0.0f, sy == 1u ? cs : 0.0f, sz == 1u ? cs : 0.0f);
    const V3 t_delta = rd.t_delta;
    V3 t_max = v3(t_off, t_off, t_off) + (next_b - pos) * inv;
    {
        const float fl = t_off + 0.00001f;
        t_max = vmax(t_max, v3(fl, fl, fl));
    }
    uint32_t idx = cx + cy * s.gx + cz * s.gx * s.gy;
    uint32_t rx = sx == 1u ? (cx < s.gx ? s.gx - 1u - cx : 0u) : cx;
    uint32_t ry = sy == 1u ? (cy < s.gy ? s.gy - 1u - cy : 0u) : cy;
    uint32_t rz = sz == 1u ? (cz < s.gz ? s.gz - 1u - cz : 0u) : cz;
    if (sx != 1u && cx > s.gx) rx = 0u;
    if (sy != 1u && cy > s.gy) ry = 0u;
    if (sz != 1u && cz > s.gz) rz = 0u;
    for (uint32_t i = 0; i < s.max_steps; ++i) {
        wave_count(ANY ? 6 : 2);
        uint2 cell = make_uint2(0u, 0u);
        if (idx < s.n_cells) cell = s.cells[idx];
        const uint32_t n = cell.y < s.max_per_cell ? cell.y : s.max_per_cell;
        for (uint32_t j = 0; j < n; ++j) {
            wave_count(ANY ? 7 : 3);
            const uint32_t r = cell.x + j;
            const float4 q0 = s.refs[3 * r], q1 = s.refs[3 * r + 1], q2 = s.refs[3 * r + 2];
            const uint32_t id = __float_as_uint(q0.w);
            const float t = moller_trumbore<false>(o, d, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z));
            if (t > 0.0f && t < closest && id != REF_INVALID) {
                closest = t;
                best_id = id;
                best_ref = r;
                if (ANY) break;
            }
        }
        if (ANY && best_id != MISS) break;
        if (best_id != MISS && closest < fminf(fminf(t_max.x, t_max.y), t_max.z)) break;
        const bool ax = t_max.x < t_max.y && t_max.x < t_max.z;
        const bool ay = !ax && t_max.y < t_max.z;
        const bool az = !ax && !ay;
        const uint32_t rem = ax ? rx : (ay ? ry : rz);
        if (rem == 0u) break;
        rx -= ax ? 1u : 0u;
        ry -= ay ? 1u : 0u;
        rz -= az ? 1u : 0u;
        idx += ax ? dix : (ay ? diy : diz);
        t_max.x += ax ? t_delta.x : 0.0f;
        t_max.y += ay ? t_delta.y : 0.0f;
        t_max.z += az ? t_delta.z : 0.0f;
    }
    out = HitRec{closest, best_id, best_ref};
    return best_id != MISS;
}

// VARIANT 3: VARIANT 2 (instrumented) + skipping of re-tests by the per-reference masks
template <bool ANY>
__device__ inline bool trace_ray_m(const DevScene& s, V3 o, const RayDir& rd, HitRec& out) {
    float closest = 1e10f;
    uint32_t best_id = MISS, best_ref = MISS;
    const V3 d = rd.d, inv = rd.inv;
    const V3 bmin = v3(s.bmin[0], s.bmin[1], s.bmin[2]), bmax = v3(s.bmax[0], s.bmax[1], s.bmax[2]);
    const float cs = s.cell_size;
    V3 pos = o;
    float t_off = 0.0f;
    wave_count(ANY ? 4 : 0);
    if (pos.x < bmin.x || pos.x > bmax.x || pos.y < bmin.y || pos.y > bmax.y || pos.z < bmin.z || pos.z > bmax.z) {
        const float t_entry = slab(o, inv, bmin, bmax);
        if (t_entry < 0.0f) return false;
        t_off = t_entry + 0.001f;
        pos = o + d * t_off;
        const float e = 0.0001f;
        pos = vmin(vmax(pos, bmin + v3(e, e, e)), bmax - v3(e, e, e));
    }
    wave_count(ANY ? 5 : 1);
    uint32_t cx, cy, cz;
    {
        const V3 rel = pos - bmin;
        const float ics = s.inv_cell_size;
        cx = f2u_sat(fmaxf(0.0f, floorf(rel.x * ics)));
        cy = f2u_sat(fmaxf(0.0f, floorf(rel.y * ics)));
        cz = f2u_sat(fmaxf(0.0f, floorf(rel.z * ics)));
    }
    const uint32_t sx = rd.sx, sy = rd.sy, sz = rd.sz, dix = rd.dix, diy = rd.diy, diz = rd.diz;
    const V3 cell_world = bmin + v3((float)cx, (float)cy, (float)cz) * cs;
    const V3 next_b = cell_world + v3(sx == 1u ? cs : 0.0f, sy == 1u ? cs : 0.0f, sz == 1u ? cs : 0.0f);
    const V3 t_delta = rd.t_delta;
    V3 t_max = v3(t_off, t_off, t_off) + (next_b - pos) * inv;
    {
        const float fl = t_off + 0.00001f;
        t_max = vmax(t_max, v3(fl, fl, fl));
    }
    uint32_t idx = cx + cy * s.gx + cz * s.gx * s.gy;
    uint32_t rx = sx == 1u ? (cx < s.gx ? s.gx - 1u - cx : 0u) : cx;
    uint32_t ry = sy == 1u ? (cy < s.gy ? s.gy - 1u - cy : 0u) : cy;
    uint32_t rz = sz == 1u ? (cz < s.gz ? s.gz - 1u - cz : 0u) : cz;
    if (sx != 1u && cx > s.gx) rx = 0u;
    if (sy != 1u && cy > s.gy) ry = 0u;
    if (sz != 1u && cz > s.gz) rz = 0u;
    const uint32_t mx = sx == 1u ? 1u : 2u, my = sy == 1u ? 4u : 8u, mz = sz == 1u ? 16u : 32u;
    uint32_t arrival = 0;  // face through which the current cell was entered (0 = start cell)
    for (uint32_t i = 0; i < s.max_steps; ++i) {
        wave_count(ANY ? 6 : 2);
        uint2 cell = make_uint2(0u, 0u);
        if (idx < s.n_cells) cell = s.cells[idx];
        const uint32_t n = cell.y < s.max_per_cell ? cell.y : s.max_per_cell;
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t r = cell.x + j;
            const float4 q0 = s.refs[3 * r], q1 = s.refs[3 * r + 1], q2 = s.refs[3 * r + 2];
            const uint32_t id = __float_as_uint(q0.w);
            if (__float_as_uint(q1.w) & arrival) continue;  // tested one cell ago with the same outcome
            wave_count(ANY ? 7 : 3);
            const float t = moller_trumbore<false>(o, d, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z));
            if (t > 0.0f && t < closest && id != REF_INVALID) {
                closest = t;
                best_id = id;
                best_ref = r;
                if (ANY) break;
            }
        }
        if (ANY && best_id != MISS) break;
        if (best_id != MISS && closest < fminf(fminf(t_max.x, t_max.y), t_max.z)) break;
        const bool ax = t_max.x < t_max.y && t_max.x < t_max.z;
        const bool ay = !ax && t_max.y < t_max.z;
        const bool az = !ax && !ay;
        const uint32_t rem = ax ? rx : (ay ? ry : rz);
        if (rem == 0u) break;
        rx -= ax ? 1u : 0u;
        ry -= ay ? 1u : 0u;
        rz -= az ? 1u : 0u;
        idx += ax ? dix : (ay ? diy : diz);
        arrival = ax ? mx : (ay ? my : mz);
        t_max.x += ax ? t_delta.x : 0.0f;
        t_max.y += ay ? t_delta.y : 0.0f;
        t_max.z += az ? t_delta.z : 0.0f;
    }
    out = HitRec{closest, best_id, best_ref};
    return best_id != MISS;
}


// VARIANT 4: VARIANT 3 without the trip counters
template <bool ANY>
__device__ inline bool trace_ray_m2(const DevScene& s, V3 o, const RayDir& rd, HitRec& out) {
    float closest = 1e10f;
    uint32_t best_id = MISS, best_ref = MISS;
    const V3 d = rd.d, inv = rd.inv;
    const V3 bmin = v3(s.bmin[0], s.bmin[1], s.bmin[2]), bmax = v3(s.bmax[0], s.bmax[1], s.bmax[2]);
    const float cs = s.cell_size;
    V3 pos = o;
    float t_off = 0.0f;
    if (pos.x < bmin.x || pos.x > bmax.x || pos.y < bmin.y || pos.y > bmax.y || pos.z < bmin.z || pos.z > bmax.z) {
        const float t_entry = slab(o, inv, bmin, bmax);
        if (t_entry < 0.0f) return false;
        t_off = t_entry + 0.001f;
        pos = o + d * t_off;
        const float e = 0.0001f;
        pos = vmin(vmax(pos, bmin + v3(e, e, e)), bmax - v3(e, e, e));
    }
    uint32_t cx, cy, cz;
    {
        const V3 rel = pos - bmin;
        const float ics = s.inv_cell_size;
        cx = f2u_sat(fmaxf(0.0f, floorf(rel.x * ics)));
        cy = f2u_sat(fmaxf(0.0f, floorf(rel.y * ics)));
        cz = f2u_sat(fmaxf(0.0f, floorf(rel.z * ics)));
    }
    const uint32_t sx = rd.sx, sy = rd.sy, sz = rd.sz, dix = rd.dix, diy = rd.diy, diz = rd.diz;
    const V3 cell_world = bmin + v3((float)cx, (float)cy, (float)cz) * cs;
    const V3 next_b = cell_world + v3(sx == 1u ? cs : 0.0f, sy == 1u ? cs : 0.0f, sz == 1u ? cs : 0.0f);
    const V3 t_delta = rd.t_delta;
    V3 t_max = v3(t_off, t_off, t_off) + (next_b - pos) * inv;
    {
        const float fl = t_off + 0.00001f;
        t_max = vmax(t_max, v3(fl, fl, fl));
    }
    uint32_t idx = cx + cy * s.gx + cz * s.gx * s.gy;
    uint32_t rx = sx == 1u ? (cx < s.gx ? s.gx - 1u - cx : 0u) : cx;
    uint32_t ry = sy == 1u ? (cy < s.gy ? s.gy - 1u - cy : 0u) : cy;
    uint32_t rz = sz == 1u ? (cz < s.gz ? s.gz - 1u - cz : 0u) : cz;
    if (sx != 1u && cx > s.gx) rx = 0u;
    if (sy != 1u && cy > s.gy) ry = 0u;
    if (sz != 1u && cz > s.gz) rz = 0u;
    const uint32_t mx = sx == 1u ? 1u : 2u, my = sy == 1u ? 4u : 8u, mz = sz == 1u ? 16u : 32u;
    uint32_t arrival = 0;  // face through which the current cell was entered (0 = start cell)
    for (uint32_t i = 0; i < s.max_steps; ++i) {
        uint2 cell = make_uint2(0u, 0u);
        if (idx < s.n_cells) cell = s.cells[idx];
        const uint32_t n = cell.y < s.max_per_cell ? cell.y : s.max_per_cell;
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t r = cell.x + j;
            const float4 q0 = s.refs[3 * r], q1 = s.refs[3 * r + 1], q2 = s.refs[3 * r + 2];
            const uint32_t id = __float_as_uint(q0.w);
            if (__float_as_uint(q1.w) & arrival) continue;  // tested one cell ago with the same outcome
            const float t = moller_trumbore<false>(o, d, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z));
            if (t > 0.0f && t < closest && id != REF_INVALID) {
                closest = t;
                best_id = id;
                best_ref = r;
                if (ANY) break;
            }
        }
        if (ANY && best_id != MISS) break;
        if (best_id != MISS && closest < fminf(fminf(t_max.x, t_max.y), t_max.z)) break;
        const bool ax = t_max.x < t_max.y && t_max.x < t_max.z;
        const bool ay = !ax && t_max.y < t_max.z;
        const bool az = !ax && !ay;
        const uint32_t rem = ax ? rx : (ay ? ry : rz);
        if (rem == 0u) break;
        rx -= ax ? 1u : 0u;
        ry -= ay ? 1u : 0u;
        rz -= az ? 1u : 0u;
        idx += ax ? dix : (ay ? diy : diz);
        arrival = ax ? mx : (ay ? my : mz);
        t_max.x += ax ? t_delta.x : 0.0f;
        t_max.y += ay ? t_delta.y : 0.0f;
        t_max.z += az ? t_delta.z : 0.0f;
    }
    out = HitRec{closest, best_id, best_ref};
    return best_id != MISS;
}



// VARIANT 7 (P=1), 8 (P=3), 9 (P=7): counter-based exit + re-test skipping, with
//   P&1: the next candidate's record is loaded while the current one is tested,
//   P&2: the next cell's record is loaded before the current cell's candidates are tested,
//   P&4: a wave that has run many candidate tests raises its issue priority.
template <int P, bool ANY>
__device__ inline bool trace_ray_p(const DevScene& s, V3 o, const RayDir& rd, HitRec& out) {
    constexpr bool PF_REF = P & 1, PF_CELL = P & 2, PRIO = P & 4;
    float closest = 1e10f;
    uint32_t best_id = MISS, best_ref = MISS;
    const V3 d = rd.d, inv = rd.inv;
    const V3 bmin = v3(s.bmin[0], s.bmin[1], s.bmin[2]), bmax = v3(s.bmax[0], s.bmax[1], s.bmax[2]);
    const float cs = s.cell_size;
    V3 pos = o;
    float t_off = 0.0f;
    if (pos.x < bmin.x || pos.x > bmax.x || pos.y < bmin.y || pos.y > bmax.y || pos.z < bmin.z || pos.z > bmax.z) {
        const float t_entry = slab(o, inv, bmin, bmax);
        if (t_entry < 0.0f) return false;
        t_off = t_entry + 0.001f;
        pos = o + d * t_off;
        const float e = 0.0001f;
        pos = vmin(vmax(pos, bmin + v3(e, e, e)), bmax - v3(e, e, e));
    }
    uint32_t cx, cy, cz;
    {
        const V3 rel = pos - bmin;
        const float ics = s.inv_cell_size;
        cx = f2u_sat(fmaxf(0.0f, floorf(rel.x * ics)));
        cy = f2u_sat(fmaxf(0.0f, floorf(rel.y * ics)));
        cz = f2u_sat(fmaxf(0.0f, floorf(rel.z * ics)));
    }
    const uint32_t sx = rd.sx, sy = rd.sy, sz = rd.sz, dix = rd.dix, diy = rd.diy, diz = rd.diz;
    const V3 cell_world = bmin + v3((float)cx, (float)cy, (float)cz) * cs;
    const V3 next_b = cell_world + v3(sx == 1u ? cs : 0.0f, sy == 1u ? cs : 0.0f, sz == 1u ? cs : 0.0f);
    const V3 t_delta = rd.t_delta;
    V3 t_max = v3(t_off, t_off, t_off) + (next_b - pos) * inv;
    {
        const float fl = t_off + 0.00001f;
        t_max = vmax(t_max, v3(fl, fl, fl));
    }
    uint32_t idx = cx + cy * s.gx + cz * s.gx * s.gy;
    uint32_t rx = sx == 1u ? (cx < s.gx ? s.gx - 1u - cx : 0u) : cx;
    uint32_t ry = sy == 1u ? (cy < s.gy ? s.gy - 1u - cy : 0u) : cy;
    uint32_t rz = sz == 1u ? (cz < s.gz ? s.gz - 1u - cz : 0u) : cz;
    if (sx != 1u && cx > s.gx) rx = 0u;
    if (sy != 1u && cy > s.gy) ry = 0u;
    if (sz != 1u && cz > s.gz) rz = 0u;
    const uint32_t mx = sx == 1u ? 1u : 2u, my = sy == 1u ? 4u : 8u, mz = sz == 1u ? 16u : 32u;
    uint32_t arrival = 0;
    uint32_t work = 0;
    uint2 cell = make_uint2(0u, 0u);
    if (idx < s.n_cells) cell = s.cells[idx];
    for (uint32_t i = 0; i < s.max_steps; ++i) {
        // the step this iteration will take depends on t_max only: decide it now and fetch the next cell's
        // record while this cell's candidates are tested
        const bool ax = t_max.x < t_max.y && t_max.x < t_max.z;
        const bool ay = !ax && t_max.y < t_max.z;
        const bool az = !ax && !ay;
        const uint32_t rem = ax ? rx : (ay ? ry : rz);
        const uint32_t idx_next = idx + (ax ? dix : (ay ? diy : diz));
        uint2 cell_next = make_uint2(0u, 0u);
        if (PF_CELL) {
            if (rem != 0u && idx_next < s.n_cells) cell_next = s.cells[idx_next];
        }
        const uint32_t n = cell.y < s.max_per_cell ? cell.y : s.max_per_cell;
        if (n) {
            uint32_t r = cell.x;
            float4 q0 = s.refs[3 * r], q1 = s.refs[3 * r + 1], q2 = s.refs[3 * r + 2];
            for (uint32_t j = 0; j < n; ++j, ++r) {
                const float4 c0 = q0, c1 = q1, c2 = q2;
                if (PF_REF) {  // one record past the list is still inside the (padded) array
                    q0 = s.refs[3 * r + 3], q1 = s.refs[3 * r + 4], q2 = s.refs[3 * r + 5];
                }
                const uint32_t id = __float_as_uint(c0.w);
                if (!(__float_as_uint(c1.w) & arrival)) {
                    const float t = moller_trumbore<false>(o, d, v3(c0.x, c0.y, c0.z), v3(c1.x, c1.y, c1.z), v3(c2.x, c2.y, c2.z));
                    if (t > 0.0f && t < closest && id != REF_INVALID) {
                        closest = t;
                        best_id = id;
                        best_ref = r;
                        if (ANY) break;
                    }
                    if (PRIO) ++work;
                }
                if (!PF_REF) {
                    q0 = s.refs[3 * r + 3], q1 = s.refs[3 * r + 4], q2 = s.refs[3 * r + 5];
                }
            }
            if (PRIO) {
                if (__builtin_amdgcn_readfirstlane(work) > 64u) __builtin_amdgcn_s_setprio(3);
            }
        }
        if (ANY && best_id != MISS) break;
        if (best_id != MISS && closest < fminf(fminf(t_max.x, t_max.y), t_max.z)) break;
        if (rem == 0u) break;
        rx -= ax ? 1u : 0u;
        ry -= ay ? 1u : 0u;
        rz -= az ? 1u : 0u;
        idx = idx_next;
        arrival = ax ? mx : (ay ? my : mz);
        t_max.x += ax ? t_delta.x : 0.0f;
        t_max.y += ay ? t_delta.y : 0.0f;
        t_max.z += az ? t_delta.z : 0.0f;
        if (PF_CELL) {
            cell = cell_next;
        } else {
            cell = make_uint2(0u, 0u);
            if (idx < s.n_cells) cell = s.cells[idx];
        }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    out = HitRec{closest, best_id, best_ref};
    return best_id != MISS;
}

// VARIANT 8: as 7, with the candidate loop unrolled by two (no register rotation between the record being
// tested and the one in flight), one running record pointer with immediate offsets, and a DDA tail without
// short-circuit control flow.
template <bool ANY>
__device__ inline bool trace_ray_u(const DevScene& s, V3 o, const RayDir& rd, HitRec& out) {
    float closest = 1e10f;
    uint32_t best_id = MISS, best_ref = MISS;
    const V3 d = rd.d, inv = rd.inv;
    const V3 bmin = v3(s.bmin[0], s.bmin[1], s.bmin[2]), bmax = v3(s.bmax[0], s.bmax[1], s.bmax[2]);
    const float cs = s.cell_size;
    V3 pos = o;
    float t_off = 0.0f;
    if (pos.x < bmin.x || pos.x > bmax.x || pos.y < bmin.y || pos.y > bmax.y || pos.z < bmin.z || pos.z > bmax.z) {
        const float t_entry = slab(o, inv, bmin, bmax);
        if (t_entry < 0.0f) return false;
        t_off = t_entry + 0.001f;
        pos = o + d * t_off;
        const float e = 0.0001f;
        pos = vmin(vmax(pos, bmin + v3(e, e, e)), bmax - v3(e, e, e));
    }
    uint32_t cx, cy, cz;
    {
        const V3 rel = pos - bmin;
        const float ics = s.inv_cell_size;
        cx = f2u_sat(fmaxf(0.0f, floorf(rel.x * ics)));
        cy = f2u_sat(fmaxf(0.0f, floorf(rel.y * ics)));
        cz = f2u_sat(fmaxf(0.0f, floorf(rel.z * ics)));
    }
    const uint32_t sx = rd.sx, sy = rd.sy, sz = rd.sz, dix = rd.dix, diy = rd.diy, diz = rd.diz;
    const V3 cell_world = bmin + v3((float)cx, (float)cy, (float)cz) * cs;
    const V3 next_b = cell_world + v3(sx == 1u ? cs : 0.0f, sy == 1u ? cs : 0.0f, sz == 1u ? cs : 0.0f);
    const V3 t_delta = rd.t_delta;
    V3 t_max = v3(t_off, t_off, t_off) + (next_b - pos) * inv;
    {
        const float fl = t_off + 0.00001f;
        t_max = vmax(t_max, v3(fl, fl, fl));
    }
    uint32_t idx = cx + cy * s.gx + cz * s.gx * s.gy;
    uint32_t rx = sx == 1u ? (cx < s.gx ? s.gx - 1u - cx : 0u) : (cx > s.gx ? 0u : cx);
    uint32_t ry = sy == 1u ? (cy < s.gy ? s.gy - 1u - cy : 0u) : (cy > s.gy ? 0u : cy);
    uint32_t rz = sz == 1u ? (cz < s.gz ? s.gz - 1u - cz : 0u) : (cz > s.gz ? 0u : cz);
    const uint32_t mx = sx == 1u ? 1u : 2u, my = sy == 1u ? 4u : 8u, mz = sz == 1u ? 16u : 32u;
    uint32_t arrival = 0;
    uint2 cell = make_uint2(0u, 0u);
    if (idx < s.n_cells) cell = s.cells[idx];
    // tests one record; true = stop scanning (any-hit found)
    auto test = [&](const float4& c0, const float4& c1, const float4& c2, uint32_t r) -> bool {
        if (!(__float_as_uint(c1.w) & arrival)) {
            const float t = moller_trumbore<false>(o, d, v3(c0.x, c0.y, c0.z), v3(c1.x, c1.y, c1.z), v3(c2.x, c2.y, c2.z));
            const uint32_t id = __float_as_uint(c0.w);
            if (t > 0.0f && t < closest && id != REF_INVALID) {
                closest = t;
                best_id = id;
                best_ref = r;
                return ANY;
            }
        }
        return false;
    };
    for (uint32_t i = 0; i < s.max_steps; ++i) {
        const bool ax = t_max.x < t_max.y && t_max.x < t_max.z;
        const bool ay = !ax && t_max.y < t_max.z;
        const bool az = !ax && !ay;
        const uint32_t rem = ax ? rx : (ay ? ry : rz);
        const uint32_t idx_next = idx + (ax ? dix : (ay ? diy : diz));
        const float t_exit = fminf(fminf(t_max.x, t_max.y), t_max.z);
        const uint32_t n = cell.y < s.max_per_cell ? cell.y : s.max_per_cell;
        if (n) {
            uint32_t r = cell.x;
            const uint32_t r_end = r + n;
            const float4* p = s.refs + 3 * (size_t)r;
            float4 a0 = p[0], a1 = p[1], a2 = p[2];
            for (;;) {
                const float4 b0 = p[3], b1 = p[4], b2 = p[5];  // one past the list is inside the padded array
                if (test(a0, a1, a2, r)) break;
                if (++r == r_end) break;
                a0 = p[6], a1 = p[7], a2 = p[8];
                p += 6;
                if (test(b0, b1, b2, r)) break;
                if (++r == r_end) break;
            }
        }
        const bool hit = best_id != MISS;
        const bool stop = (ANY & hit) | (hit & (closest < t_exit)) | (rem == 0u);
        if (stop) break;
        rx -= ax ? 1u : 0u;
        ry -= ay ? 1u : 0u;
        rz -= az ? 1u : 0u;
        idx = idx_next;
        arrival = ax ? mx : (ay ? my : mz);
        t_max.x += ax ? t_delta.x : 0.0f;
        t_max.y += ay ? t_delta.y : 0.0f;
        t_max.z += az ? t_delta.z : 0.0f;
        cell = make_uint2(0u, 0u);
        if (idx < s.n_cells) cell = s.cells[idx];
    }
    out = HitRec{closest, best_id, best_ref};
    return best_id != MISS;
}

// VARIANT 0: the shipped trace_ray
// VARIANT 5 / 6: body 2 / 3 (trip counters, without / with re-test skipping) + per-wave start/end timestamps (s_memrealtime, 100 MHz) into g_wave_t
__device__ unsigned long long g_wave_t[2 * 65536];

// VARIANT 10+v: body v with one wave per workgroup; 20+v: body v with __launch_bounds__(256, 8);
// 30+v: body v with per-wave timestamps
template <int VARIANT>
struct VarCfg {
    static constexpr int body = VARIANT % 10 + (VARIANT >= 40 ? 10 : 0);
    static constexpr bool stamps = VARIANT >= 30 && VARIANT < 40;
    static constexpr int threads = (VARIANT >= 10 && VARIANT < 20) ? 64 : 256;
    static constexpr int min_waves = VARIANT >= 20 ? 8 : 1;
};
template <int VARIANT>
__global__ __launch_bounds__(VarCfg<VARIANT>::threads, VarCfg<VARIANT>::min_waves) void k_trace_primary_var(
    DevScene s, DevFrame f, ShardMap m, uint2* __restrict__ out_p, uint2* __restrict__ out_s) {
    constexpr int BODY = VarCfg<VARIANT>::body == 5 ? 2 : (VarCfg<VARIANT>::body == 6 ? 3 : VarCfg<VARIANT>::body);
    constexpr bool STAMPS = VarCfg<VARIANT>::stamps || VARIANT == 5 || VARIANT == 6;
    const unsigned long long t_start = STAMPS ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t gw = VarCfg<VARIANT>::threads == 64 ? blockIdx.x : blockIdx.x * 4 + (threadIdx.x >> 6);  // global wave
    const uint32_t w4 = gw & 3, b16 = gw >> 2;
    const uint32_t blk_x = (m.sub_x + 1) / 2, blk_y = (m.sub_y + 1) / 2;
    const uint32_t blk_per_tile = blk_x * blk_y;
    const uint32_t lt = b16 / blk_per_tile, blk = b16 % blk_per_tile;
    if (lt >= m.n_local) return;
    const uint32_t sub_x = (blk % blk_x) * 2 + (w4 & 1), sub_y = (blk / blk_x) * 2 + (w4 >> 1);
    const uint32_t lx = sub_x * 8 + (lane & 7), ly = sub_y * 8 + (lane >> 3);
    const uint32_t tile = m.first + lt * m.stride;
    const uint32_t px = (tile % m.tiles_x) * m.tile_w + lx, py = (tile / m.tiles_x) * m.tile_h + ly;
    const bool in_tile = lx < m.tile_w && ly < m.tile_h;
    const bool live = in_tile && px < f.width && py < f.height;
    uint2 ph = make_uint2(MISS, 0u), sh = make_uint2(MISS, 0u);
    Counters ctr{0, 0, 0, 0};
    auto trace = [&](V3 o, const RayDir& rd, HitRec& h, auto any) -> bool {
        constexpr bool ANY = decltype(any)::value;
        if constexpr (BODY == 1) return trace_ray_c<ANY>(s, o, rd, h);
        else if constexpr (BODY == 2) return trace_ray_t<ANY>(s, o, rd, h);
        else if constexpr (BODY == 7) return trace_ray_p<1, ANY>(s, o, rd, h);
        else if constexpr (BODY == 8) return trace_ray_u<ANY>(s, o, rd, h);
        else if constexpr (BODY == 9) return trace_ray_p<4, ANY>(s, o, rd, h);
        else if constexpr (BODY == 3) return trace_ray_m<ANY>(s, o, rd, h);
        else if constexpr (BODY == 4) return trace_ray_m2<ANY>(s, o, rd, h);
        else return trace_ray<false, ANY, false, false>(s, o, rd, h, ctr);
    };
    if (live) {
        const V3 o = v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
        const V3 d = primary_dir(f, px, py);
        HitRec h;
        const RayDir rd = make_ray_dir<false>(d, s.cell_size, s.gx, s.gy);
        if (trace(o, rd, h, std::false_type{})) {
            ph = make_uint2(h.id, __float_as_uint(h.t));
            const V3 pos = o + d * h.t;
            V3 n = hit_normal(s, h, pos);
            if (dot(n, d) > 0.0f) n = neg(n);
            const V3 so = pos + n * f.shadow_offset;
            HitRec oh;
            if (trace(so, f.shadow, oh, std::true_type{})) sh = make_uint2(oh.id, __float_as_uint(oh.t));
        }
    }
    if (in_tile) {
        const size_t oi = (size_t)lt * m.tile_w * m.tile_h + (size_t)ly * m.tile_w + lx;
        out_p[oi] = ph;
        out_s[oi] = sh;
    }
    if (STAMPS && lane == 0 && gw < 65536) {
        g_wave_t[2 * gw] = t_start;
        g_wave_t[2 * gw + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace rt
