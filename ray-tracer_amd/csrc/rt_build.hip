// On-GPU grid binning: the device counterpart of HierarchicalGrid::build (src/grid.rs:110-216)
// and TriangleGrid::build (src/grid_triangles.rs:29-124). Same bounds, same dimensions, same
// per-object cell ranges, ids in ascending order inside every cell, inserts beyond the per-cell
// cap dropped (i.e. the lowest `cap` ids survive). The three coarse count levels of the
// reference are not produced: its shader never reads them (raytracer_unified.wgsl:134 is the
// only mention).
//
// Passes: bounds reduction -> (object, cell) pairs per object + scan -> per-cell counts (one lane per pair,
// atomics) -> exclusive scan -> fill (one lane per pair, atomics, arbitrary order) -> per-cell sort by id ->
// cap + compaction -> 48-byte reference records -> re-test masks.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <cmath>

#include "rt_internal.hpp"

using namespace rt;

namespace {

struct BinGrid {
    float bmin[3];
    float cell;
    uint32_t gx, gy, gz;
    uint32_t nb, nt;
    int tri_mode;
};

// Bounds of object `id` (boxes first, then triangles): src/types.rs:110-115, 235-244.
__device__ inline void object_bounds(const BoxData* boxes, const float* tri_v, uint32_t nb, uint32_t id, V3& mn, V3& mx) {
    if (id < nb) {
        mn = ld3(boxes[id].min);
        mx = ld3(boxes[id].max);
    } else {
        const float* v = tri_v + 9 * (size_t)(id - nb);
        V3 a = ld3(v), b = ld3(v + 3), c = ld3(v + 6);
        mn = vmin(vmin(a, b), c);
        mx = vmax(vmax(a, b), c);
    }
}

__global__ __launch_bounds__(256) void k_bounds_partial(const BoxData* __restrict__ boxes, const float* __restrict__ tri_v,
                                                        uint32_t nb, uint32_t total, float* __restrict__ partial) {
    __shared__ float red[6][256];
    V3 mn = v3(INFINITY, INFINITY, INFINITY), mx = v3(-INFINITY, -INFINITY, -INFINITY);
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < total; id += gridDim.x * blockDim.x) {
        V3 a, b;
        object_bounds(boxes, tri_v, nb, id, a, b);
        mn = vmin(mn, a);
        mx = vmax(mx, b);
    }
    const uint32_t t = threadIdx.x;
    red[0][t] = mn.x, red[1][t] = mn.y, red[2][t] = mn.z;
    red[3][t] = mx.x, red[4][t] = mx.y, red[5][t] = mx.z;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (t < s) {
            for (int k = 0; k < 3; ++k) red[k][t] = fminf(red[k][t], red[k][t + s]);
            for (int k = 3; k < 6; ++k) red[k][t] = fmaxf(red[k][t], red[k][t + s]);
        }
        __syncthreads();
    }
    if (t < 6) partial[blockIdx.x * 6 + t] = red[t][0];
}

__global__ void k_bounds_final(const float* __restrict__ partial, uint32_t n_blocks, float* __restrict__ out6) {
    const uint32_t k = threadIdx.x;
    if (k >= 6) return;
    float v = partial[k];
    for (uint32_t b = 1; b < n_blocks; ++b) v = k < 3 ? fminf(v, partial[b * 6 + k]) : fmaxf(v, partial[b * 6 + k]);
    out6[k] = v;
}

struct Range3 {
    int lo[3], hi[3];  // inclusive, empty when lo > hi
};

// src/grid.rs:218-237,292-299 (clamp at 0, `as u32`, cells outside the grid skipped) and
// src/grid_triangles.rs:171-209 (`as i32`, loops clamped to the grid).
__device__ inline Range3 cell_range(const BinGrid& g, V3 omin, V3 omax) {
    const V3 bmin = v3(g.bmin[0], g.bmin[1], g.bmin[2]);
    const V3 rmin = omin - bmin, rmax = omax - bmin;
    const float lo_f[3] = {floorf(rmin.x / g.cell), floorf(rmin.y / g.cell), floorf(rmin.z / g.cell)};
    const float hi_f[3] = {floorf(rmax.x / g.cell), floorf(rmax.y / g.cell), floorf(rmax.z / g.cell)};
    const uint32_t dims[3] = {g.gx, g.gy, g.gz};
    Range3 r;
    for (int k = 0; k < 3; ++k) {
        long long lo, hi;
        if (g.tri_mode) {
            lo = f2i_sat(lo_f[k]);
            hi = f2i_sat(hi_f[k]);
            if (lo < 0) lo = 0;
        } else {
            lo = f2u_sat(fmaxf(lo_f[k], 0.0f));
            hi = f2u_sat(fmaxf(hi_f[k], 0.0f));
        }
        if (hi > (long long)dims[k] - 1) hi = (long long)dims[k] - 1;
        if (lo > hi) {  // empty: normalise so the loops below do nothing
            lo = 1;
            hi = 0;
        }
        r.lo[k] = (int)lo;
        r.hi[k] = (int)hi;
    }
    return r;
}

// Number of (object, cell) pairs each object contributes = volume of its cell range.
__global__ __launch_bounds__(256) void k_obj_pairs(BinGrid g, const BoxData* __restrict__ boxes, const float* __restrict__ tri_v,
                                                   uint32_t* __restrict__ obj_n) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = g.nb + g.nt;
    if (id > total) return;
    uint32_t n = 0;
    if (id < total) {
        V3 mn, mx;
        object_bounds(boxes, tri_v, g.nb, id, mn, mx);
        const Range3 r = cell_range(g, mn, mx);
        // saturate: the host rejects totals beyond its limit, a wrapped product would slip through
        const unsigned long long v = (unsigned long long)(r.hi[0] - r.lo[0] + 1) * (unsigned long long)(r.hi[1] - r.lo[1] + 1) *
                                     (unsigned long long)(r.hi[2] - r.lo[2] + 1);
        n = r.lo[0] > r.hi[0] ? 0u : (v > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)v);
    }
    obj_n[id] = n;  // entry `total` is the scan's closing element
}

// One lane per (object, cell) pair, so an object that spans thousands of cells (a ground plane) is spread over
// thousands of lanes instead of one lane's loop. obj_off = exclusive scan of the per-object pair counts.
// COUNT pass: slot = counts[cell]++, kept per pair.  FILL pass: the id goes to the cell's segment at that slot
// (order inside a cell is arbitrary here, the per-cell sort that follows makes it ascending).
template <bool FILL>
__global__ __launch_bounds__(256) void k_bin(BinGrid g, const BoxData* __restrict__ boxes, const float* __restrict__ tri_v,
                                             const uint32_t* __restrict__ obj_off, uint32_t n_pairs,
                                             uint32_t* __restrict__ counts, uint32_t* __restrict__ slots,
                                             const uint32_t* __restrict__ tmp_start, uint32_t* __restrict__ tmp_ids,
                                             uint32_t* __restrict__ tmp_cell) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    // object of a pair: last id with obj_off[id] <= p (objects without cells share their successor's offset).
    // Two lanes search the whole table for the workgroup's first and last pair; everyone else searches between
    // those two objects only (a few steps, from cache, instead of log2(objects) dependent loads per pair).
    __shared__ uint32_t window[2];
    if (threadIdx.x < 2) {
        const uint32_t last = n_pairs - 1;
        const uint32_t q = threadIdx.x == 0 ? blockIdx.x * blockDim.x : min(blockIdx.x * blockDim.x + blockDim.x - 1, last);
        uint32_t lo = 0, hi = g.nb + g.nt;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) / 2;
            if (obj_off[mid] <= q) lo = mid; else hi = mid;
        }
        window[threadIdx.x] = lo;
    }
    __syncthreads();
    if (p >= n_pairs) return;
    uint32_t lo = window[0], hi = window[1] + 1;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (obj_off[mid] <= p) lo = mid; else hi = mid;
    }
    const uint32_t id = lo;
    V3 mn, mx;
    object_bounds(boxes, tri_v, g.nb, id, mn, mx);
    const Range3 r = cell_range(g, mn, mx);
    const uint32_t nx = (uint32_t)(r.hi[0] - r.lo[0] + 1), ny = (uint32_t)(r.hi[1] - r.lo[1] + 1);
    const uint32_t j = p - obj_off[id];
    const uint32_t x = (uint32_t)r.lo[0] + j % nx, y = (uint32_t)r.lo[1] + (j / nx) % ny, z = (uint32_t)r.lo[2] + j / (nx * ny);
    const uint32_t c = x + y * g.gx + z * g.gx * g.gy;
    if (!FILL) {
        slots[p] = atomicAdd(&counts[c], 1u);  // the pair's place inside its cell: the fill pass needs no atomics
    } else {
        const uint32_t dst = tmp_start[c] + slots[p];
        tmp_ids[dst] = id;
        tmp_cell[dst] = c;
    }
}

// One 48-byte record per reference, in cell order: geometry, id word, material word, re-test mask.
//
// Id and material words (raytracer_unified.wgsl:291-316): a triangle whose material discards all of its hits
// (constant alpha below the MASK cutoff / the BLEND threshold, no base-colour texture) keeps its record - the shader
// tests it - but can never be accepted: REF_INVISIBLE. A triangle whose alpha comes from a base-colour texture carries
// material + 1 in the spare word of its third quad: the trace evaluates the rule per hit.
//
// Re-test masks (grids built here: ref_cell != nullptr). A reference (cell C, object T) gets bit 2a (2a+1) set when
// the cell one step back along axis a in the - (+) direction also lists T and lists it within the part of its list
// that is traversed. A ray that ENTERS C through that face has therefore just tested T against the same ray: testing
// it again would give the same t, which can neither beat the current closest hit (strict <) nor turn a rejection into
// a hit, so the trace kernel skips it. T's cells form a box and a line crosses a box in one run of face-adjacent
// cells, so "listed one step back" is the same as "already tested". Cells whose lists may have been cut by the
// per-cell cap never vouch for their neighbours (their bits stay 0), which keeps the skip exact. full_start: the
// reference's lists (whether a cell was cut by the cap is a property of those, not of lists with copies left out).
__device__ inline void emit_record(const BinGrid& g, const BoxData* __restrict__ boxes, const float* __restrict__ tri_v,
                                   const uint32_t* __restrict__ indices, const uint32_t* __restrict__ ref_cell,
                                   const uint32_t* __restrict__ full_start, uint32_t cap, const uint32_t* __restrict__ tri_mat,
                                   const float4* __restrict__ materials, uint32_t n_materials, float4* __restrict__ refs,
                                   uint32_t r) {
    const uint32_t id = indices[r];
    uint32_t word = id, mat_word = 0, mask = 0;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
    V3 mn = v3(0.f, 0.f, 0.f), mx = mn;
    bool known = true;
    if (id >= g.nb) {
        const uint32_t ti = id - g.nb;
        if (ti < g.nt) {
            const float* v = tri_v + 9 * (size_t)ti;
            const V3 v0 = ld3(v), v1 = ld3(v + 3), v2 = ld3(v + 6);
            const V3 e1 = v1 - v0, e2 = v2 - v0;
            q0 = make_float4(v0.x, v0.y, v0.z, 0.f);
            q1 = make_float4(e1.x, e1.y, e1.z, 0.f);
            q2 = make_float4(e2.x, e2.y, e2.z, 0.f);
            mn = vmin(vmin(v0, v1), v2);
            mx = vmax(vmax(v0, v1), v2);
            const uint32_t mat = tri_mat[ti];
            if (mat < n_materials) {
                const float4 m0 = materials[4 * mat], m1 = materials[4 * mat + 1], m3 = materials[4 * mat + 3];
                const int texture = (int)__float_as_uint(m1.w);
                const uint32_t mode = __float_as_uint(m3.x);
                if (texture < 0) {
                    if ((mode == 1u && m0.w < m3.y) || (mode == 2u && m0.w < 0.01f)) word = REF_INVISIBLE;
                } else if (mode == 1u || mode == 2u) {
                    mat_word = mat + 1u;
                }
            }
        } else {
            word = REF_INVALID;
            known = false;
        }
    } else {
        mn = ld3(boxes[id].min);
        mx = ld3(boxes[id].max);
    }
    if (ref_cell && known) {
        const uint32_t c = ref_cell[r];
        const uint32_t cc[3] = {c % g.gx, (c / g.gx) % g.gy, c / (g.gx * g.gy)};
        const uint32_t stride[3] = {1u, g.gx, g.gx * g.gy};
        const Range3 rg = cell_range(g, mn, mx);
        for (int a = 0; a < 3; ++a) {
            if ((int)cc[a] > rg.lo[a]) {  // arrived by a + step: previous cell is c - stride
                const uint32_t p = c - stride[a];
                if (full_start[p + 1] - full_start[p] < cap) mask |= 1u << (2 * a);
            }
            if ((int)cc[a] < rg.hi[a]) {  // arrived by a - step: previous cell is c + stride
                const uint32_t p = c + stride[a];
                if (full_start[p + 1] - full_start[p] < cap) mask |= 2u << (2 * a);
            }
        }
    }
    q0.w = __uint_as_float(word);
    q1.w = __uint_as_float(mask);
    q2.w = __uint_as_float(mat_word);
    refs[3 * (size_t)r] = q0;
    refs[3 * (size_t)r + 1] = q1;
    refs[3 * (size_t)r + 2] = q2;
}

__global__ void k_emit_records(BinGrid g, const BoxData* __restrict__ boxes, const float* __restrict__ tri_v,
                               const uint32_t* __restrict__ indices, const uint32_t* __restrict__ ref_cell,
                               const uint32_t* __restrict__ full_start, uint32_t n_refs, uint32_t cap,
                               const uint32_t* __restrict__ tri_mat, const float4* __restrict__ materials, uint32_t n_materials,
                               float4* __restrict__ refs) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_refs) return;
    emit_record(g, boxes, tri_v, indices, ref_cell, full_start, cap, tri_mat, materials, n_materials, refs, r);
}

__global__ void k_clamp_counts(const uint32_t* __restrict__ counts, uint32_t n_cells, uint32_t cap,
                               uint32_t* __restrict__ clamped) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_cells) clamped[c] = counts[c] < cap ? counts[c] : cap;
    if (c == n_cells) clamped[c] = 0;
}

__global__ void k_compact(const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ tmp_cell,
                          const uint32_t* __restrict__ tmp_start, const uint32_t* __restrict__ final_start, uint32_t n_tmp,
                          uint32_t cap, uint32_t* __restrict__ indices, uint32_t* __restrict__ ref_cell) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_tmp) return;
    const uint32_t c = tmp_cell[r];
    const uint32_t j = r - tmp_start[c];
    if (j < cap) {
        indices[final_start[c] + j] = sorted[r];
        ref_cell[final_start[c] + j] = c;
    }
}

// ---- duplicate triangles -------------------------------------------------------------------------------
// A triangle whose nine vertex words equal those of a triangle with a lower id is met right after it in every cell
// list (same bounds, same cells; lists are ascending and the cap keeps the lowest ids) and gives the same t. The
// shader replaces the closest hit only when t is strictly smaller, so the copy can neither win nor be accepted
// where the original was rejected - provided the original's acceptance depends on geometry alone (no alpha rule).

__global__ void k_tri_hash(const float* __restrict__ tri_v, uint32_t nt, unsigned long long* __restrict__ keys,
                           uint32_t* __restrict__ vals) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nt) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(tri_v) + 9 * (size_t)k;
    unsigned long long h = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 9; ++i) {
        h ^= w[i];
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 29;
    }
    keys[k] = h;
    vals[k] = k;
}

__global__ void k_run_heads(const unsigned long long* __restrict__ keys, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    flag[p] = (p == 0 || keys[p] != keys[p - 1]) ? p : 0u;
}

__global__ void k_mark_duplicates(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ head, uint32_t nt,
                                  const float* __restrict__ tri_v, const uint32_t* __restrict__ tri_mat,
                                  const float4* __restrict__ materials, uint32_t n_materials, uint32_t* __restrict__ dup) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nt) return;
    const uint32_t k = vals[p], j = vals[head[p]];  // j: lowest id with this key (the sort is stable)
    uint32_t d = 0;
    if (j < k) {
        const uint32_t* a = reinterpret_cast<const uint32_t*>(tri_v) + 9 * (size_t)k;
        const uint32_t* b = reinterpret_cast<const uint32_t*>(tri_v) + 9 * (size_t)j;
        bool same = true;
        for (int i = 0; i < 9; ++i) same = same && a[i] == b[i];
        bool plain = true;  // the original's acceptance depends on geometry alone
        const uint32_t mat = tri_mat[j];
        if (mat < n_materials) {
            const uint32_t mode = __float_as_uint(materials[4 * mat + 3].x);
            plain = mode != 1u && mode != 2u;
        }
        d = same && plain ? 1u : 0u;
    }
    dup[k] = d;
}

__global__ void k_keep_flags(const uint32_t* __restrict__ indices, uint32_t n_refs, uint32_t nb, uint32_t nt,
                             const uint32_t* __restrict__ dup, uint32_t* __restrict__ keep) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_refs) return;
    uint32_t v = 0;
    if (r < n_refs) {
        const uint32_t id = indices[r];
        v = (id >= nb && id - nb < nt && dup[id - nb]) ? 0u : 1u;
    }
    keep[r] = v;  // entry n_refs closes the scan
}

__global__ void k_compact_refs(const uint32_t* __restrict__ indices, const uint32_t* __restrict__ ref_cell,
                               const uint32_t* __restrict__ pos, uint32_t n_refs, uint32_t* __restrict__ out,
                               uint32_t* __restrict__ out_cell) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_refs) return;
    if (pos[r + 1] != pos[r]) {
        out[pos[r]] = indices[r];
        out_cell[pos[r]] = ref_cell[r];
    }
}

__global__ void k_compact_starts(const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ pos, uint32_t n_cells,
                                 uint32_t* __restrict__ out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c <= n_cells) out[c] = pos[cell_start[c]];
}


// ---- rebuild without host round trips ----------------------------------------------------------------------
// The animated path rebuilds the grid of a small scene every frame (rt_update_vertices / rt_skin_vertices, then
// rt_build_grid). The general path above asks the host for sizes four times per build (bounds -> dimensions, pair
// count, reference count, duplicates) and goes through about thirty launches; for a scene of a few thousand triangles
// those waits and launch gaps cost more than the kernels. This path keeps every size on the device (FastMeta), launches
// over the CAPACITY of the buffers the last build left behind, and reads a few words back once, at the end. Whatever does
// not fit - more cells, pairs or references than the buffers hold, a cell list too long for the quadratic ranking
// below - raises `overflow`, and the host then runs the general path, which is exact for everything. The cell lists
// are the same either way: same bounds (min / max are order-independent), same dimensions (the same correctly rounded
// f32 operations), ids ascending inside a cell, the lowest `cap` ids kept.
struct FastMeta {
    float bmin[3], bmax[3];
    uint32_t gx, gy, gz, n_cells;
    uint32_t n_refs, overflow;
    unsigned long long n_pairs;  // summed in 64 bits: a total that does not fit must be seen, not wrapped
};
static_assert(sizeof(FastMeta) == 56, "FastMeta: one small read-back at the end of a build");

struct FastCaps {
    uint32_t cells;     // entries of the per-cell arrays minus one
    uint32_t bordered;  // records of the bordered cell array
    uint32_t pairs;     // entries of the per-pair arrays
    uint32_t refs;      // reference records minus the padding record
};

constexpr uint32_t FAST_MAX_OBJECTS = 65536;  // one workgroup scans the per-object pair counts
constexpr uint32_t FAST_MAX_LIST = 1024;      // longest cell list the ranking kernel accepts

__device__ inline BinGrid fast_grid(const FastMeta& m, float cell, uint32_t nb, uint32_t nt, int tri_mode) {
    BinGrid g;
    g.bmin[0] = m.bmin[0], g.bmin[1] = m.bmin[1], g.bmin[2] = m.bmin[2];
    g.cell = cell;
    g.gx = m.gx, g.gy = m.gy, g.gz = m.gz;
    g.nb = nb, g.nt = nt;
    g.tri_mode = tri_mode;
    return g;
}

// Bounds, dimensions, per-object pair counts and their scan in ONE workgroup and one launch (the rebuild of a small
// scene is a chain of launches that each cost as much as their work): the bounds of all objects, padded, the grid's
// dimensions and the capacity checks; the per-cell counters cleared; then every thread takes a run of consecutive objects, counts their cells,
// the workgroup scans the threads' totals, and every thread writes the exclusive prefix of its run (obj_off[total] = all
// pairs, as the scan of a closing zero gave). Sums wrap in 32 bits as the library scan's did; the 64-bit total goes to
// meta->n_pairs, which is what the capacity checks read.
__global__ __launch_bounds__(1024) void kf_head(const BoxData* __restrict__ boxes, const float* __restrict__ tri_v, uint32_t nb, uint32_t nt,
                                               float cell_size, int tri_mode, FastCaps caps, FastMeta* __restrict__ meta,
                                               uint32_t* __restrict__ obj_off, uint32_t* __restrict__ counts, uint32_t n_counts) {
    __shared__ float red[6][1024];
    __shared__ FastMeta sm;
    __shared__ unsigned long long part[1024];
    const uint32_t t = threadIdx.x, total = nb + nt;
    V3 mn = v3(INFINITY, INFINITY, INFINITY), mx = v3(-INFINITY, -INFINITY, -INFINITY);
    for (uint32_t id = t; id < total; id += 1024) {
        V3 a, b;
        object_bounds(boxes, tri_v, nb, id, a, b);
        mn = vmin(mn, a);
        mx = vmax(mx, b);
    }
    red[0][t] = mn.x, red[1][t] = mn.y, red[2][t] = mn.z;
    red[3][t] = mx.x, red[4][t] = mx.y, red[5][t] = mx.z;
    __syncthreads();
    for (uint32_t s = 512; s > 0; s >>= 1) {
        if (t < s) {
            for (int k = 0; k < 3; ++k) red[k][t] = fminf(red[k][t], red[k][t + s]);
            for (int k = 3; k < 6; ++k) red[k][t] = fmaxf(red[k][t], red[k][t + s]);
        }
        __syncthreads();
    }
    if (t == 0) {
        FastMeta m;
        memset(&m, 0, sizeof(m));
        const V3 bmin = v3(red[0][0], red[1][0], red[2][0]) - v3(1.0f, 1.0f, 1.0f);  // src/grid.rs:130-134
        const V3 bmax = v3(red[3][0], red[4][0], red[5][0]) + v3(1.0f, 1.0f, 1.0f);
        m.bmin[0] = bmin.x, m.bmin[1] = bmin.y, m.bmin[2] = bmin.z;
        m.bmax[0] = bmax.x, m.bmax[1] = bmax.y, m.bmax[2] = bmax.z;
        const V3 e = bmax - bmin;
        const float ext[3] = {e.x, e.y, e.z};
        const float div = tri_mode ? fmaxf(cell_size, RT_TRI_MIN_CELL_SIZE) : cell_size;
        unsigned long long dims[3];
        for (int k = 0; k < 3; ++k) {  // src/grid.rs:9-16; src/grid_triangles.rs:126-139
            const float q = ceilf(ext[k] / div);
            unsigned long long v = !(q > 0.0f) ? 0ull : (q >= 4294967296.0f ? 0xFFFFFFFFull : (unsigned long long)q);
            if (tri_mode) v = v < 1 ? 1 : (v > RT_TRI_MAX_GRID_DIM ? RT_TRI_MAX_GRID_DIM : v);
            dims[k] = v;
        }
        const bool bad = dims[0] == 0 || dims[1] == 0 || dims[2] == 0 || dims[0] > 0xFFFFull * 16 || dims[1] > 0xFFFFull * 16 ||
                         dims[2] > 0xFFFFull * 16;
        unsigned long long cells = bad ? 0 : dims[0] * dims[1] * dims[2];
        const unsigned long long bordered = bad ? 0 : (dims[0] + 2) * (dims[1] + 2) * (dims[2] + 2);
        if (bad || cells > caps.cells || bordered > caps.bordered) {
            m.overflow = 1;
            cells = 0;
        }
        m.gx = (uint32_t)dims[0], m.gy = (uint32_t)dims[1], m.gz = (uint32_t)dims[2];
        m.n_cells = (uint32_t)cells;
        sm = m;
        *meta = m;
    }
    __syncthreads();
    for (uint32_t k = t; k < n_counts; k += 1024) counts[k] = 0;  // the per-cell counters start at zero
    const uint32_t per = (total + 1 + 1023) / 1024;
    const uint32_t a = t * per < total + 1 ? t * per : total + 1, b = a + per < total + 1 ? a + per : total + 1;
    const BinGrid g = fast_grid(sm, cell_size, nb, nt, tri_mode);
    unsigned long long sum = 0;
    for (uint32_t id = a; id < b; ++id) {
        uint32_t n = 0;
        if (id < total && !sm.overflow) {
            V3 lo, hi;
            object_bounds(boxes, tri_v, nb, id, lo, hi);
            const Range3 r = cell_range(g, lo, hi);
            const unsigned long long v = (unsigned long long)(r.hi[0] - r.lo[0] + 1) * (unsigned long long)(r.hi[1] - r.lo[1] + 1) *
                                         (unsigned long long)(r.hi[2] - r.lo[2] + 1);
            n = r.lo[0] > r.hi[0] ? 0u : (v > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)v);
        }
        obj_off[id] = n;
        sum += n;
    }
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {  // inclusive scan of the threads' totals
        const unsigned long long below = t >= off ? part[t - off] : 0ull;
        __syncthreads();
        part[t] += below;
        __syncthreads();
    }
    unsigned long long run = part[t] - sum;
    for (uint32_t id = a; id < b; ++id) {
        const uint32_t n = obj_off[id];
        obj_off[id] = (uint32_t)run;
        run += n;
    }
    if (t == 1023) meta->n_pairs = part[1023];
}

// One lane per (object, cell) pair: the pair's cell, id and its slot in the cell's list (arbitrary order).
__global__ __launch_bounds__(256) void kf_bin_count(FastMeta* __restrict__ meta, FastCaps caps, float cell, uint32_t nb, uint32_t nt,
                                                    int tri_mode, const BoxData* __restrict__ boxes, const float* __restrict__ tri_v,
                                                    const uint32_t* __restrict__ obj_off, uint32_t* __restrict__ counts,
                                                    uint32_t* __restrict__ slots, uint32_t* __restrict__ pair_cell,
                                                    uint32_t* __restrict__ pair_id) {
    if (meta->n_pairs > caps.pairs) {  // more pairs than the buffers (and this launch) cover
        if (blockIdx.x == 0 && threadIdx.x == 0) meta->overflow = 1;
        return;
    }
    const uint32_t n_pairs = (uint32_t)meta->n_pairs;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= n_pairs) return;  // whole workgroup
    __shared__ uint32_t window[2];
    if (threadIdx.x < 2) {
        const uint32_t last = n_pairs - 1;
        const uint32_t q = threadIdx.x == 0 ? blockIdx.x * blockDim.x : min(blockIdx.x * blockDim.x + blockDim.x - 1, last);
        uint32_t lo = 0, hi = nb + nt;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) / 2;
            if (obj_off[mid] <= q) lo = mid; else hi = mid;
        }
        window[threadIdx.x] = lo;
    }
    __syncthreads();
    if (p >= n_pairs) return;
    uint32_t lo = window[0], hi = window[1] + 1;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (obj_off[mid] <= p) lo = mid; else hi = mid;
    }
    const uint32_t id = lo;
    const BinGrid g = fast_grid(*meta, cell, nb, nt, tri_mode);
    V3 mn, mx;
    object_bounds(boxes, tri_v, nb, id, mn, mx);
    const Range3 r = cell_range(g, mn, mx);
    const uint32_t nx = (uint32_t)(r.hi[0] - r.lo[0] + 1), ny = (uint32_t)(r.hi[1] - r.lo[1] + 1);
    const uint32_t j = p - obj_off[id];
    const uint32_t x = (uint32_t)r.lo[0] + j % nx, y = (uint32_t)r.lo[1] + (j / nx) % ny, z = (uint32_t)r.lo[2] + j / (nx * ny);
    const uint32_t c = x + y * g.gx + z * g.gx * g.gy;
    slots[p] = atomicAdd(&counts[c], 1u);
    pair_cell[p] = c;
    pair_id[p] = id;
}

// The two prefix sums every cell needs travel through ONE scan as the halves of a 64-bit word: where the cell's unsorted
// ids go (sum of the counts, high half) and where its kept ids go (sum of the counts cut to the cap, low half). Neither
// half can carry into the other: both totals are at most the pair count, below 2^32.
struct PackCounts {
    uint32_t cap;
    __device__ unsigned long long operator()(uint32_t v) const { return ((unsigned long long)v << 32) | (v < cap ? v : cap); }
};

// Thread t has two jobs: entry t of the scan is split into the two 32-bit prefix arrays the later kernels and
// rt_read_grid_csr read, and pair t's id goes to its slot of its cell's unsorted list (the place comes straight from
// the scanned word: the split of that cell may not have happened yet).
__global__ __launch_bounds__(256) void kf_split_fill(FastMeta* __restrict__ meta, const unsigned long long* __restrict__ scanned,
                                                     uint32_t n_entries, uint32_t* __restrict__ tmp_start, uint32_t* __restrict__ final_start,
                                                     const uint32_t* __restrict__ slots, const uint32_t* __restrict__ pair_cell,
                                                     const uint32_t* __restrict__ pair_id, uint32_t* __restrict__ tmp_ids) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_entries) {
        const unsigned long long v = scanned[t];
        tmp_start[t] = (uint32_t)(v >> 32);
        final_start[t] = (uint32_t)v;
        if (t == meta->n_cells) meta->n_refs = (uint32_t)v;
    }
    if (!meta->overflow && t < meta->n_pairs) tmp_ids[(uint32_t)(scanned[pair_cell[t]] >> 32) + slots[t]] = pair_id[t];
}

// Sort, cap and compaction in one step: an id's place in its cell's final list is the number of smaller ids in that
// cell (ids are distinct inside a cell); places beyond the cap are dropped, which keeps the lowest ids. A list too long
// for this quadratic count, or more references than the buffers hold, raises `overflow`.
__global__ __launch_bounds__(256) void kf_rank_place(FastMeta* __restrict__ meta, FastCaps caps, const uint32_t* __restrict__ pair_cell,
                                                     const uint32_t* __restrict__ pair_id, const uint32_t* __restrict__ tmp_start,
                                                     const uint32_t* __restrict__ tmp_ids, const uint32_t* __restrict__ final_start,
                                                     uint32_t cap, uint32_t* __restrict__ indices, uint32_t* __restrict__ ref_cell) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (meta->overflow || p >= meta->n_pairs) return;
    if (meta->n_refs > caps.refs) {
        meta->overflow = 1;
        return;
    }
    const uint32_t c = pair_cell[p], id = pair_id[p];
    const uint32_t a = tmp_start[c], b = tmp_start[c + 1];
    if (b - a > FAST_MAX_LIST) {
        meta->overflow = 1;
        return;
    }
    uint32_t rank = 0;
    for (uint32_t k = a; k < b; ++k) rank += tmp_ids[k] < id ? 1u : 0u;
    if (rank < cap) {
        indices[final_start[c] + rank] = id;
        ref_cell[final_start[c] + rank] = c;
    }
}

// k_emit_cells and k_emit_records in one launch, their sizes read from the device: thread t writes bordered cell record
// t and reference record t, whichever of the two exist.
__global__ void kf_emit(const FastMeta* __restrict__ meta, float cell, uint32_t nb, uint32_t nt, int tri_mode, const BoxData* __restrict__ boxes,
                        const float* __restrict__ tri_v, const uint32_t* __restrict__ indices, const uint32_t* __restrict__ ref_cell,
                        const uint32_t* __restrict__ final_start, uint32_t list_cap, uint32_t mask_cap, const uint32_t* __restrict__ tri_mat,
                        const float4* __restrict__ materials, uint32_t n_materials, uint2* __restrict__ cells, float4* __restrict__ refs) {
    if (meta->overflow) return;
    const uint32_t gx = meta->gx, gy = meta->gy, gz = meta->gz;
    const uint32_t px = gx + 2, py = gy + 2, pz = gz + 2;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < (size_t)px * py * pz) {
        const uint32_t x = (uint32_t)(p % px), y = (uint32_t)((p / px) % py), z = (uint32_t)(p / ((size_t)px * py));
        uint2 rec = make_uint2(0u, CELL_SENTINEL);
        if (x >= 1 && x <= gx && y >= 1 && y <= gy && z >= 1 && z <= gz) {
            const size_t c = (size_t)(x - 1) + (size_t)(y - 1) * gx + (size_t)(z - 1) * gx * gy;
            const uint32_t a = final_start[c], b = final_start[c + 1];
            rec = make_uint2(a, b - a < list_cap ? b - a : list_cap);
        }
        cells[p] = rec;
    }
    if (p < meta->n_refs)
        emit_record(fast_grid(*meta, cell, nb, nt, tri_mode), boxes, tri_v, indices, ref_cell, final_start, mask_cap, tri_mat, materials,
                    n_materials, refs, (uint32_t)p);
}

}  // namespace

namespace rt {

int rebuild_trace_records(rt_ctx* c) {
    hipStream_t st = c->stream;
    c->t_cell_start = c->cell_start.as<uint32_t>();
    c->t_indices = c->indices.as<uint32_t>();
    c->t_ref_cell = c->grid_built ? c->ref_cell.as<uint32_t>() : nullptr;
    c->n_refs_t = c->n_refs;
    const uint32_t nb = c->grid_nb, nt = c->nt, n_refs = c->n_refs, n_cells = c->n_cells;
    // Only for lists this library built (lowest ids survive the cap), when the shader's own cap does not cut them
    // further, and not for the counting kernels, which walk what the reference walks.
    const bool dedup = c->grid_built && c->params.skip_duplicate_triangles && c->params.collect_stats != 1 && nt > 1 && n_refs > 0 &&
                       c->params.max_objects_per_cell >= c->build_cap && c->dup_state != 0;
    if (dedup) {
        RT_HIP(c, c->dd_keys.reserve((size_t)nt * 8));
        RT_HIP(c, c->dd_keys2.reserve((size_t)nt * 8));
        RT_HIP(c, c->dd_vals.reserve((size_t)nt * 4));
        RT_HIP(c, c->dd_vals2.reserve((size_t)nt * 4));
        RT_HIP(c, c->dd_flag.reserve(std::max((size_t)nt, (size_t)n_refs + 1) * 4));  // run heads, then keep flags
        RT_HIP(c, c->dd_head.reserve((size_t)nt * 4));
        RT_HIP(c, c->dd_dup.reserve((size_t)nt * 4));
        unsigned long long* keys = c->dd_keys.as<unsigned long long>();
        unsigned long long* keys2 = c->dd_keys2.as<unsigned long long>();
        uint32_t *vals = c->dd_vals.as<uint32_t>(), *vals2 = c->dd_vals2.as<uint32_t>();
        uint32_t *flag = c->dd_flag.as<uint32_t>(), *head = c->dd_head.as<uint32_t>(), *dup = c->dd_dup.as<uint32_t>();
        const dim3 tg((nt + 255) / 256);
        hipLaunchKernelGGL(k_tri_hash, tg, dim3(256), 0, st, c->tri_v.as<float>(), nt, keys, vals);
        RT_HIP(c, hipGetLastError());
        size_t bytes = 0;
        RT_HIP(c, rocprim::radix_sort_pairs(nullptr, bytes, keys, keys2, vals, vals2, (size_t)nt, 0, 64, st));
        RT_HIP(c, c->b_scan_tmp.reserve(bytes + 256));
        RT_HIP(c, rocprim::radix_sort_pairs(c->b_scan_tmp.p, bytes, keys, keys2, vals, vals2, (size_t)nt, 0, 64, st));
        hipLaunchKernelGGL(k_run_heads, tg, dim3(256), 0, st, (const unsigned long long*)keys2, nt, flag);
        RT_HIP(c, hipGetLastError());
        RT_HIP(c, rocprim::inclusive_scan(nullptr, bytes, flag, head, (size_t)nt, rocprim::maximum<uint32_t>(), st));
        RT_HIP(c, c->b_scan_tmp.reserve(bytes + 256));
        RT_HIP(c, rocprim::inclusive_scan(c->b_scan_tmp.p, bytes, flag, head, (size_t)nt, rocprim::maximum<uint32_t>(), st));
        hipLaunchKernelGGL(k_mark_duplicates, tg, dim3(256), 0, st, (const uint32_t*)vals2, (const uint32_t*)head, nt,
                           c->tri_v.as<float>(), c->tri_mat.as<uint32_t>(), c->materials.as<float4>(), c->n_materials, dup);
        RT_HIP(c, hipGetLastError());
        // which references stay, and where they go
        RT_HIP(c, c->tc_pos.reserve(((size_t)n_refs + 1) * 4));
        uint32_t* keep = c->dd_flag.as<uint32_t>();
        uint32_t* pos = c->tc_pos.as<uint32_t>();
        hipLaunchKernelGGL(k_keep_flags, dim3((n_refs + 1 + 255) / 256), dim3(256), 0, st, c->indices.as<uint32_t>(), n_refs, nb,
                           nt, (const uint32_t*)dup, keep);
        RT_HIP(c, hipGetLastError());
        RT_HIP(c, rocprim::exclusive_scan(nullptr, bytes, keep, pos, 0u, (size_t)n_refs + 1, rocprim::plus<uint32_t>(), st));
        RT_HIP(c, c->b_scan_tmp.reserve(bytes + 256));
        RT_HIP(c, rocprim::exclusive_scan(c->b_scan_tmp.p, bytes, keep, pos, 0u, (size_t)n_refs + 1, rocprim::plus<uint32_t>(), st));
        uint32_t kept = 0;
        RT_HIP(c, hipMemcpyAsync(&kept, pos + n_refs, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RT_HIP(c, hipStreamSynchronize(st));
        c->dup_state = kept < n_refs ? 1 : 0;
        c->dup_share = (float)(n_refs - kept) / (float)n_refs;
        if (kept < n_refs) {
            RT_HIP(c, c->tc_indices.reserve((size_t)kept * 4 + 16));
            RT_HIP(c, c->tc_ref_cell.reserve((size_t)kept * 4 + 16));
            RT_HIP(c, c->tc_start.reserve(((size_t)n_cells + 1) * 4));
            hipLaunchKernelGGL(k_compact_refs, dim3((n_refs + 255) / 256), dim3(256), 0, st, c->indices.as<uint32_t>(),
                               (const uint32_t*)c->ref_cell.as<uint32_t>(), (const uint32_t*)pos, n_refs,
                               c->tc_indices.as<uint32_t>(), c->tc_ref_cell.as<uint32_t>());
            hipLaunchKernelGGL(k_compact_starts, dim3((n_cells + 1 + 255) / 256), dim3(256), 0, st, c->cell_start.as<uint32_t>(),
                               (const uint32_t*)pos, n_cells, c->tc_start.as<uint32_t>());
            RT_HIP(c, hipGetLastError());
            c->t_cell_start = c->tc_start.as<uint32_t>();
            c->t_indices = c->tc_indices.as<uint32_t>();
            c->t_ref_cell = c->tc_ref_cell.as<uint32_t>();
            c->n_refs_t = kept;
        }
    }
    if (int r = emit_cells(c)) return r;
    if (c->n_refs_t) {
        BinGrid g;
        g.bmin[0] = c->meta.bounds_min[0], g.bmin[1] = c->meta.bounds_min[1], g.bmin[2] = c->meta.bounds_min[2];
        g.cell = c->meta.finest_cell_size;
        g.gx = c->meta.grid_sizes[RT_GRID_LEVELS - 1][0];
        g.gy = c->meta.grid_sizes[RT_GRID_LEVELS - 1][1];
        g.gz = c->meta.grid_sizes[RT_GRID_LEVELS - 1][2];
        g.nb = nb, g.nt = nt;
        g.tri_mode = c->grid_mode == RT_GRID_TRIANGLES ? 1 : 0;
        const uint32_t cap = std::min(c->build_cap, c->params.max_objects_per_cell);
        hipLaunchKernelGGL(k_emit_records, dim3((c->n_refs_t + 255) / 256), dim3(256), 0, st, g, c->boxes.as<BoxData>(),
                           c->tri_v.as<float>(), c->t_indices, c->t_ref_cell, (const uint32_t*)c->cell_start.as<uint32_t>(),
                           c->n_refs_t, cap, c->tri_mat.as<uint32_t>(), c->materials.as<float4>(), c->n_materials,
                           c->refs.as<float4>());
        RT_HIP(c, hipGetLastError());
    }
    return RT_OK;
}


// Dimensions of the three coarse levels, reported only (src/grid.rs:302-322); their counts are not built.
static void coarse_dims(GridMetadata& meta, bool tri_mode) {
    const float cell_size = meta.finest_cell_size;
    for (uint32_t level = 0; level + 1 < RT_GRID_LEVELS; ++level) {
        const float cs = cell_size * (float)(1u << (RT_GRID_LEVELS - 1 - level));
        const float div = tri_mode ? fmaxf(cs, RT_TRI_MIN_CELL_SIZE) : cs;
        const float ext[3] = {meta.bounds_max[0] - meta.bounds_min[0], meta.bounds_max[1] - meta.bounds_min[1],
                              meta.bounds_max[2] - meta.bounds_min[2]};
        for (int k = 0; k < 3; ++k) {
            float q = ceilf(ext[k] / div);
            uint64_t v = !(q > 0.0f) ? 0ull : (q >= 4294967296.0f ? 0xFFFFFFFFull : (uint64_t)q);
            if (tri_mode) v = std::min<uint64_t>(std::max<uint64_t>(v, 1), RT_TRI_MAX_GRID_DIM);
            meta.grid_sizes[level][k] = (uint32_t)v;
        }
    }
}

// The rebuild without host round trips (see FastMeta above). Returns RT_OK with *done = false when this build has to
// go through the general path: not eligible, or something did not fit.
static int build_grid_fast(rt_ctx* c, float cell_size, uint32_t max_per_cell, int mode, bool* done) {
    *done = false;
    const bool tri_mode = mode == RT_GRID_TRIANGLES;
    const uint32_t nb = tri_mode ? 0u : c->nb, nt = c->nt, total = nb + nt;
    // a grid of this scene, mode, cell size and cap has been built before (its buffers are the capacities), and
    // nothing needs the duplicate search
    if (getenv("RT_DEBUG_BUILD"))
        fprintf(stderr, "fast build: ready %d total %u/%u mode %d/%d cell %g/%g cap %u/%u dup %d\n", (int)c->fast_ready, total, c->fast_total,
                c->grid_mode, mode, c->fast_cell, cell_size, c->build_cap, max_per_cell, c->dup_state);
    if (!c->fast_ready || total == 0 || total > FAST_MAX_OBJECTS || c->fast_total != total || c->grid_mode != mode ||
        c->fast_cell != cell_size || c->build_cap != max_per_cell || c->params.max_objects_per_cell < max_per_cell)
        return RT_OK;
    // Duplicate triangles: this path does not search for them. Not searched yet -> general path. Found, but they are
    // less than a twentieth of the references (a model with a few doubled faces) -> tested like any other triangle
    // here, which changes no answer; a scene made of copies keeps the general path, whose search pays for itself.
    if (c->params.skip_duplicate_triangles && c->params.collect_stats != 1 && nt > 1 &&
        (c->dup_state < 0 || (c->dup_state == 1 && c->dup_share > 0.05f)))
        return RT_OK;
    auto words = [](const DevBuf& b) { return (uint32_t)std::min<size_t>(b.cap / 4, 0xFFFFFFF0u); };
    FastCaps caps;
    caps.cells = std::min({words(c->b_counts), words(c->b_tmp_start), words(c->cell_start)});
    caps.cells = caps.cells ? caps.cells - 1 : 0;
    caps.bordered = (uint32_t)std::min<size_t>(c->cells.cap / sizeof(uint2), 0xFFFFFFF0u);
    caps.pairs = std::min({words(c->b_flags), words(c->b_tmp_ids), words(c->b_tmp_sorted), words(c->b_tmp_cell)});
    const size_t ref_recs = c->refs.cap / 48;
    caps.refs = std::min({words(c->indices), words(c->ref_cell), (uint32_t)std::min<size_t>(ref_recs ? ref_recs - 1 : 0, 0xFFFFFFF0u)});
    if (!caps.cells || !caps.pairs || !caps.refs || !caps.bordered || words(c->b_obj_n) < total + 1 || words(c->b_obj_off) < total + 1)
        return RT_OK;
    hipStream_t st = c->stream;
    RT_HIP(c, c->fast_meta.reserve(sizeof(FastMeta)));
    FastMeta* meta = c->fast_meta.as<FastMeta>();
    const int tm = tri_mode ? 1 : 0;
    const BoxData* boxes = c->boxes.as<BoxData>();
    const float* tri_v = c->tri_v.as<float>();
    uint32_t *counts = c->b_counts.as<uint32_t>(), *tmp_start = c->b_tmp_start.as<uint32_t>();
    uint32_t *final_start = c->cell_start.as<uint32_t>(), *obj_off = c->b_obj_off.as<uint32_t>();
    uint32_t *slots = c->b_flags.as<uint32_t>(), *tmp_ids = c->b_tmp_ids.as<uint32_t>(), *pair_id = c->b_tmp_sorted.as<uint32_t>();
    uint32_t* pair_cell = c->b_tmp_cell.as<uint32_t>();
    // launches cover what the last build needed plus a margin, never more than the buffers hold: a frame that needs
    // more raises `overflow` (the pair count is checked against the launch's reach too)
    auto reach = [](uint32_t last, uint32_t cap) { return std::min<uint64_t>(cap, (uint64_t)last + last / 2 + 4096); };
    FastCaps lim = caps;
    lim.cells = (uint32_t)reach(c->n_cells, caps.cells);
    lim.bordered = (uint32_t)reach(c->fast_bordered, caps.bordered);
    lim.pairs = (uint32_t)reach(c->fast_pairs, caps.pairs);
    lim.refs = (uint32_t)reach(c->n_refs, caps.refs);
    hipLaunchKernelGGL(kf_head, dim3(1), dim3(1024), 0, st, boxes, tri_v, nb, nt, cell_size, tm, lim, meta, obj_off, counts, lim.cells + 1);
    // the scan below runs over a size the host knows - the CAPACITY of the per-cell arrays (entries past the grid's last
    // cell are zero, so every prefix up to it is what a scan of the exact length would give)
    size_t bytes = 0;
    const dim3 pair_grid((lim.pairs + 255) / 256);
    hipLaunchKernelGGL(kf_bin_count, pair_grid, dim3(256), 0, st, meta, lim, cell_size, nb, nt, tm, boxes, tri_v, (const uint32_t*)obj_off,
                       counts, slots, pair_cell, pair_id);
    const size_t n_entries = (size_t)lim.cells + 1;
    RT_HIP(c, c->fast_scan.reserve(n_entries * sizeof(unsigned long long)));
    unsigned long long* scanned = c->fast_scan.as<unsigned long long>();
    auto packed = rocprim::make_transform_iterator((const uint32_t*)counts, PackCounts{max_per_cell});
    RT_HIP(c, rocprim::exclusive_scan(nullptr, bytes, packed, scanned, 0ull, n_entries, rocprim::plus<unsigned long long>(), st));
    RT_HIP(c, c->b_scan_tmp.reserve(bytes + 256));
    RT_HIP(c, rocprim::exclusive_scan(c->b_scan_tmp.p, bytes, packed, scanned, 0ull, n_entries, rocprim::plus<unsigned long long>(), st));
    hipLaunchKernelGGL(kf_split_fill, dim3((uint32_t)((std::max<size_t>(n_entries, lim.pairs) + 255) / 256)), dim3(256), 0, st, meta,
                       (const unsigned long long*)scanned, (uint32_t)n_entries, tmp_start, final_start, (const uint32_t*)slots,
                       (const uint32_t*)pair_cell, (const uint32_t*)pair_id, tmp_ids);
    hipLaunchKernelGGL(kf_rank_place, pair_grid, dim3(256), 0, st, meta, lim, (const uint32_t*)pair_cell, (const uint32_t*)pair_id,
                       (const uint32_t*)tmp_start, (const uint32_t*)tmp_ids, (const uint32_t*)final_start, max_per_cell,
                       c->indices.as<uint32_t>(), c->ref_cell.as<uint32_t>());
    const uint32_t trace_cap = std::min(max_per_cell, c->params.max_objects_per_cell);
    hipLaunchKernelGGL(kf_emit, dim3((std::max(lim.bordered, lim.refs) + 255) / 256), dim3(256), 0, st, (const FastMeta*)meta, cell_size, nb, nt,
                       tm, boxes, tri_v, (const uint32_t*)c->indices.as<uint32_t>(), (const uint32_t*)c->ref_cell.as<uint32_t>(),
                       (const uint32_t*)final_start, c->params.max_objects_per_cell, trace_cap, c->tri_mat.as<uint32_t>(),
                       c->materials.as<float4>(), c->n_materials, c->cells.as<uint2>(), c->refs.as<float4>());
    RT_HIP(c, hipGetLastError());
    static_assert(sizeof(FastMeta) <= 256, "the landing buffer holds one FastMeta");
    if (!c->fast_landing) RT_HIP(c, hipHostMalloc(&c->fast_landing, 256, hipHostMallocDefault));
    RT_HIP(c, hipMemcpyAsync(c->fast_landing, meta, sizeof(FastMeta), hipMemcpyDeviceToHost, st));
    RT_HIP(c, wait_short(st));
    (void)hipGetLastError();
    const FastMeta h = *static_cast<const FastMeta*>(c->fast_landing);
    if (getenv("RT_DEBUG_BUILD"))
        fprintf(stderr, "fast build: overflow %u dims %u %u %u cells %u pairs %llu refs %u | lim cells %u bordered %u pairs %u refs %u\n",
                h.overflow, h.gx, h.gy, h.gz, h.n_cells, h.n_pairs, h.n_refs, lim.cells, lim.bordered, lim.pairs, lim.refs);
    if (h.overflow) return RT_OK;  // the general path builds this frame (and leaves larger buffers behind)
    GridMetadata gm;
    memset(&gm, 0, sizeof(gm));
    for (int k = 0; k < 3; ++k) {
        gm.bounds_min[k] = h.bmin[k];
        gm.bounds_max[k] = h.bmax[k];
    }
    gm.num_levels = RT_GRID_LEVELS;
    gm.finest_cell_size = cell_size;
    gm.grid_sizes[RT_GRID_LEVELS - 1][0] = h.gx, gm.grid_sizes[RT_GRID_LEVELS - 1][1] = h.gy, gm.grid_sizes[RT_GRID_LEVELS - 1][2] = h.gz;
    coarse_dims(gm, tri_mode);
    c->meta = gm;
    c->n_cells = h.n_cells;
    c->n_refs = c->n_refs_t = h.n_refs;
    c->grid_nb = nb;
    c->grid_built = true;
    c->t_cell_start = c->cell_start.as<uint32_t>();
    c->t_indices = c->indices.as<uint32_t>();
    c->t_ref_cell = c->ref_cell.as<uint32_t>();
    c->fast_pairs = (uint32_t)h.n_pairs;
    c->fast_bordered = (h.gx + 2) * (h.gy + 2) * (h.gz + 2);
    *done = true;
    return RT_OK;
}


int build_grid_device(rt_ctx* c, float cell_size, uint32_t max_per_cell, int mode) {
    {
        bool done = false;
        if (int r = build_grid_fast(c, cell_size, max_per_cell, mode, &done)) return r;
        c->last_build_fast = done;
        if (done) return RT_OK;
    }
    const bool tri_mode = mode == RT_GRID_TRIANGLES;
    const uint32_t nb = tri_mode ? 0u : c->nb;
    const uint32_t nt = c->nt;
    const uint32_t total = nb + nt;
    hipStream_t st = c->stream;

    // ---- bounds: union of all object bounds, padded by 1.0 (src/grid.rs:112-134; src/grid_triangles.rs:30-55)
    V3 bmin = v3(-1.0f, -1.0f, -1.0f), bmax = v3(1.0f, 1.0f, 1.0f);
    const bool tri_empty = tri_mode && nt == 0;
    if (total > 0) {
        const uint32_t blocks = std::min<uint32_t>((total + 255) / 256, 1024u);
        RT_HIP(c, c->b_partial.reserve(((size_t)blocks + 1) * 6 * sizeof(float)));
        float* partial = c->b_partial.as<float>();
        float* out6 = partial + (size_t)blocks * 6;
        hipLaunchKernelGGL(k_bounds_partial, dim3(blocks), dim3(256), 0, st, c->boxes.as<BoxData>(), c->tri_v.as<float>(),
                           nb, total, partial);
        hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(64), 0, st, partial, blocks, out6);
        RT_HIP(c, hipGetLastError());
        float h[6];
        RT_HIP(c, hipMemcpyAsync(h, out6, sizeof(h), hipMemcpyDeviceToHost, st));
        RT_HIP(c, hipStreamSynchronize(st));
        bmin = v3(h[0], h[1], h[2]);
        bmax = v3(h[3], h[4], h[5]);
    }
    if (!tri_empty) {
        bmin = bmin - v3(1.0f, 1.0f, 1.0f);
        bmax = bmax + v3(1.0f, 1.0f, 1.0f);
    }

    // ---- dimensions (src/grid.rs:9-16; src/grid_triangles.rs:126-139)
    uint64_t dims[3];
    {
        const V3 e = bmax - bmin;
        const float ext[3] = {e.x, e.y, e.z};
        const float div = tri_mode ? fmaxf(cell_size, RT_TRI_MIN_CELL_SIZE) : cell_size;
        for (int k = 0; k < 3; ++k) {
            float q = ceilf(ext[k] / div);
            uint64_t v = !(q > 0.0f) ? 0ull : (q >= 18446744073709551616.0f ? UINT64_MAX : (uint64_t)q);
            if (tri_mode) v = std::min<uint64_t>(std::max<uint64_t>(v, 1), RT_TRI_MAX_GRID_DIM);
            if (tri_empty) v = 1;
            dims[k] = v;
        }
    }
    if (dims[0] == 0 || dims[1] == 0 || dims[2] == 0 || dims[0] > 0x7FFFFFFF || dims[1] > 0x7FFFFFFF ||
        dims[2] > 0x7FFFFFFF || dims[0] * dims[1] > 0x7FFFFFFFull || dims[0] * dims[1] * dims[2] > 0x7FFFFFFFull)
        return fail(c, RT_ERR_LIMIT, "rt_build_grid: grid of %llu x %llu x %llu cells", (unsigned long long)dims[0],
                    (unsigned long long)dims[1], (unsigned long long)dims[2]);
    const uint32_t n_cells = (uint32_t)(dims[0] * dims[1] * dims[2]);

    GridMetadata meta;
    memset(&meta, 0, sizeof(meta));
    meta.bounds_min[0] = bmin.x, meta.bounds_min[1] = bmin.y, meta.bounds_min[2] = bmin.z;
    meta.bounds_max[0] = bmax.x, meta.bounds_max[1] = bmax.y, meta.bounds_max[2] = bmax.z;
    meta.num_levels = RT_GRID_LEVELS;
    meta.finest_cell_size = cell_size;
    for (int k = 0; k < 3; ++k) meta.grid_sizes[RT_GRID_LEVELS - 1][k] = (uint32_t)dims[k];
    if (!tri_empty) coarse_dims(meta, tri_mode);

    BinGrid g;
    g.bmin[0] = bmin.x, g.bmin[1] = bmin.y, g.bmin[2] = bmin.z;
    g.cell = cell_size;
    g.gx = (uint32_t)dims[0], g.gy = (uint32_t)dims[1], g.gz = (uint32_t)dims[2];
    g.nb = nb, g.nt = nt;
    g.tri_mode = tri_mode ? 1 : 0;

    // ---- pairs per object, their scan, per-cell counts
    const size_t cell_bytes = ((size_t)n_cells + 1) * sizeof(uint32_t);
    RT_HIP(c, c->b_counts.reserve(cell_bytes));
    RT_HIP(c, c->b_tmp_start.reserve(cell_bytes));
    RT_HIP(c, c->b_cursor.reserve(cell_bytes));
    RT_HIP(c, c->cell_start.reserve(cell_bytes));
    RT_HIP(c, c->b_obj_n.reserve(((size_t)total + 1) * sizeof(uint32_t)));
    RT_HIP(c, c->b_obj_off.reserve(((size_t)total + 1) * sizeof(uint32_t)));
    uint32_t* counts = c->b_counts.as<uint32_t>();
    uint32_t* tmp_start = c->b_tmp_start.as<uint32_t>();
    uint32_t* cursor = c->b_cursor.as<uint32_t>();
    uint32_t* final_start = c->cell_start.as<uint32_t>();
    uint32_t* obj_n = c->b_obj_n.as<uint32_t>();
    uint32_t* obj_off = c->b_obj_off.as<uint32_t>();
    RT_HIP(c, hipMemsetAsync(counts, 0, cell_bytes, st));
    size_t scan_bytes = 0;
    unsigned long long n_pairs64 = 0;
    if (total) {
        hipLaunchKernelGGL(k_obj_pairs, dim3((total + 1 + 255) / 256), dim3(256), 0, st, g, c->boxes.as<BoxData>(),
                           c->tri_v.as<float>(), obj_n);
        RT_HIP(c, hipGetLastError());
        // 64-bit sums: the total is checked against the limit below before anything is sized by it
        RT_HIP(c, c->b_obj_sum.reserve(sizeof(unsigned long long)));
        size_t red_bytes = 0;
        RT_HIP(c, rocprim::reduce(nullptr, red_bytes, obj_n, c->b_obj_sum.as<unsigned long long>(), 0ull, (size_t)total,
                                  rocprim::plus<unsigned long long>(), st));
        RT_HIP(c, c->b_scan_tmp.reserve(red_bytes + 256));
        RT_HIP(c, rocprim::reduce(c->b_scan_tmp.p, red_bytes, obj_n, c->b_obj_sum.as<unsigned long long>(), 0ull, (size_t)total,
                                  rocprim::plus<unsigned long long>(), st));
        RT_HIP(c, rocprim::exclusive_scan(nullptr, scan_bytes, obj_n, obj_off, 0u, (size_t)total + 1, rocprim::plus<uint32_t>(), st));
        RT_HIP(c, c->b_scan_tmp.reserve(scan_bytes + 256));
        RT_HIP(c, rocprim::exclusive_scan(c->b_scan_tmp.p, scan_bytes, obj_n, obj_off, 0u, (size_t)total + 1,
                                          rocprim::plus<uint32_t>(), st));
        RT_HIP(c, hipMemcpyAsync(&n_pairs64, c->b_obj_sum.p, sizeof(n_pairs64), hipMemcpyDeviceToHost, st));
        RT_HIP(c, hipStreamSynchronize(st));
    }
    if (n_pairs64 > 0xFFFFFFF0ull / 2)
        return fail(c, RT_ERR_LIMIT, "rt_build_grid: %llu cell references (max %u)", n_pairs64, 0xFFFFFFF0u / 2);
    const uint32_t n_tmp = (uint32_t)n_pairs64;
    const dim3 pair_grid((n_tmp + 255) / 256);
    if (n_tmp) {
        RT_HIP(c, c->b_flags.reserve((size_t)n_tmp * sizeof(uint32_t)));
        hipLaunchKernelGGL(k_bin<false>, pair_grid, dim3(256), 0, st, g, c->boxes.as<BoxData>(), c->tri_v.as<float>(),
                           (const uint32_t*)obj_off, n_tmp, counts, c->b_flags.as<uint32_t>(), (const uint32_t*)nullptr,
                           (uint32_t*)nullptr, (uint32_t*)nullptr);
        RT_HIP(c, hipGetLastError());
    }

    // ---- scan of the uncapped counts: where each cell's unsorted ids go
    RT_HIP(c, rocprim::exclusive_scan(nullptr, scan_bytes, counts, tmp_start, 0u, (size_t)n_cells + 1, rocprim::plus<uint32_t>(), st));
    RT_HIP(c, c->b_scan_tmp.reserve(scan_bytes + 256));
    RT_HIP(c, rocprim::exclusive_scan(c->b_scan_tmp.p, scan_bytes, counts, tmp_start, 0u, (size_t)n_cells + 1,
                                      rocprim::plus<uint32_t>(), st));

    uint32_t n_refs = 0;
    if (n_tmp) {
        // ---- fill (arbitrary order inside a cell), then sort every cell's ids ascending
        const size_t ref_bytes = (size_t)n_tmp * sizeof(uint32_t);
        RT_HIP(c, c->b_tmp_ids.reserve(ref_bytes));
        RT_HIP(c, c->b_tmp_sorted.reserve(ref_bytes));
        RT_HIP(c, c->b_tmp_cell.reserve(ref_bytes));
        uint32_t* tmp_ids = c->b_tmp_ids.as<uint32_t>();
        uint32_t* tmp_sorted = c->b_tmp_sorted.as<uint32_t>();
        uint32_t* tmp_cell = c->b_tmp_cell.as<uint32_t>();
        hipLaunchKernelGGL(k_bin<true>, pair_grid, dim3(256), 0, st, g, c->boxes.as<BoxData>(), c->tri_v.as<float>(),
                           (const uint32_t*)obj_off, n_tmp, (uint32_t*)nullptr, c->b_flags.as<uint32_t>(),
                           (const uint32_t*)tmp_start, tmp_ids, tmp_cell);
        RT_HIP(c, hipGetLastError());
        // rocPRIM's segmented sort launches work per segment and stops sorting (without an error) from about 2^25
        // segments on, so the cells go through it in batches; offsets stay absolute, so every batch addresses the
        // same two key arrays.
        const uint32_t batch = 1u << 24;
        size_t sort_bytes = 0;
        RT_HIP(c, rocprim::segmented_radix_sort_keys(nullptr, sort_bytes, tmp_ids, tmp_sorted, n_tmp, std::min(n_cells, batch),
                                                     tmp_start, tmp_start + 1, 0, 32, st));
        RT_HIP(c, c->b_scan_tmp.reserve(sort_bytes + 256));
        for (uint32_t first = 0; first < n_cells; first += batch) {
            const uint32_t n = std::min(batch, n_cells - first);
            size_t bytes = sort_bytes;
            RT_HIP(c, rocprim::segmented_radix_sort_keys(c->b_scan_tmp.p, bytes, tmp_ids, tmp_sorted, n_tmp, n, tmp_start + first,
                                                         tmp_start + first + 1, 0, 32, st));
        }

        // ---- cap: keep the first max_per_cell ids of every cell, compact
        hipLaunchKernelGGL(k_clamp_counts, dim3((n_cells + 1 + 255) / 256), dim3(256), 0, st, counts, n_cells, max_per_cell,
                           cursor);
        RT_HIP(c, hipGetLastError());
        RT_HIP(c, rocprim::exclusive_scan(nullptr, scan_bytes, cursor, final_start, 0u, (size_t)n_cells + 1,
                                          rocprim::plus<uint32_t>(), st));
        RT_HIP(c, c->b_scan_tmp.reserve(scan_bytes + 256));
        RT_HIP(c, rocprim::exclusive_scan(c->b_scan_tmp.p, scan_bytes, cursor, final_start, 0u, (size_t)n_cells + 1,
                                          rocprim::plus<uint32_t>(), st));
        RT_HIP(c, hipMemcpyAsync(&n_refs, final_start + n_cells, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RT_HIP(c, hipStreamSynchronize(st));
        RT_HIP(c, c->indices.reserve((size_t)n_refs * sizeof(uint32_t) + 16));
        RT_HIP(c, c->ref_cell.reserve((size_t)n_refs * sizeof(uint32_t) + 16));
        hipLaunchKernelGGL(k_compact, dim3((n_tmp + 255) / 256), dim3(256), 0, st, (const uint32_t*)tmp_sorted,
                           (const uint32_t*)tmp_cell, (const uint32_t*)tmp_start, (const uint32_t*)final_start, n_tmp,
                           max_per_cell, c->indices.as<uint32_t>(), c->ref_cell.as<uint32_t>());
        RT_HIP(c, hipGetLastError());
    } else {
        RT_HIP(c, hipMemsetAsync(final_start, 0, cell_bytes, st));
        RT_HIP(c, c->indices.reserve(16));
    }

    c->meta = meta;
    c->n_cells = n_cells;
    c->n_refs = n_refs;
    // a triangle-only grid numbers its triangles from 0 and knows no boxes (src/grid_triangles.rs:92-99)
    c->grid_nb = nb;
    c->grid_built = true;
    c->grid_mode = mode;
    c->build_cap = max_per_cell;
    if (int r = rebuild_trace_records(c)) return r;
    RT_HIP(c, hipStreamSynchronize(st));
    // what a later rebuild of this scene may reuse without asking the host for sizes (build_grid_fast)
    c->fast_ready = n_tmp != 0;
    c->fast_total = total;
    c->fast_cell = cell_size;
    c->fast_pairs = n_tmp;
    c->fast_bordered = (uint32_t)((dims[0] + 2) * (dims[1] + 2) * (dims[2] + 2));
    return RT_OK;
}

}  // namespace rt
