// Device side of the trace path: primary-ray generation, 3D-DDA over the uniform grid,
// per-cell candidate tests, closest-hit / any-hit resolution, shadow-ray generation.
//
// Replaces the geometry part of the reference's compute shader
// (src/raytracer_unified.wgsl:430-588 trace_ray, :625-653 ray generation).
//
// Data layout in HBM (built by rt_build.hip / the gather kernels in rt_api.hip):
//   cells[c]      uint2 (first ref, ref count cut to the per-cell cap) per fine grid cell, with a one-cell border on every side whose
//                 count is CELL_SENTINEL: c = (x+1) + (y+1)*px + (z+1)*px*py, px = gx+2, py = gy+2
//   refs[3*r..]   one 48-byte record per (cell, object) reference, stored in cell order so a
//                 cell's candidates are contiguous:  (v0.xyz, id bits) (e1.xyz, 0) (e2.xyz, 0)
//                 for a triangle, (0,0,0, id bits) for a box (box data is per frame, below); the spare word of
//                 the second quad holds the re-test mask (6 bits, see trace_ray)
//   tri_normal[k] per triangle k: (normalize(e1 x e2), 0) - read once per primary hit to place the shadow ray
//   box_frame[4*b..] per box, recomputed each frame: (lo.xyz, flag bits) (hi.xyz, 0)
//                 (centre.xyz, 0) (half.xyz, 0) with lo/hi = interpolated centre -/+ half
#pragma once

#include <type_traits>

#include "rt_math.hpp"
#include "rt_texture.hpp"

namespace rt {

constexpr uint32_t MISS = 0xFFFFFFFFu;
constexpr uint32_t REF_INVALID = 0xFFFFFFFEu;  // reference to a triangle index >= num_triangles: skipped
// Triangle whose material discards every hit (untextured alpha MASK below its cutoff, or BLEND below 0.01,
// raytracer_unified.wgsl:310-321): tested by the shader, never accepted.
constexpr uint32_t REF_INVISIBLE = 0xFFFFFFFDu;
constexpr uint32_t CELL_SENTINEL = 0xFFFFFFFFu;  // count of a border cell: the ray has left the grid
constexpr uint32_t BOX_CULLED_GRID = 1u;       // LOD cull outcome when met through a grid cell
constexpr uint32_t BOX_CULLED_PREPASS = 2u;    // LOD cull outcome in the moving-box pre-pass

// Checked build (make -C ray-tracer_amd/csrc checked: -DRT_CHECKED, rt_api.hip only): every index the traversal forms
// is compared with the array it is about to read - reference record against the records built, flattened cell index
// against the bordered cell array, triangle id against the triangle count, staging slot against the LDS buffer. A
// violation is COUNTED (g_check[0]; the first one's kind and operands in [1..3]) and reported by rt_checked_violations;
// it never traps, so a bug found this way costs a failed test, not a GPU fault. The shipped build compiles none of it.
#ifdef RT_CHECKED
__device__ uint32_t g_check[4];
__device__ inline void check_failed(uint32_t kind, uint32_t a, uint32_t b) {
    if (atomicAdd(&g_check[0], 1u) == 0u) g_check[1] = kind, g_check[2] = a, g_check[3] = b;
}
#define RT_CHECK(cond, kind, a, b)                \
    do {                                          \
        if (!(cond)) check_failed((kind), (a), (b)); \
    } while (0)
#else
#define RT_CHECK(cond, kind, a, b) \
    do {                           \
    } while (0)
#endif
constexpr uint32_t CHECK_REF = 1, CHECK_CELL = 2, CHECK_TRIANGLE = 3, CHECK_STAGE = 4, CHECK_DEAL = 5;

struct DevScene {
    const float4* refs;
    const uint2* cells;
    const float4* tri_normal;  // per triangle: normalize(e1 x e2), evaluated once when the scene is uploaded
    const float4* box_frame;
    uint32_t nb, nt, n_cells, n_refs;
    uint32_t gx, gy, gz;
    uint32_t px, pxy;  // strides of the bordered cell array: gx + 2 and (gx + 2) * (gy + 2)
    uint32_t n_bordered;  // its length: (gx + 2) * (gy + 2) * (gz + 2)
    uint32_t max_steps, moving_start;
    float bmin[3], bmax[3];
    float cell_size;
    float inv_cell_size;  // 1 / max(cell_size, 1e-8), evaluated once on the host
    DevTex tex;           // materials and textures, for hits whose acceptance depends on a texture's alpha
};

// Direction-dependent constants of a ray (raytracer_unified.wgsl:494-515). Per lane for primary rays; for shadow
// rays the direction is the same for the whole frame, so the host evaluates them once and they live in scalar
// registers.
struct RayDir {
    V3 d, inv, t_delta;     // direction, safe inverse, |cell_size * inv|
    uint32_t sx, sy, sz;    // +1 / -1 (as 0xFFFFFFFF) per axis
    uint32_t dix, diy, diz; // change of the flattened cell index for a step along each axis
};

// px, pxy: strides of the bordered cell array.
template <bool WGSL>
__host__ __device__ inline RayDir make_ray_dir(V3 d, float cell_size, uint32_t px, uint32_t pxy) {
    RayDir r;
    r.d = d;
    r.inv = v3(safe_inv<WGSL>(d.x), safe_inv<WGSL>(d.y), safe_inv<WGSL>(d.z));
    r.t_delta = vabs(r.inv * cell_size);
    r.sx = d.x >= 0.0f ? 1u : 0xFFFFFFFFu;
    r.sy = d.y >= 0.0f ? 1u : 0xFFFFFFFFu;
    r.sz = d.z >= 0.0f ? 1u : 0xFFFFFFFFu;
    r.dix = r.sx;
    r.diy = r.sy * px;
    r.diz = r.sz * pxy;
    return r;
}

struct DevFrame {
    float cam_pos[3], fwd[3], right[3], up[3];
    float fov_scale;      // tan(fov), evaluated once on the host
    RayDir shadow;        // towards the light: -normalize(light_direction); evaluated once on the host
    float shadow_offset;
    uint32_t width, height;
};

// Frames of one launch (blockIdx.y picks one): consecutive frames of a camera path traced or rendered together, so that
// the long dependent chains of the few expensive pixel blocks of one frame overlap the bulk of the others.
constexpr uint32_t MAX_FRAMES_PER_LAUNCH = 16;
struct DevFrames {
    DevFrame f[MAX_FRAMES_PER_LAUNCH];
    float show_grid[MAX_FRAMES_PER_LAUNCH];
};

struct ShardMap {
    uint32_t tile_w, tile_h, first, stride, count, tiles_x, tiles_y, n_tiles, n_local;
    uint32_t sub_x, sub_y;  // 8x8 sub-tiles per tile
    const uint32_t* offsets;  // owner table given: the `count` places in a round of `stride` tiles this call traces
};

// Frame tile of local tile lt: tile `first` of every round of `stride` tiles, or the tiles at `offsets` of every round.
__device__ inline uint32_t shard_tile(const ShardMap& m, uint32_t lt) {
    if (!m.offsets) return m.first + lt * m.stride;
    const uint32_t round = lt / m.count;
    return round * m.stride + m.offsets[lt - round * m.count];
}

struct Counters {
    uint32_t cells, tri, box, steps;
    // STATS == 2: what this lane asked the memory system for - cell records (8 bytes each), reference records read
    // directly (48 bytes each, the look-ahead record included) and reference records copied into LDS for the wave
    uint32_t cell_recs, ref_recs, staged_recs;
    uint32_t cell_recs_uniform;  // of cell_recs: fetched while every active lane of the wave asked for the same record
    uint32_t dealt_recs;         // trace_ray_repack: reference records this lane read for pairs dealt to it (added to ref_recs)
};

struct HitRec {
    float t;
    uint32_t id;
    uint32_t ref;  // index of the winning reference record (triangles) - used to rebuild the normal
};

// Structured buffer loads (buffer_load_dwordx{2,4} ... idxen): the address is base + index * stride + constant,
// formed by the memory unit from a record index in one VGPR and a descriptor in four SGPRs, so the traversal
// spends no vector instructions on 64-bit address arithmetic, and the compiler cannot split or re-shape the
// loads the way it does plain pointer loads. An index >= num_records reads as zeros.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ v4i llvm_struct_buffer_load_v4i32(v4i rsrc, int vindex, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.struct.buffer.load.v4i32");
__device__ v2i llvm_struct_buffer_load_v2i32(v4i rsrc, int vindex, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.struct.buffer.load.v2i32");

// Scalar buffer load (s_buffer_load_dwordx2): for an offset that is the same in every lane. Read-only use.
__device__ v2i llvm_s_buffer_load_v2i32(v4i rsrc, int offset, int aux) __asm("llvm.amdgcn.s.buffer.load.v2i32");
__device__ int llvm_s_buffer_load_i32(v4i rsrc, int offset, int aux) __asm("llvm.amdgcn.s.buffer.load.i32");

__device__ inline v4i make_rsrc(const void* base, uint32_t stride, uint32_t num_records) {
    const uint64_t b = (uint64_t)base;
    v4i r;
    r.x = (int)(uint32_t)b;
    r.y = (int)(((uint32_t)(b >> 32) & 0xFFFFu) | (stride << 16));
    r.z = (int)num_records;
    r.w = 0x00020000;  // untyped 32-bit data, no swizzle
    return r;
}
template <int BYTE_OFFSET>
__device__ inline float4 buf_load16(v4i rsrc, uint32_t index) {
    const v4i v = llvm_struct_buffer_load_v4i32(rsrc, (int)index, BYTE_OFFSET, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ inline uint2 buf_load8(v4i rsrc, uint32_t index) {
    const v2i v = llvm_struct_buffer_load_v2i32(rsrc, (int)index, 0, 0, 0);
    return make_uint2((uint32_t)v.x, (uint32_t)v.y);
}

// Per-wave LDS buffer for a cell list's reference records (see "LDS staging" in trace_ray): one array per kernel,
// shared by every inlined traversal; kernels run at most four waves per workgroup.
constexpr uint32_t STAGE_RECORDS = 64;
__device__ inline float4* wave_stage() {
    __shared__ float4 staged[4][3 * STAGE_RECORDS];
    return staged[(threadIdx.x >> 6) & 3];
}

// One ray through the grid; mirrors raytracer_unified.wgsl:430-588.
//
// Per-wave cost on gfx950 is set by instruction issue and by the length of the dependent chain of the
// heaviest wave (a few 8x8 pixel blocks over the model's densest cells run 10-30x the average number of
// candidate tests and finish last), not by memory bandwidth: the scene is cache resident. So:
//   - the DDA step is branch-free (selects instead of an if / else-if / else ladder), the flattened cell index
//     is kept incrementally and leaving the grid is detected by a sentinel border around the cell array, not by
//     rebuilding and range-checking cell coordinates,
//   - "have a hit" is best_id != MISS, not a separate flag,
//   - a candidate's 48-byte record is loaded while the previous candidate is tested (loop unrolled by two),
//   - a cell list that the whole wave scans together is staged in LDS once (see scan below),
//   - SKIP: a candidate that the ray already tested in the cell it just left is not tested again. The shader
//     re-tests it, gets the same t, and that t can neither beat the closest hit under its strict '<' nor turn
//     a rejection into a hit, so skipping changes nothing. The per-reference mask in q1.w says through which
//     faces of this cell a ray can arrive having already met the object (see k_ref_masks in rt_build.hip);
//     it is 0 (never skip) for grids supplied by the caller.
// ANY: stop at the first accepted intersection. BOXES: scene has boxes (pre-pass + box candidates).
// STATS: 0 = none. 1 = count what the REFERENCE algorithm does (no skipping), for parity of the counters and for the
// algorithmic byte count. 2 = count what THIS kernel fetches (skipping, look-ahead and LDS staging as shipped): the
// bytes the timed kernels request from the memory system. Returns true on a hit.
// TEX: the scene has materials whose alpha comes from a texture; a reference to such a triangle carries its material
// in the spare word of the third quad, and a geometric hit that would become the closest one is first put to the
// shader's alpha rule (raytracer_unified.wgsl:303-316) at its interpolated uv. (The shader applies the rule inside
// the triangle test, before comparing distances; a discarded or a farther hit changes nothing either way.)
// WIDE: the reference records or the cell array span 4 GiB or more. A buffer descriptor forms its byte offset
// (index * stride) in 32 bits, so such arrays are read through plain 64-bit pointers instead.
// X: 0 = the loop below; TRACE_QUAD (5) = four lanes per ray, see there; other values: development builds only.
constexpr int TRACE_QUAD = 5;

template <bool WGSL, bool ANY, int STATS, bool BOXES, bool TEX = false, bool WIDE = false, int X = 0>
__device__ inline bool trace_ray(const DevScene& s, V3 o, const RayDir& rd, HitRec& out, Counters& ctr, uint32_t quad_step = 1) {
    constexpr bool SKIP = STATS != 1;
    // X == TRACE_QUAD: the instantiation that can run FOUR LANES PER RAY, chosen per wave at run time (quad_step = 4;
    // with quad_step = 1 it is the ordinary loop). The four lanes of a quad (lanes 4r .. 4r+3) then carry the SAME ray
    // and walk the grid in lockstep; a cell's candidate list is dealt to them (lane q tests entries q, q+4, ...), and
    // after the list the quad agrees on one result: the smallest t, and among equal t the earliest list position -
    // which is what scanning the whole list in order with a strict '<' yields (any-hit: the earliest accepted
    // position). Every lane then holds the same closest hit again, so the four take the same decisions at every step.
    // A cell's list costs a quarter of the dependent chain; the steps between cells are done four times over. For
    // pixel blocks known to be expensive. quad_step is the same in every lane of the wave.
    constexpr bool QCAP = X == TRACE_QUAD;
    static_assert(!QCAP || (!BOXES && !TEX && !WIDE && STATS == 0), "four lanes per ray: plain triangle scenes only");
    const uint32_t qstep = QCAP ? quad_step : 1u;
    const uint32_t qlane = QCAP && quad_step != 1u ? (threadIdx.x & 3u) : 0u;
    float closest = 1e10f;
    uint32_t best_id = MISS, best_ref = MISS;
    const V3 d = rd.d, inv = rd.inv;
    const RayPk rp = make_ray_pk(o, d);
    const v4i refs_d = make_rsrc(s.refs, 48, s.n_refs + 1);
    const v4i cells_d = make_rsrc(s.cells, 8, s.n_bordered);
    auto ref_quad = [&](uint32_t r, auto quad) -> float4 {
        constexpr int Q = decltype(quad)::value;
        RT_CHECK(r <= s.n_refs + (QCAP ? 4u : 0u), CHECK_REF, r, s.n_refs);  // one record of padding; the quad form's look-ahead reads zeros
        if constexpr (WIDE)
            return s.refs[3 * (size_t)r + Q];
        else
            return buf_load16<16 * Q>(refs_d, r);
    };
    auto cell_rec = [&](uint32_t idx) -> uint2 {
        if (STATS == 2) {
            ctr.cell_recs++;
            const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
            if (__builtin_amdgcn_ballot_w64(idx == first) == __builtin_amdgcn_ballot_w64(true)) ctr.cell_recs_uniform++;
        }
        RT_CHECK(idx < s.n_bordered, CHECK_CELL, idx, s.n_bordered);
        if constexpr (WIDE)
            return s.cells[idx];
        else
            return buf_load8(cells_d, idx);
    };
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>;
    if (STATS == 1) ctr.steps = 0;

    // moving boxes are tested directly, before the grid (:442-461)
    if (BOXES) {
        for (uint32_t i = s.moving_start; i < s.nb; ++i) {
            if (STATS == 1) ctr.steps++;
            const float4 b0 = s.box_frame[4 * i], b1 = s.box_frame[4 * i + 1];
            if (!(__float_as_uint(b0.w) & BOX_CULLED_PREPASS)) {
                if (STATS == 1) ctr.box++;
                const float t = slab(o, inv, v3(b0.x, b0.y, b0.z), v3(b1.x, b1.y, b1.z));
                if (t >= 0.0f && t < closest) {
                    closest = t;
                    best_id = i;
                    if (ANY) {
                        out = HitRec{closest, best_id, MISS};
                        return true;
                    }
                }
            }
        }
    }

    const V3 bmin = v3(s.bmin[0], s.bmin[1], s.bmin[2]), bmax = v3(s.bmax[0], s.bmax[1], s.bmax[2]);
    const float cs = s.cell_size;

    // entry point (:467-489); a ray that misses the grid reports no hit at all
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

    // DDA set-up (:492-517)
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
    // face through which the current cell was entered, as a bit of the re-test mask (0 = start cell)
    const uint32_t mx = sx == 1u ? 1u : 2u, my = sy == 1u ? 4u : 8u, mz = sz == 1u ? 16u : 32u;
    uint32_t arrival = 0;

    // One candidate record against the ray (:531-556); returns true when the scan of the list must stop
    // (any-hit mode and this candidate was accepted).
    auto test = [&](const float4& c0, const float4& c1, const float4& c2, uint32_t r) -> bool {
        if (SKIP && (__float_as_uint(c1.w) & arrival)) return false;
        const uint32_t id = __float_as_uint(c0.w);
        if (BOXES && id < s.nb) {
            const float4 b0 = s.box_frame[4 * id], b1 = s.box_frame[4 * id + 1];
            float t = -1.0f;
            if (!(__float_as_uint(b0.w) & BOX_CULLED_GRID)) {
                if (STATS == 1) ctr.box++;
                t = slab(o, inv, v3(b0.x, b0.y, b0.z), v3(b1.x, b1.y, b1.z));
            }
            if (t >= 0.0f && t < closest) {
                closest = t;
                best_id = id;
                best_ref = MISS;
                return ANY;
            }
        } else {
            if (STATS == 1 && id != REF_INVALID) ctr.tri++;
            RT_CHECK(id >= REF_INVISIBLE || id - s.nb < s.nt, CHECK_TRIANGLE, id, s.nt);
            const float t = moller_trumbore_pk<WGSL>(rp, c0, c1, c2);
            if (t > 0.0f && t < closest && id < REF_INVISIBLE) {
                if (TEX && __float_as_uint(c2.w) != 0u) {
                    const uint32_t mat = __float_as_uint(c2.w) - 1u;
                    float bu, bv;
                    hit_barycentrics(o, d, v3(c0.x, c0.y, c0.z), v3(c1.x, c1.y, c1.z), v3(c2.x, c2.y, c2.z), bu, bv);
                    const float4 bc = base_color(s.tex, mat, hit_uv(s.tex, id - s.nb, bu, bv));
                    if (alpha_discards(s.tex, mat, bc.w)) return false;
                }
                closest = t;
                best_id = id;
                best_ref = r;
                return ANY;
            }
        }
        return false;
    };

    // A cell's candidate list. Two records per trip, each loaded while the other is tested, so the record in
    // flight never has to be copied into the registers of the record under test. The load one past the list
    // stays inside the array (padded by one record) and its result is not used.
    auto scan_list = [&](uint32_t r, uint32_t n) {
        if constexpr (X != 2) {
            // LDS staging. When every active lane of the wave is about to scan the SAME cell list (rays of an 8x8
            // pixel block mostly are in the same cell), the wave copies the list's records into LDS once - each
            // active lane fetches every n_active-th record - and all lanes read their candidates from there (same
            // address in all lanes: a broadcast read, no bank conflicts) instead of issuing the same three 16-byte
            // loads 64 times over. Order of the tests and their operands are unchanged. (X == 2: development build
            // without it, for comparison.)
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r), n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
            const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
            const bool uniform = __builtin_amdgcn_ballot_w64(r == r0 && n == n0) == act;
            if (uniform) {
                float4* mine = wave_stage();
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
                const uint32_t n_act = (uint32_t)__popcll(act);
                bool done = false;  // any-hit: this lane has accepted a candidate
                for (uint32_t base = 0; base < n0; base += STAGE_RECORDS) {  // lists longer than the buffer: in pieces
                    const uint32_t cnt = n0 - base < STAGE_RECORDS ? n0 - base : STAGE_RECORDS;
                    for (uint32_t k = rank; k < cnt; k += n_act) {
                        if (STATS == 2) ctr.staged_recs++;
                        RT_CHECK(k < STAGE_RECORDS, CHECK_STAGE, k, cnt);
                        mine[3 * k] = ref_quad(r0 + base + k, Q0{});
                        mine[3 * k + 1] = ref_quad(r0 + base + k, Q1{});
                        mine[3 * k + 2] = ref_quad(r0 + base + k, Q2{});
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (!QCAP || qstep == 1u) {  // the loop counter is the same in every lane: a scalar loop
                        for (uint32_t k = 0; k < cnt && !done; ++k)
                            done = test(mine[3 * k], mine[3 * k + 1], mine[3 * k + 2], r0 + base + k);
                    } else {
                        for (uint32_t k = qlane; k < cnt && !done; k += 4)
                            done = test(mine[3 * k], mine[3 * k + 1], mine[3 * k + 2], r0 + base + k);
                    }
                    __builtin_amdgcn_wave_barrier();  // the next piece may overwrite the buffer only after every lane is done
                    if (ANY && __builtin_amdgcn_ballot_w64(!done) == 0ull) break;
                }
                return;
            }
        }
        const uint32_t r_end = r + n;
        if constexpr (QCAP) {
            // this lane's share of the list: r + q, r + q + step, ... (step 1: the whole list, as below). A look-ahead
            // load past the list reads zeros: the buffer descriptor bounds it.
            r += qlane;
            if (r >= r_end) return;
            float4 a0 = ref_quad(r, Q0{}), a1 = ref_quad(r, Q1{}), a2 = ref_quad(r, Q2{});
            for (;;) {
                const float4 b0 = ref_quad(r + qstep, Q0{}), b1 = ref_quad(r + qstep, Q1{}), b2 = ref_quad(r + qstep, Q2{});
                if (test(a0, a1, a2, r)) break;
                if ((r += qstep) >= r_end) break;
                a0 = ref_quad(r + qstep, Q0{}), a1 = ref_quad(r + qstep, Q1{}), a2 = ref_quad(r + qstep, Q2{});
                if (test(b0, b1, b2, r)) break;
                if ((r += qstep) >= r_end) break;
            }
            return;
        }
        float4 a0 = ref_quad(r, Q0{}), a1 = ref_quad(r, Q1{}), a2 = ref_quad(r, Q2{});
        if (STATS == 2) ctr.ref_recs++;
        for (;;) {
            const float4 b0 = ref_quad(r + 1, Q0{}), b1 = ref_quad(r + 1, Q1{}), b2 = ref_quad(r + 1, Q2{});
            if (STATS == 2) ctr.ref_recs++;
            if (test(a0, a1, a2, r)) break;
            if (++r == r_end) break;
            a0 = ref_quad(r + 1, Q0{}), a1 = ref_quad(r + 1, Q1{}), a2 = ref_quad(r + 1, Q2{});
            if (STATS == 2) ctr.ref_recs++;
            if (test(b0, b1, b2, r)) break;
            if (++r == r_end) break;
        }
    };
    auto scan = [&](uint32_t r, uint32_t n) {
        scan_list(r, n);
        if (QCAP && qstep != 1u) {
            // the quad's four partial results -> one: lane ^ 1, then lane ^ 2 (quad permutes: no LDS, no waiting)
#define RT_QUAD_MERGE(CTRL)                                                                                                  \
    {                                                                                                                        \
        const uint32_t t_p = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(closest), CTRL, 0xF, 0xF, false); \
        const uint32_t ref_p = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)best_ref, CTRL, 0xF, 0xF, false);               \
        const uint32_t id_p = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)best_id, CTRL, 0xF, 0xF, false);                 \
        const float tp = __uint_as_float(t_p);                                                                               \
        const bool take = ANY ? (ref_p < best_ref) : (tp < closest || (tp == closest && ref_p < best_ref));                  \
        closest = take ? tp : closest;                                                                                       \
        best_ref = take ? ref_p : best_ref;                                                                                  \
        best_id = take ? id_p : best_id;                                                                                     \
    }
            RT_QUAD_MERGE(0xB1)  // quad_perm [1, 0, 3, 2]
            RT_QUAD_MERGE(0x4E)  // quad_perm [2, 3, 0, 1]
#undef RT_QUAD_MERGE
        }
    };

    if (cx >= s.gx || cy >= s.gy || cz >= s.gz) {
        // Start cell outside the grid: an origin exactly on a far bound (or metadata that disagrees with the
        // bounds). The shader carries on with the out-of-range coordinates: the flattened index
        // x + y*gx + z*gx*gy (:406-409) then names some other cell, or none, until a step brings the ray inside,
        // and a step whose target is outside ends the traversal (:568-583). Rare, so it is restated literally.
        for (uint32_t i = 0; i < s.max_steps; ++i) {
            if (STATS == 1) {
                ctr.steps++;
                ctr.cells++;
            }
            const uint32_t fi = cx + cy * s.gx + cz * s.gx * s.gy;  // wrapping, as in the shader
            uint32_t n = 0, first = 0;
            if (fi < s.n_cells) {
                const uint32_t x = fi % s.gx, y = (fi / s.gx) % s.gy, z = fi / (s.gx * s.gy);
                const uint2 cell = cell_rec((x + 1u) + (y + 1u) * s.px + (z + 1u) * s.pxy);
                first = cell.x;
                n = cell.y;  // an inner cell; counts are stored cut to the cap
            }
            if (n) scan(first, n);
            const bool hit = best_id != MISS;
            if (ANY && hit) break;
            if (hit && closest < fminf(fminf(t_max.x, t_max.y), t_max.z)) break;
            if (t_max.x < t_max.y && t_max.x < t_max.z) {
                const uint32_t next = cx + sx;
                if (next >= s.gx) break;
                cx = next;
                t_max.x += t_delta.x;
            } else if (t_max.y < t_max.z) {
                const uint32_t next = cy + sy;
                if (next >= s.gy) break;
                cy = next;
                t_max.y += t_delta.y;
            } else {
                const uint32_t next = cz + sz;
                if (next >= s.gz) break;
                cz = next;
                t_max.z += t_delta.z;
            }
        }
        out = HitRec{closest, best_id, best_ref};
        return best_id != MISS;
    }

    // traversal (:520-588). The cell array has a one-cell border of sentinel records on every side, so "the
    // step leaves the grid" (:568-583) is "the cell just fetched is a sentinel": no coordinates, no per-axis
    // counters and no range check on the index.
    uint32_t idx = (cx + 1u) + (cy + 1u) * s.px + (cz + 1u) * s.pxy;
    uint2 cell = cell_rec(idx);
    if constexpr (X == 3) {
        // Interleaved form: in every trip of ONE loop a lane either tests the next candidate of its current cell or,
        // when that list is done, leaves the cell (early-out test, step, next cell record). Lanes of a wave that sit
        // in cells with lists of different lengths - or in empty cells - no longer wait for the longest list at every
        // step: a lane in empty space keeps stepping while its neighbours test candidates. Per lane the sequence of
        // tests, comparisons and steps is exactly that of the nested loops below, so the result is the same.
        uint32_t left = s.max_steps;
        uint32_t r = cell.x, r_end = cell.x + cell.y;
        bool alive = left != 0 && cell.y != CELL_SENTINEL;
        while (alive) {
            if (r < r_end) {
                const float4 a0 = ref_quad(r, Q0{}), a1 = ref_quad(r, Q1{}), a2 = ref_quad(r, Q2{});
                if (STATS == 2) ctr.ref_recs++;
                const bool accepted = test(a0, a1, a2, r);
                ++r;
                if (ANY && accepted) alive = false;
            } else {
                const bool ax = t_max.x < t_max.y && t_max.x < t_max.z;
                const bool ay = !ax && t_max.y < t_max.z;
                const bool az = !ax && !ay;
                const float t_exit = fminf(fminf(t_max.x, t_max.y), t_max.z);
                if (STATS == 1) {
                    ctr.steps++;
                    ctr.cells++;
                }
                const bool hit = best_id != MISS;
                --left;
                const bool stop = (ANY & hit) | (hit & (closest < t_exit)) | (left == 0);
                idx += ax ? dix : (ay ? diy : diz);
                arrival = ax ? mx : (ay ? my : mz);
                t_max.x += ax ? t_delta.x : 0.0f;
                t_max.y += ay ? t_delta.y : 0.0f;
                t_max.z += az ? t_delta.z : 0.0f;
                cell = cell_rec(idx);
                alive = !stop && cell.y != CELL_SENTINEL;
                r = cell.x;
                r_end = cell.x + cell.y;
            }
        }
        out = HitRec{closest, best_id, best_ref};
        return best_id != MISS;
    }
    // One exit test per step: leaving the grid, the early-out and the step limit all end the loop through the
    // sentinel count (a stopped ray's next record is overwritten with it). Three separate exits cost a divergent
    // break each - exec-mask bookkeeping in scalar instructions - on every step of every ray; the one record fetched
    // past a stop is a neighbour of an inner cell, so it lies inside the bordered array, and nothing reads it.
    uint32_t left = s.max_steps;
    if (left == 0) cell.y = CELL_SENTINEL;
    while (cell.y != CELL_SENTINEL) {
        // the step this iteration ends with (:566-587: x if tx < ty and tx < tz, else y if ty < tz, else z)
        // depends on t_max only
        const bool ax = t_max.x < t_max.y && t_max.x < t_max.z;
        const bool ay = !ax && t_max.y < t_max.z;
        const bool az = !ax && !ay;
        const float t_exit = fminf(fminf(t_max.x, t_max.y), t_max.z);
        if (STATS == 1) {
            ctr.steps++;
            ctr.cells++;
        }
        const uint32_t n = cell.y;  // stored cut to the cap of raytracer_unified.wgsl:531
        if (n) scan(cell.x, n);
        // any-hit done, or early-out (:561: the hit lies inside this cell), or the step limit (:520); plain '|'
        // and '&': short-circuit forms become branches
        const bool hit = best_id != MISS;
        --left;
        const bool stop = (ANY & hit) | (hit & (closest < t_exit)) | (left == 0);
        idx += ax ? dix : (ay ? diy : diz);
        arrival = ax ? mx : (ay ? my : mz);
        t_max.x += ax ? t_delta.x : 0.0f;
        t_max.y += ay ? t_delta.y : 0.0f;
        t_max.z += az ? t_delta.z : 0.0f;
        if constexpr (X == 4) {
            // Experiment (development builds, or the whole library with -DRT_SCALAR_CELLS): when every lane still in the loop
            // asks for the SAME cell record, one scalar load serves the wave (through the scalar cache, off the vector
            // memory path); lanes in different cells keep the vector load. Bit-identical; measured 1 % slower than the
            // vector load alone on the 1080p frame of the static model (the test per step costs what the load saves), and
            // 4 % slower when the list scan also takes its "same list in every lane" answer from here.
            const uint32_t i0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
            if (__builtin_amdgcn_ballot_w64(idx != i0) == 0ull) {
                const v2i v = llvm_s_buffer_load_v2i32(cells_d, (int)(i0 * 8u), 0);
                cell = make_uint2((uint32_t)v.x, (uint32_t)v.y);
            } else {
                cell = cell_rec(idx);
            }
        } else {
            cell = cell_rec(idx);
        }
        if (stop) cell.y = CELL_SENTINEL;
    }
    out = HitRec{closest, best_id, best_ref};
    return best_id != MISS;
}

// Geometric normal of a hit (triangle: normalize(e1 x e2), :277; box: axis of largest |p|-half, :218-227).
__device__ inline V3 hit_normal(const DevScene& s, const HitRec& h, V3 pos) {
    if (h.ref != MISS) {
        const float4 n = s.tri_normal[h.id - s.nb];
        return v3(n.x, n.y, n.z);
    }
    const float4 c = s.box_frame[4 * h.id + 2], hf = s.box_frame[4 * h.id + 3];
    V3 p = pos - v3(c.x, c.y, c.z);
    V3 dd = vabs(p) - v3(hf.x, hf.y, hf.z);
    if (dd.x > dd.y && dd.x > dd.z) return v3(sign0(p.x), 0.0f, 0.0f);
    if (dd.y > dd.z) return v3(0.0f, sign0(p.y), 0.0f);
    return v3(0.0f, 0.0f, sign0(p.z));
}

// Primary ray direction for a pixel (:634-649).
__device__ inline V3 primary_dir(const DevFrame& f, uint32_t px, uint32_t py) {
    float u = ((float)px + 0.5f) / (float)f.width;
    float v = ((float)py + 0.5f) / (float)f.height;
    float ndc_x = u * 2.0f - 1.0f;
    float ndc_y = v * 2.0f - 1.0f;
    float aspect = (float)f.width / (float)f.height;
    V3 a = v3(f.fwd[0], f.fwd[1], f.fwd[2]);
    V3 b = ((v3(f.right[0], f.right[1], f.right[2]) * ndc_x) * aspect) * f.fov_scale;
    V3 c = (v3(f.up[0], f.up[1], f.up[2]) * (-ndc_y)) * f.fov_scale;
    return normalize((a + b) + c);
}

// Wave-level sum of per-lane counters, one atomic per wave per counter.
__device__ inline void flush_stats(unsigned long long* stats, unsigned long long rays, unsigned long long phits,
                                   unsigned long long srays, unsigned long long shits, const Counters& c) {
    unsigned long long v[11] = {rays, phits, srays, shits, c.cells, c.tri, c.box, c.cell_recs, c.ref_recs, c.staged_recs, c.cell_recs_uniform};
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&stats[k], x);
    }
}

// One lane per pixel, one wave per 8x8 pixel block, four waves (a 16x16 pixel block) per workgroup.
// Workgroups are dealt to the XCDs round-robin in launch order, so neighbouring 16x16 blocks run on
// different XCDs at the same time: cheap (sky) and expensive (dense geometry) image regions are spread
// evenly over the chip. (Giving each XCD one contiguous band of the image instead left most XCDs idle
// behind the one that owned the geometry: 30 % average occupancy on the headline frame.)
// The first two arguments of the primary-ray kernels as the kernel-argument segment lays them out.
struct PrimaryArgs {
    DevScene s;
    DevFrame f;
    ShardMap m;
    uint2* out_p;
    uint2* out_s;
    const uint32_t* cull_mask;  // k_trace_primary_lean only: one bit per 8x8 pixel block of the frame, or null
    uint32_t* cull_zero;        // the mask buffer of the next frame on this stream: zeroed by this launch (or null)
    uint32_t cull_zero_words;
};

// The arguments of k_trace_primary_frames as the kernel-argument segment lays them out.
struct PrimaryFramesArgs {
    DevScene s;
    DevFrames fs;
    ShardMap m;
    uint2* out_p;
    uint2* out_s;
    const uint32_t* cull_mask;  // the frames' block masks, cull_words words apart, or null
    uint32_t* cull_zero;
    uint32_t cull_zero_words;
    uint32_t cull_words;
    unsigned long long frame_stride;  // hit records between the frames of an output array
};

// A word / a pointer of the kernel-argument segment, read where it is needed (see late_shadow_dir).
__device__ inline uint32_t late_arg_u32(uint32_t byte_offset) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) uint8_t* KernArg;
    KernArg base = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));
    return *reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(base + byte_offset);
#else
    return byte_offset;
#endif
}
__device__ inline const uint32_t* late_arg_cptr(uint32_t byte_offset) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) uint8_t* KernArg;
    KernArg base = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
    return *reinterpret_cast<const uint32_t* const __attribute__((address_space(4)))*>(base + byte_offset);
#else
    return nullptr;
#endif
}
__device__ inline unsigned long long late_arg_u64(uint32_t byte_offset) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) uint8_t* KernArg;
    KernArg base = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));
    return *reinterpret_cast<const __attribute__((address_space(4))) unsigned long long*>(base + byte_offset);
#else
    return byte_offset;
#endif
}
__device__ inline uint2* late_arg_ptr(uint32_t byte_offset) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) uint8_t* KernArg;
    KernArg base = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));
    return *reinterpret_cast<uint2* const __attribute__((address_space(4)))*>(base + byte_offset);
#else
    return nullptr;
#endif
}

// The shadow ray's direction record, read from the kernel-argument segment where it is first needed instead of at
// the start of the kernel: fifteen scalar registers that would otherwise sit through the whole primary traversal,
// and push the kernel over the 80 scalar registers that eight waves per SIMD leave each wave.
__device__ inline RayDir late_shadow_dir(uint32_t frame_byte_offset) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) uint8_t* KernArg;
    KernArg base = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));  // the loads below stay below
    const __attribute__((address_space(4))) uint32_t* w = reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(
        base + frame_byte_offset + offsetof(DevFrame, shadow));
    RayDir r;
    r.d = v3(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]));
    r.inv = v3(__uint_as_float(w[3]), __uint_as_float(w[4]), __uint_as_float(w[5]));
    r.t_delta = v3(__uint_as_float(w[6]), __uint_as_float(w[7]), __uint_as_float(w[8]));
    r.sx = w[9], r.sy = w[10], r.sz = w[11];
    r.dix = w[12], r.diy = w[13], r.diz = w[14];
    return r;
#else
    return RayDir{};  // the host pass of the compiler only parses this
#endif
}
static_assert(sizeof(RayDir) == 60 && offsetof(RayDir, sx) == 36 && offsetof(RayDir, dix) == 48, "late_shadow_dir reads RayDir word by word");
// Kernels that read arguments late (k_trace_primary_lean, k_trace_primary_frames, k_trace_primary_planned,
// k_trace_primary_queue) take (DevScene, DevFrame or DevFrames, ...) as their first arguments; the kernel-argument
// segment lays arguments out in order, each at its natural alignment, which is what offsetof() of the *Args structs
// (same members, same order) evaluates to. Reordering or inserting an argument in one of those kernels means changing
// its struct too.
static_assert(offsetof(PrimaryArgs, s) == 0 && offsetof(PrimaryArgs, f) == (sizeof(DevScene) + alignof(DevFrame) - 1) / alignof(DevFrame) * alignof(DevFrame) &&
                  offsetof(PrimaryFramesArgs, fs) == offsetof(PrimaryArgs, f),
              "late argument reads assume the scene first, the frame(s) right behind it");

#ifdef RT_EXPERIMENTS
__device__ unsigned long long g_wave_t[2 * 65536];  // development builds: s_memrealtime (100 MHz) at wave start / end,
__device__ uint32_t g_wave_chain[65536];            // and the cost a block reported (bits 30-31: its class)
#endif

// LATE: the kernel's arguments are exactly PrimaryArgs' members in that order, and what the epilogue needs of them is
// read again there instead of being kept in scalar registers through both traversals.
// FRAMES (with LATE): the kernel's arguments are PrimaryFramesArgs', blockIdx.y picks the frame.
template <bool WGSL, bool SHADOW, int STATS, bool BOXES, bool TEX, bool WIDE, bool LATE = false, bool FRAMES = false>
__device__ inline void trace_primary_block(const DevScene& s, const DevFrame& f, const ShardMap& m, uint2* __restrict__ out_p,
                                           uint2* __restrict__ out_s, unsigned long long* stats, const uint32_t* cull_mask = nullptr) {
    static_assert(!FRAMES || LATE, "the frames kernel reads its arguments late");
#ifdef RT_EXPERIMENTS
    const unsigned long long t_wave_start = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr uint32_t A_M = FRAMES ? offsetof(PrimaryFramesArgs, m) : offsetof(PrimaryArgs, m);
    constexpr uint32_t A_OUT_P = FRAMES ? offsetof(PrimaryFramesArgs, out_p) : offsetof(PrimaryArgs, out_p);
    constexpr uint32_t A_OUT_S = FRAMES ? offsetof(PrimaryFramesArgs, out_s) : offsetof(PrimaryArgs, out_s);
    constexpr uint32_t A_MASK = FRAMES ? offsetof(PrimaryFramesArgs, cull_mask) : offsetof(PrimaryArgs, cull_mask);
    const uint32_t frame_no = FRAMES ? blockIdx.y : 0u;
    const uint32_t a_frame = FRAMES ? (uint32_t)offsetof(PrimaryFramesArgs, fs) + frame_no * (uint32_t)sizeof(DevFrame)
                                    : (uint32_t)offsetof(PrimaryArgs, f);
    const uint32_t lane = threadIdx.x & 63, w4 = threadIdx.x >> 6;
    const uint32_t blk_x = (m.sub_x + 1) / 2, blk_y = (m.sub_y + 1) / 2;  // 16x16 blocks per tile
    const uint32_t blk_per_tile = blk_x * blk_y;
    const uint32_t lt = blockIdx.x / blk_per_tile, blk = blockIdx.x % blk_per_tile;
    if (lt >= m.n_local) return;  // whole workgroup
    const uint32_t sub_x = (blk % blk_x) * 2 + (w4 & 1), sub_y = (blk / blk_x) * 2 + (w4 >> 1);
    const uint32_t lx = sub_x * 8 + (lane & 7), ly = sub_y * 8 + (lane >> 3);
    const uint32_t tile = shard_tile(m, lt);
    const uint32_t px = (tile % m.tiles_x) * m.tile_w + lx, py = (tile / m.tiles_x) * m.tile_h + ly;
    const bool in_tile = lx < m.tile_w && ly < m.tile_h;
    const bool in_frame = in_tile && px < f.width && py < f.height;
    bool live = in_frame;
    {
        // A clear bit in the frame's block mask (k_block_mask in rt_cull.hpp): no primary ray of this wave's 8x8 pixel
        // block meets an occupied cell, so all 64 records are misses. Tiles are whole multiples of 8 pixels when a mask
        // is given, so the wave's block is one block of the frame's 8x8 raster; everything here is the same in every lane.
        const uint32_t* mask = LATE ? late_arg_cptr(A_MASK) : cull_mask;
        if (FRAMES && mask) mask += (size_t)frame_no * late_arg_u32(offsetof(PrimaryFramesArgs, cull_words));
        if (mask) {
            const uint32_t bx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((tile % m.tiles_x) * m.tile_w >> 3) + sub_x));
            const uint32_t by = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((tile / m.tiles_x) * m.tile_h >> 3) + sub_y));
            const uint32_t blocks_x = (f.width + 7u) >> 3, blocks_y = (f.height + 7u) >> 3;
            if (bx < blocks_x && by < blocks_y) {
                const uint32_t word = mask[(size_t)by * ((blocks_x + 31u) >> 5) + (bx >> 5)];
                if (((word >> (bx & 31u)) & 1u) == 0u) live = false;
            }
        }
    }

    uint2 ph = make_uint2(MISS, 0u), sh = make_uint2(MISS, 0u);
    Counters ctr{};
    unsigned long long n_rays = 0, n_ph = 0, n_sr = 0, n_sh = 0;
    if (STATS && in_frame && !live) n_rays++;  // a ray answered by the block mask is still a ray of the frame
    if (live) {
        const V3 o = v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
        const V3 d = primary_dir(f, px, py);
        const RayDir rd = make_ray_dir<WGSL>(d, s.cell_size, s.px, s.pxy);
        HitRec h;
        n_rays++;
#ifdef RT_SCALAR_CELLS
        constexpr int XB = LATE && !FRAMES ? 4 : 0;
#else
        constexpr int XB = 0;
#endif
        if (trace_ray<WGSL, false, STATS, BOXES, TEX, WIDE, XB>(s, o, rd, h, ctr)) {
            n_ph++;
            ph = make_uint2(h.id, __float_as_uint(h.t));
            if (SHADOW) {
                const V3 pos = o + d * h.t;
                V3 n = hit_normal(s, h, pos);
                if (dot(n, d) > 0.0f) n = neg(n);
                const V3 so = pos + n * f.shadow_offset;
                HitRec oh;
                n_rays++;
                n_sr++;
                // LATE: the kernel's leading arguments are PrimaryArgs' (checked where the kernels are declared)
                const RayDir to_light = LATE ? late_shadow_dir(a_frame) : f.shadow;
                if (trace_ray<WGSL, true, STATS, BOXES, TEX, WIDE, XB>(s, so, to_light, oh, ctr)) {
                    n_sh++;
                    sh = make_uint2(oh.id, __float_as_uint(oh.t));
                }
            }
        }
    }
    if (in_tile) {
        if (LATE) {
            const uint32_t tw = late_arg_u32(A_M + offsetof(ShardMap, tile_w));
            const uint32_t th = late_arg_u32(A_M + offsetof(ShardMap, tile_h));
            size_t oi = (size_t)lt * tw * th + (size_t)ly * tw + lx;
            if (FRAMES) oi += (size_t)frame_no * late_arg_u64(offsetof(PrimaryFramesArgs, frame_stride));
            late_arg_ptr(A_OUT_P)[oi] = ph;
            if (SHADOW) late_arg_ptr(A_OUT_S)[oi] = sh;
        } else {
            const size_t oi = (size_t)lt * m.tile_w * m.tile_h + (size_t)ly * m.tile_w + lx;
            out_p[oi] = ph;
            if (SHADOW) out_s[oi] = sh;
        }
    }
    if (STATS) flush_stats(stats, n_rays, n_ph, n_sr, n_sh, ctr);
#ifdef RT_EXPERIMENTS
    if (LATE && !FRAMES && lane == 0) {  // development builds: when this wave ran (tools/wave_timeline.py)
        const uint32_t gw = blockIdx.x * 4 + w4;
        if (gw < 65536) {
            g_wave_t[2 * gw] = t_wave_start;
            g_wave_t[2 * gw + 1] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

template <bool WGSL, bool SHADOW, int STATS, bool BOXES, bool TEX = false, bool WIDE = false>
__global__ __launch_bounds__(256) void k_trace_primary(DevScene s, DevFrame f, ShardMap m, uint2* __restrict__ out_p,
                                                       uint2* __restrict__ out_s, unsigned long long* stats,
                                                       const uint32_t* __restrict__ cull_mask) {
    trace_primary_block<WGSL, SHADOW, STATS, BOXES, TEX, WIDE>(s, f, m, out_p, out_s, stats, cull_mask);
}

// k_trace_primary for triangle-only scenes without counters (the common case) inside 80 scalar registers: eight waves
// per SIMD instead of seven.
template <bool WGSL, bool SHADOW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_trace_primary_lean(
    DevScene s, DevFrame f, ShardMap m, uint2* __restrict__ out_p, uint2* __restrict__ out_s, const uint32_t* __restrict__ cull_mask,
    uint32_t* __restrict__ cull_zero, uint32_t cull_zero_words) {
    if (cull_zero)
        for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < cull_zero_words; k += gridDim.x * 256u) cull_zero[k] = 0u;
    trace_primary_block<WGSL, SHADOW, 0, false, false, false, true>(s, f, m, out_p, out_s, nullptr);
}

// The same for up to MAX_FRAMES_PER_LAUNCH frames of a camera path in one launch (blockIdx.y = frame): one launch and
// one mask launch for all of them, and the tail of a frame overlaps the bulk of the next without a second stream.
template <bool WGSL, bool SHADOW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_trace_primary_frames(
    DevScene s, DevFrames fs, ShardMap m, uint2* __restrict__ out_p, uint2* __restrict__ out_s, const uint32_t* __restrict__ cull_mask,
    uint32_t* __restrict__ cull_zero, uint32_t cull_zero_words, uint32_t cull_words, unsigned long long frame_stride) {
    if (cull_zero)
        for (uint32_t k = (blockIdx.y * gridDim.x + blockIdx.x) * 256u + threadIdx.x; k < cull_zero_words; k += gridDim.x * gridDim.y * 256u)
            cull_zero[k] = 0u;
    trace_primary_block<WGSL, SHADOW, 0, false, false, false, true, true>(s, fs.f[blockIdx.y], m, out_p, out_s, nullptr);
}

// Persistent waves with a block queue (rt_params.persistent_waves): as many workgroups as the GPU holds at once, each
// workgroup takes QUEUE_CHUNK consecutive 8x8 pixel blocks (numbered as k_trace_primary_lean's waves: the four blocks of
// a 16x16 pixel square follow each other, one per wave) from an atomic counter until none is left. Every workgroup
// reaches the exit test after finitely many blocks, and all its waves take it together, so the grid always drains.
//
// The block body sits in a loop here, and what a loop carries costs registers: the compiler hoists everything that
// does not depend on the pixel out of it - the kernel's arguments, and the per-frame float arithmetic (bounds relative
// to the camera, aspect ratio, reciprocals: uniform values, but float results live in VECTOR registers) - 91 vector
// registers and a hundred spilled scalar ones, five waves per SIMD. So nothing is carried: every turn re-reads the
// kernel's arguments from the kernel-argument segment through a pointer the compiler cannot see through (scalar loads
// that hit the constant cache), and the body compiles as it does in k_trace_primary_lean plus the loop's own state: 71
// vector registers, no scratch, seven waves per SIMD.
struct QueueArgs {  // the kernel's one argument = the kernel-argument segment
    DevScene s;
    DevFrame f;
    ShardMap m;
    uint2* out_p;
    uint2* out_s;
    const uint32_t* cull_mask;  // the frame's block mask, or null
    uint32_t* queue;            // next chunk; zero when the launch starts, and zero again when its last workgroup has drawn
    uint32_t n_waves;           // 8x8 blocks of the launch: four per workgroup of the plain grid
    uint32_t* cull_zero;        // the mask buffer of the next frame on this stream: zeroed by this launch (or null)
    uint32_t cull_zero_words;
};
constexpr uint32_t QUEUE_CHUNK = 16;  // 8x8 blocks a WORKGROUP takes per turn: four 16x16 pixel squares, a quadrant of each per wave

template <class T>
__device__ inline void load_kernarg(T& out, const __attribute__((address_space(4))) T* p) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    const __attribute__((address_space(4))) uint32_t* w = (const __attribute__((address_space(4))) uint32_t*)p;
    uint32_t* o = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(T) / 4; ++i) o[i] = w[i];
}

template <bool WGSL, bool SHADOW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_trace_primary_queue(QueueArgs in_kernarg_segment) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) QueueArgs* KernArgs;
    // One atomic per workgroup and turn (same-address atomics complete at about one per 15 ns on this chip: one per wave
    // and four blocks made a 1080p frame four times slower than the plain grid). Thread 0 draws, the others read the
    // ticket from LDS after the barrier; two slots used in turn, so one barrier per turn is enough: the slot a turn
    // writes was last read before the previous turn's barrier.
    __shared__ uint32_t ticket[2];
    {
        typedef const __attribute__((address_space(4))) QueueArgs* KA;
        KA ka0 = (KA)__builtin_amdgcn_kernarg_segment_ptr();
        if (uint32_t* zero = ka0->cull_zero)
            for (uint32_t k = blockIdx.x * 256u + threadIdx.x, n = ka0->cull_zero_words; k < n; k += gridDim.x * 256u) zero[k] = 0u;
    }
    for (uint32_t turn = 0;; ++turn) {
        KernArgs ka = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));  // a pointer the compiler knows nothing about: what is read through it is read now
        const uint32_t n_waves = ka->n_waves;
        if (threadIdx.x == 0) ticket[turn & 1u] = atomicAdd(ka->queue, 1u);
        __syncthreads();
        const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket[turn & 1u]);
        const uint32_t n_chunks = (n_waves + QUEUE_CHUNK - 1) / QUEUE_CHUNK;
        if (c >= n_chunks) {  // the same in every wave of the workgroup
            // Every workgroup draws exactly one ticket beyond the chunks; the one that draws the last of those leaves the
            // counter at zero for the next launch on this stream (no other workgroup touches it any more).
            if (threadIdx.x == 0 && c == n_chunks + gridDim.x - 1u) *ka->queue = 0u;
            break;
        }
        for (uint32_t j = 0; j < QUEUE_CHUNK / 4; ++j) {
            const uint32_t gw = c * QUEUE_CHUNK + 4u * j + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            if (gw >= n_waves) break;
            asm volatile("" : "+s"(ka));
            DevScene s;
            DevFrame f;
            ShardMap m;
            load_kernarg(s, &ka->s);
            load_kernarg(f, &ka->f);
            load_kernarg(m, &ka->m);
            const uint32_t lane = threadIdx.x & 63;
            const uint32_t w4 = gw & 3, b16 = gw >> 2;
            const uint32_t blk_x = (m.sub_x + 1) / 2, blk_y = (m.sub_y + 1) / 2;  // 16x16 blocks per tile
            const uint32_t blk_per_tile = blk_x * blk_y;
            const uint32_t lt = b16 / blk_per_tile, blk = b16 % blk_per_tile;
            if (lt >= m.n_local) continue;
            const uint32_t sub_x = (blk % blk_x) * 2 + (w4 & 1), sub_y = (blk / blk_x) * 2 + (w4 >> 1);
            const uint32_t lx = sub_x * 8 + (lane & 7), ly = sub_y * 8 + (lane >> 3);
            const uint32_t tile = shard_tile(m, lt);
            const uint32_t px = (tile % m.tiles_x) * m.tile_w + lx, py = (tile / m.tiles_x) * m.tile_h + ly;
            const bool in_tile = lx < m.tile_w && ly < m.tile_h;
            bool live = in_tile && px < f.width && py < f.height;
            if (const uint32_t* mask = ka->cull_mask) {  // as in trace_primary_block: a clear bit = 64 miss records
                const uint32_t bx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((tile % m.tiles_x) * m.tile_w >> 3) + sub_x));
                const uint32_t by = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((tile / m.tiles_x) * m.tile_h >> 3) + sub_y));
                const uint32_t blocks_x = (f.width + 7u) >> 3, blocks_y = (f.height + 7u) >> 3;
                if (bx < blocks_x && by < blocks_y) {
                    const uint32_t word = mask[(size_t)by * ((blocks_x + 31u) >> 5) + (bx >> 5)];
                    if (((word >> (bx & 31u)) & 1u) == 0u) live = false;
                }
            }
            uint2 ph = make_uint2(MISS, 0u), sh = make_uint2(MISS, 0u);
            if (live) {
                Counters ctr{};
                const V3 o = v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
                const V3 d = primary_dir(f, px, py);
                const RayDir rd = make_ray_dir<WGSL>(d, s.cell_size, s.px, s.pxy);
                HitRec h;
                if (trace_ray<WGSL, false, 0, false, false, false>(s, o, rd, h, ctr)) {
                    ph = make_uint2(h.id, __float_as_uint(h.t));
                    if (SHADOW) {
                        const V3 pos = o + d * h.t;
                        V3 n = hit_normal(s, h, pos);
                        if (dot(n, d) > 0.0f) n = neg(n);
                        const V3 so = pos + n * f.shadow_offset;
                        HitRec oh;
                        const RayDir to_light = late_shadow_dir((uint32_t)offsetof(QueueArgs, f));
                        if (trace_ray<WGSL, true, 0, false, false, false>(s, so, to_light, oh, ctr)) sh = make_uint2(oh.id, __float_as_uint(oh.t));
                    }
                }
            }
            if (in_tile) {
                const size_t oi = (size_t)lt * m.tile_w * m.tile_h + (size_t)ly * m.tile_w + lx;
                ka->out_p[oi] = ph;
                if (SHADOW) ka->out_s[oi] = sh;
            }
        }
    }
#endif
}

// trace_ray for waves whose rays sit in DIFFERENT cells with lists of different lengths (rt_params.repack_candidates):
// (ray, candidate) pairs are dealt out again to the lanes of the wave at DDA steps where that keeps more lanes busy. All 64
// lanes run one wave-uniform loop, whether their own ray is still alive or not. Per step: a prefix sum of the alive
// lanes' list lengths numbers the wave's pairs 0 .. total-1; pair p goes to lane p % 64 in round p / 64, which finds its
// ray (a binary search over the prefix sums in LDS), fetches that ray and its re-test mask from LDS, the candidate's
// record from memory, and runs the ordinary test. Results meet in one 64-bit LDS word per ray through an unsigned
// minimum: closest hit (t bits, 1 + list position), which starts as (closest so far, 0) - a positive float's bits order
// as the float does, so the minimum is the smallest t, among equal t the earliest position, and a candidate that only
// equals the closest so far loses, which is what scanning the list in order with a strict '<' yields; any-hit
// (1 + list position, t bits): the earliest accepted position. Rays that start outside the grid take the ordinary
// function afterwards. Triangle-only scenes, no counters, no texture-decided hits, reference records within 4 GiB.
struct RepackLds {
    unsigned long long key[64];
    uint32_t excl[64];
    float4 o_arrival[64];  // the ray's origin and its re-test mask
    float4 d_first[64];    // its direction and the first record of its current cell's list
};

// inclusive prefix sum over the wave's 64 lanes: four shifts inside each row of 16, then the row totals across
__device__ inline uint32_t wave_prefix_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);   // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return x;
}

// the largest x of the wave's 64 lanes (valid in lane 63; read it with readlane)
__device__ inline uint32_t wave_running_max(uint32_t x) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false));
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false));
    return x;
}

// MODE 0: pairs are dealt out at every step. MODE 1 / 2: only at steps where that is estimated to cost less than every
// lane scanning its own list (rounds of 64 pairs against the longest list of the wave, weighed by what a round and a
// trip of the plain loop cost in instructions; 2 deals less readily: measured, slower, as is a more eager rule), the
// plain scan otherwise.
// COUNT: also count what the lanes ask the memory system for (Counters::cell_recs, ref_recs, as trace_ray<STATS = 2> does
// for the plain loop): a cell record per step of a ray that is still followed, a reference record per look-ahead load of
// the plain scan and one per dealt pair.
// Timing probe (experiment builds, make alt ALT=-DRT_PROBE_EXTRA_LOAD): one more 8-byte load of every candidate record,
// from the cache line already asked for, its result unused: what eight bytes per record cost on the address path.
#ifdef RT_PROBE_EXTRA_LOAD
#define RT_PROBE_LOAD(R)                                             \
    {                                                                \
        const v2i probe_x = llvm_struct_buffer_load_v2i32(refs_d, (int)(R), 8, 0, 0); \
        asm volatile("" ::"v"(probe_x.x), "v"(probe_x.y));           \
    }
#else
#define RT_PROBE_LOAD(R)
#endif
// TRIP: what a trip of the plain scan is taken to cost against 130 for a round of 64 dealt pairs (MODE 1). 55 for the
// rays of a frame; rays in no order (k_trace_rays_repack) deal more readily, 75: 4 M random rays 2.55 -> 2.25 ms, where
// the same setting costs the frames of the bundled animated model a fifth (155 -> 191 us) and the streaming frame 1 %.
#ifndef RT_DEAL_TRIP_FRAMES
#define RT_DEAL_TRIP_FRAMES 55u
#endif
template <bool WGSL, bool ANY, int MODE = 1, bool TEX = false, bool COUNT = false, unsigned TRIP = RT_DEAL_TRIP_FRAMES>
__device__ inline bool trace_ray_repack(const DevScene& s, V3 o, const RayDir& rd, bool active, HitRec& out, RepackLds& L,
                                        Counters* ctr = nullptr) {
    const uint32_t lane = threadIdx.x & 63;
    float closest = 1e10f;
    uint32_t best_ref = MISS;
    const V3 d = rd.d, inv = rd.inv;
    const v4i refs_d = make_rsrc(s.refs, 48, s.n_refs + 1);
    const v4i cells_d = make_rsrc(s.cells, 8, s.n_bordered);
    const V3 bmin = v3(s.bmin[0], s.bmin[1], s.bmin[2]), bmax = v3(s.bmax[0], s.bmax[1], s.bmax[2]);
    const float cs = s.cell_size;
    bool alive = active;
    V3 pos = o;
    float t_off = 0.0f;
    if (pos.x < bmin.x || pos.x > bmax.x || pos.y < bmin.y || pos.y > bmax.y || pos.z < bmin.z || pos.z > bmax.z) {
        const float t_entry = slab(o, inv, bmin, bmax);
        if (t_entry < 0.0f) alive = false;
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
    const bool literal = alive && (cx >= s.gx || cy >= s.gy || cz >= s.gz);  // start cell outside the grid
    if (literal) alive = false;
    const uint32_t sx = rd.sx, sy = rd.sy, sz = rd.sz, dix = rd.dix, diy = rd.diy, diz = rd.diz;
    const V3 cell_world = bmin + v3((float)cx, (float)cy, (float)cz) * cs;
    const V3 next_b = cell_world + v3(sx == 1u ? cs : 0.0f, sy == 1u ? cs : 0.0f, sz == 1u ? cs : 0.0f);
    const V3 t_delta = rd.t_delta;
    V3 t_max = v3(t_off, t_off, t_off) + (next_b - pos) * inv;
    {
        const float fl = t_off + 0.00001f;
        t_max = vmax(t_max, v3(fl, fl, fl));
    }
    const uint32_t mx = sx == 1u ? 1u : 2u, my = sy == 1u ? 4u : 8u, mz = sz == 1u ? 16u : 32u;
    uint32_t arrival = 0;
    uint32_t idx = alive ? (cx + 1u) + (cy + 1u) * s.px + (cz + 1u) * s.pxy : 0u;  // 0: a border cell, a sentinel
    uint2 cell = buf_load8(cells_d, idx);
    if (COUNT && alive) ctr->cell_recs++;
    uint32_t left = s.max_steps;
    alive = alive && left != 0 && cell.y != CELL_SENTINEL;
    while (__builtin_amdgcn_ballot_w64(alive) != 0ull) {  // the same in every lane
        // the step this trip ends with depends on t_max only: the next cell's record is asked for before the pairs
        const bool ax = t_max.x < t_max.y && t_max.x < t_max.z;
        const bool ay = !ax && t_max.y < t_max.z;
        const bool az = !ax && !ay;
        const float t_exit = fminf(fminf(t_max.x, t_max.y), t_max.z);
        const uint32_t idx_next = alive ? idx + (ax ? dix : (ay ? diy : diz)) : 0u;
        RT_CHECK(idx_next < s.n_bordered, CHECK_CELL, idx_next, s.n_bordered);
        const uint2 cell_next = buf_load8(cells_d, idx_next);
        if (COUNT && alive) ctr->cell_recs++;
        const uint32_t n = alive ? cell.y : 0u;
        const uint32_t incl = wave_prefix_sum(n);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        bool deal = total != 0;
        if (MODE != 0 && deal) {
            const uint32_t longest = (uint32_t)__builtin_amdgcn_readlane((int)wave_running_max(n), 63);
            const uint32_t rounds = (total + 63u) >> 6;
            deal = MODE == 1 ? rounds * 130u + 50u < longest * TRIP : rounds * 180u + 100u < longest * 55u;
        }
        if (!deal) {
            // One list shared by every lane that has one (rays that are still in the same cell: near the camera, or the
            // whole way on scenes whose neighbouring rays stay together): the WAVE copies it into LDS once, all 64 lanes
            // fetching a record each, and the lanes scan it from there in order - three load instructions per 64
            // records instead of three per record. The words the dealing branch uses serve as the buffer.
            const unsigned long long has = __builtin_amdgcn_ballot_w64(n != 0);
            bool shared = false;
            uint32_t r0 = 0, n0 = 0;
            if (has != 0ull) {
                const int first = __builtin_ctzll(has);
                r0 = (uint32_t)__builtin_amdgcn_readlane((int)cell.x, first);
                n0 = (uint32_t)__builtin_amdgcn_readlane((int)n, first);
                shared = __builtin_amdgcn_ballot_w64(n != 0 && cell.x == r0 && n == n0) == has;
            }
            auto test = [&](const float4& c0, const float4& c1, const float4& c2, uint32_t r) -> bool {
                    if (__float_as_uint(c1.w) & arrival) return false;
                    const float t = moller_trumbore_pk<WGSL>(make_ray_pk(o, d), c0, c1, c2);
                    if (t > 0.0f && t < closest && __float_as_uint(c0.w) < REF_INVISIBLE) {
                        if (TEX && __float_as_uint(c2.w) != 0u) {  // the hit's alpha is a texture's: sampled per hit
                            const uint32_t mat = __float_as_uint(c2.w) - 1u;
                            float bu, bv;
                            hit_barycentrics(o, d, v3(c0.x, c0.y, c0.z), v3(c1.x, c1.y, c1.z), v3(c2.x, c2.y, c2.z), bu, bv);
                            const float4 bc = base_color(s.tex, mat, hit_uv(s.tex, __float_as_uint(c0.w) - s.nb, bu, bv));
                            if (alpha_discards(s.tex, mat, bc.w)) return false;
                        }
                        closest = t;
                        best_ref = r;
                        return ANY;
                    }
                    return false;
            };
            if (shared) {
                float4* stage = reinterpret_cast<float4*>(&L);
                constexpr uint32_t PIECE = sizeof(RepackLds) / 48 < 64 ? sizeof(RepackLds) / 48 : 64;  // records per copy
                bool done = n == 0;  // lanes without the list only help to copy it
                for (uint32_t base = 0; base < n0; base += PIECE) {
                    const uint32_t cnt = n0 - base < PIECE ? n0 - base : PIECE;
                    if (lane < cnt) {
                        stage[3 * lane] = buf_load16<0>(refs_d, r0 + base + lane);
                        stage[3 * lane + 1] = buf_load16<16>(refs_d, r0 + base + lane);
                        stage[3 * lane + 2] = buf_load16<32>(refs_d, r0 + base + lane);
                        if (COUNT) ctr->staged_recs++;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    for (uint32_t k = 0; k < cnt && !done; ++k) done = test(stage[3 * k], stage[3 * k + 1], stage[3 * k + 2], r0 + base + k);
                    __builtin_amdgcn_wave_barrier();  // the next piece, or the next step's dealing, overwrites the buffer
                    if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
                }
            } else if (n != 0) {  // this lane's own list, two records per trip, each loaded while the other is tested
                uint32_t r = cell.x;
                const uint32_t r_end = r + n;
                float4 a0 = buf_load16<0>(refs_d, r), a1 = buf_load16<16>(refs_d, r), a2 = buf_load16<32>(refs_d, r);
                if (COUNT) ctr->ref_recs++;
                for (;;) {
                    const float4 b0 = buf_load16<0>(refs_d, r + 1), b1 = buf_load16<16>(refs_d, r + 1), b2 = buf_load16<32>(refs_d, r + 1);
                    RT_PROBE_LOAD(r + 1)
                    if (COUNT) ctr->ref_recs++;
                    if (test(a0, a1, a2, r)) break;
                    if (++r == r_end) break;
                    a0 = buf_load16<0>(refs_d, r + 1), a1 = buf_load16<16>(refs_d, r + 1), a2 = buf_load16<32>(refs_d, r + 1);
                    RT_PROBE_LOAD(r + 1)
                    if (COUNT) ctr->ref_recs++;
                    if (test(b0, b1, b2, r)) break;
                    if (++r == r_end) break;
                }
            }
        } else {
            L.excl[lane] = incl - n;
            L.o_arrival[lane] = make_float4(o.x, o.y, o.z, __uint_as_float(arrival));
            L.d_first[lane] = make_float4(d.x, d.y, d.z, __uint_as_float(cell.x));
            L.key[lane] = ANY ? ~0ull : ((unsigned long long)__float_as_uint(closest) << 32);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t p0 = 0; p0 < total; p0 += 64) {
                const uint32_t p = p0 + lane;
                if (p < total) {
                    uint32_t j = 0;  // the last lane whose first pair is not after p: the ray pair p belongs to
#pragma unroll
                    for (uint32_t step = 32; step != 0; step >>= 1)
                        if (L.excl[j + step] <= p) j += step;
                    const uint32_t k = p - L.excl[j];
                    const float4 df = L.d_first[j];
                    const uint32_t rec = __float_as_uint(df.w) + k;
                    RT_CHECK(j < 64u && rec < s.n_refs, CHECK_DEAL, j, rec);
                    const float4 c0 = buf_load16<0>(refs_d, rec), c1 = buf_load16<16>(refs_d, rec), c2 = buf_load16<32>(refs_d, rec);
                    RT_PROBE_LOAD(rec)
                    if (COUNT) ctr->dealt_recs++;
                    const float4 oa = L.o_arrival[j];
                    if (!(__float_as_uint(c1.w) & __float_as_uint(oa.w))) {
                        const RayPk rp = make_ray_pk(v3(oa.x, oa.y, oa.z), v3(df.x, df.y, df.z));
                        const float t = moller_trumbore_pk<WGSL>(rp, c0, c1, c2);
                        bool accepted = t > 0.0f && t < 1e10f && __float_as_uint(c0.w) < REF_INVISIBLE;
                        if (TEX && accepted && __float_as_uint(c2.w) != 0u) {
                            // The alpha rule is a function of the hit alone, so a rejected candidate is rejected whatever
                            // the scan has seen before it: the minimum over the candidates that pass it is the in-order
                            // scan's answer. Candidates that cannot win any more are not sampled.
                            const unsigned long long now = L.key[j];
                            const bool can_win = ANY ? (uint32_t)(now >> 32) > k + 1u : (uint32_t)(now >> 32) >= __float_as_uint(t);
                            accepted = can_win;
                            if (can_win) {
                                const uint32_t mat = __float_as_uint(c2.w) - 1u;
                                float bu, bv;
                                hit_barycentrics(v3(oa.x, oa.y, oa.z), v3(df.x, df.y, df.z), v3(c0.x, c0.y, c0.z), v3(c1.x, c1.y, c1.z),
                                                 v3(c2.x, c2.y, c2.z), bu, bv);
                                const float4 bc = base_color(s.tex, mat, hit_uv(s.tex, __float_as_uint(c0.w) - s.nb, bu, bv));
                                accepted = !alpha_discards(s.tex, mat, bc.w);
                            }
                        }
                        if (accepted) {
                            const unsigned long long key = ANY ? ((unsigned long long)(k + 1u) << 32) | __float_as_uint(t)
                                                               : ((unsigned long long)__float_as_uint(t) << 32) | (k + 1u);
                            atomicMin(&L.key[j], key);
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned long long key = L.key[lane];
            if (alive && n != 0) {
                if (ANY) {
                    if (key != ~0ull) {
                        best_ref = cell.x + (uint32_t)(key >> 32) - 1u;
                        closest = __uint_as_float((uint32_t)key);
                    }
                } else if ((uint32_t)key != 0u) {
                    closest = __uint_as_float((uint32_t)(key >> 32));
                    best_ref = cell.x + (uint32_t)key - 1u;
                }
            }
            __builtin_amdgcn_wave_barrier();  // the next step overwrites the words only after every lane has read its own
        }
        if (alive) {
            const bool hit = best_ref != MISS;
            --left;
            const bool stop = (ANY & hit) | (hit & (closest < t_exit)) | (left == 0);
            idx = idx_next;
            arrival = ax ? mx : (ay ? my : mz);
            t_max.x += ax ? t_delta.x : 0.0f;
            t_max.y += ay ? t_delta.y : 0.0f;
            t_max.z += az ? t_delta.z : 0.0f;
            cell = cell_next;
            alive = !stop && cell.y != CELL_SENTINEL;
        }
    }
    bool got = best_ref != MISS;
    out = HitRec{closest, MISS, best_ref};
    if (got) out.id = __float_as_uint(buf_load16<0>(refs_d, best_ref).w);  // the hit record's id, read once at the end
    if (literal) {
        Counters none{};
        // Origin and direction go through an empty asm and the direction record is made again from them (the same
        // function, the same bits): otherwise what the set-up above derived - start cell, step signs - is kept in
        // registers, or spilled, through the whole loop for this rare path.
        V3 o2 = o, d2 = rd.d;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(o2.x), "+v"(o2.y), "+v"(o2.z), "+v"(d2.x), "+v"(d2.y), "+v"(d2.z));
#endif
        const RayDir rd2 = make_ray_dir<WGSL>(d2, s.cell_size, s.px, s.pxy);
        got = trace_ray<WGSL, ANY, 0, false, TEX, false, 2>(s, o2, rd2, out, none);  // the form without an LDS buffer
    }
    return active && got;
}


// Where a lane's pixel is: recomputed from the thread index wherever it is needed (before the primary traversal, between
// the two traversals, at the stores) instead of being carried in vector registers through both traversals, with the
// frame and the tile map read again from the kernel-argument segment each time instead of sitting in scalar registers
// (kept, the compiler spilled five vector registers: 24 bytes of scratch per lane written and read back through HBM).
// `tid` and the segment pointer go through an empty asm so that the compiler cannot merge the copies.
struct RepackPixel {
    uint32_t lt, lx, ly, px, py;
    bool in_tile, live;
};
__device__ inline void late_frame_and_map(DevFrame& f, ShardMap& m) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) uint8_t* KernArg;
    KernArg base = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));
    load_kernarg(f, reinterpret_cast<const __attribute__((address_space(4))) DevFrame*>(base + offsetof(PrimaryArgs, f)));
    load_kernarg(m, reinterpret_cast<const __attribute__((address_space(4))) ShardMap*>(base + offsetof(PrimaryArgs, m)));
#endif
}
__device__ inline RepackPixel repack_pixel(const DevFrame& f, const ShardMap& m) {
    uint32_t tid = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(tid));
#endif
    RepackPixel q;
    const uint32_t lane = tid & 63, w4 = tid >> 6;
    const uint32_t blk_x = (m.sub_x + 1) / 2, blk_y = (m.sub_y + 1) / 2;  // 16x16 blocks per tile
    const uint32_t blk_per_tile = blk_x * blk_y;
    q.lt = blockIdx.x / blk_per_tile;
    const uint32_t blk = blockIdx.x % blk_per_tile;
    const uint32_t sub_x = (blk % blk_x) * 2 + (w4 & 1), sub_y = (blk / blk_x) * 2 + (w4 >> 1);
    q.lx = sub_x * 8 + (lane & 7), q.ly = sub_y * 8 + (lane >> 3);
    const uint32_t tile = shard_tile(m, q.lt);
    q.px = (tile % m.tiles_x) * m.tile_w + q.lx, q.py = (tile / m.tiles_x) * m.tile_h + q.ly;
    q.in_tile = q.lx < m.tile_w && q.ly < m.tile_h;
    q.live = q.in_tile && q.px < f.width && q.py < f.height;
    return q;
}

// k_trace_primary with trace_ray_repack as the traversal: every lane of a wave enters both traversals, with or without
// a ray of its own. The kernel's leading arguments are PrimaryArgs' (scene, frame, tile map, the two output arrays):
// the frame and the map are read through late_frame_and_map, never through the parameters.
// COUNT: the counting build (rt_params.collect_stats = 2): rays, hits, and the records the lanes ask for.
#ifndef RT_REPACK_WAVES
// Waves per SIMD the dealing kernel is held to. Seven (72 vector registers, two of them spilled outside the loops: 12 bytes
// of scratch per lane) against six (74 registers, no scratch), same process, records identical: the scene that does not
// fit the caches (10 M triangles, 3840x2160) 1.84 against 1.94 ms per frame, the animated model 151 against 148 us.
#define RT_REPACK_WAVES 7
#endif
template <bool WGSL, bool SHADOW, bool COUNT = false>
__global__ __launch_bounds__(256, RT_REPACK_WAVES) void k_trace_primary_repack(DevScene s, DevFrame, ShardMap, uint2* __restrict__ out_p,
                                                              uint2* __restrict__ out_s, unsigned long long* stats) {
    __shared__ RepackLds repack[4];
    RepackLds& L = repack[(threadIdx.x >> 6) & 3];
    uint2 ph = make_uint2(MISS, 0u), sh = make_uint2(MISS, 0u);
    HitRec h;
    bool hit;
    Counters ctr{};
    unsigned long long n_rays = 0, n_sh = 0;
    {
        DevFrame f;
        ShardMap m;
        late_frame_and_map(f, m);
        const uint32_t blk_per_tile = ((m.sub_x + 1) / 2) * ((m.sub_y + 1) / 2);
        if (blockIdx.x / blk_per_tile >= m.n_local) return;  // whole workgroup
        const RepackPixel q = repack_pixel(f, m);
        const V3 d = primary_dir(f, q.live ? q.px : 0u, q.live ? q.py : 0u);
        const RayDir rd = make_ray_dir<WGSL>(d, s.cell_size, s.px, s.pxy);
        hit = trace_ray_repack<WGSL, false, 1, false, COUNT>(s, v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]), rd, q.live, h, L, &ctr);
        if (COUNT && q.live) n_rays++;
    }
    if (hit) ph = make_uint2(h.id, __float_as_uint(h.t));
    if (SHADOW) {
        DevFrame f;
        ShardMap m;
        late_frame_and_map(f, m);
        const V3 o = v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
        V3 so = o;
        if (hit) {
            const RepackPixel q = repack_pixel(f, m);
            const V3 d = primary_dir(f, q.px, q.py);  // the same arithmetic on the same pixel: the same bits
            const V3 pos = o + d * h.t;
            V3 n = hit_normal(s, h, pos);
            if (dot(n, d) > 0.0f) n = neg(n);
            so = pos + n * f.shadow_offset;
        }
        HitRec oh;
        if (trace_ray_repack<WGSL, true, 1, false, COUNT>(s, so, late_shadow_dir((uint32_t)offsetof(PrimaryArgs, f)), hit, oh, L, &ctr)) {
            sh = make_uint2(oh.id, __float_as_uint(oh.t));
            n_sh++;
        }
    }
    if (COUNT) {
        ctr.ref_recs += ctr.dealt_recs;
        flush_stats(stats, n_rays + (hit ? 1u : 0u), hit ? 1u : 0u, hit && SHADOW ? 1u : 0u, n_sh, ctr);
    }
    DevFrame f;
    ShardMap m;
    late_frame_and_map(f, m);
    const RepackPixel q = repack_pixel(f, m);
    if (q.in_tile) {
        const size_t oi = (size_t)q.lt * m.tile_w * m.tile_h + (size_t)q.ly * m.tile_w + q.lx;
        late_arg_ptr(offsetof(PrimaryArgs, out_p))[oi] = ph;
        if (SHADOW) late_arg_ptr(offsetof(PrimaryArgs, out_s))[oi] = sh;
    }
}

// Pixel blocks that are expensive: a plan made by one launch for the next one of the same frame shape.
//
// A frame's duration is not its work: the chip is full for the first two thirds of a launch, then drains for as long
// again while a few hundred waves finish - 8x8 pixel blocks over the model's densest cells, whose rays walk hundreds
// of candidates (one dependent chain per wave), and moderately expensive blocks that merely started late because they
// lie at the bottom of the picture. Which blocks those are hardly changes from one frame of a camera path (or of an
// animation) to the next, so every launch records how long each block's wave ran (the 100 MHz s_memrealtime clock:
// two scalar reads per wave, nothing in the traversal loop) and sorts the block into one of three classes for the
// next launch:
//   heavy  (>= threshold):      traced with FOUR LANES PER RAY (trace_ray<TRACE_QUAD>: a workgroup per block, a 4x4
//                               pixel quadrant per wave, candidate lists dealt to the four lanes of a ray), a quarter
//                               of the chain per list, by the workgroups at the very front of the grid;
//   early  (>= threshold / 2):  traced as ever, but by the workgroups that come next, so none of them starts late;
//   others:                     in picture order, by the rest of the grid.
// A wrong plan (first frame, camera cut) costs time, never correctness: every block is traced by exactly one wave or
// workgroup, and both shapes give the same bits.
struct BlockPlan {
    // One buffer per plan: the two list lengths (each in 128 bytes of its own; words 1 and 2: the threshold
    // the launch that follows this plan uses, and the one the lists were made with), then cap_heavy and cap_early block
    // numbers (8x8 blocks, numbered like the plain kernel's waves), then one class byte per 8x8 block (0 = others).
    // Three buffers rotate, so nothing needs clearing between launches.
    const uint8_t* cur;  // the plan this launch follows
    uint8_t* next;       // the plan it writes for the next launch; its list lengths are zero when the launch starts
    uint8_t* after;      // the buffer of the launch after that: this launch zeroes its list lengths
    uint32_t cap_heavy;  // most blocks the heavy list holds = workgroups that lead the grid
    uint32_t cap_early;  // most blocks the early list holds (a multiple of four) = 4 x the workgroups that follow
    uint32_t threshold;  // wave run time, in 10 ns ticks, below which no block is heavy (see plan_threshold)

    uint32_t bytes;      // size of one buffer

    static constexpr uint32_t HEADER = 256;
    __device__ static uint32_t& word(uint8_t* b, size_t at) { return *reinterpret_cast<uint32_t*>(b + at); }
    __device__ uint32_t at_heavy(uint32_t j) const { return HEADER + j * 4; }
    __device__ uint32_t at_early(uint32_t j) const { return HEADER + (cap_heavy + j) * 4; }
    __device__ uint32_t at_class(uint32_t gw) const { return HEADER + (cap_heavy + cap_early) * 4 + gw; }
};

// What a launch reads of its plan (every offset is the same in all lanes of the wave) goes through the scalar cache:
// the vector memory path is where the traversal's loads queue, and a wave would wait there before it has begun.
struct PlanReader {
    v4i d;
    __device__ explicit PlanReader(const BlockPlan& p) : d(make_rsrc(p.cur, 0, p.bytes)) {}
    __device__ uint32_t at(uint32_t byte_offset) const { return (uint32_t)llvm_s_buffer_load_i32(d, (int)byte_offset, 0); }
};

// One lane reports what its block cost. threshold: the launch's (see plan_threshold).
__device__ inline void plan_block(const BlockPlan& p, uint32_t gw, uint32_t ticks, uint32_t threshold) {
    uint8_t cls = 0;
    if (ticks >= threshold) {
        const uint32_t j = atomicAdd(&BlockPlan::word(p.next, 0), 1u);
        if (j < p.cap_heavy) {
            BlockPlan::word(p.next, p.at_heavy(j)) = gw;
            cls = 2;
        }
    }
    if (cls == 0 && ticks >= threshold / 2) {  // also where a full heavy list turned the block away
        const uint32_t j = atomicAdd(&BlockPlan::word(p.next, 128), 1u);
        if (j < p.cap_early) {
            BlockPlan::word(p.next, p.at_early(j)) = gw;
            cls = 1;
        }
    }
    p.next[p.at_class(gw)] = cls;
}

// The run time from which this launch calls a block heavy: the caller's threshold to begin with, then whatever the
// launches before have steered it to (word 1 of the plan a launch follows). Four lanes per ray pay for a few hundred
// blocks, the ones a frame waits for; which run time singles those out depends on the scene and on how many frames
// are in flight, so every launch looks at the length of the list it was given and moves the threshold for the next
// one: up by a quarter above PLAN_MANY blocks (doubled when the list was full), down by a sixteenth below PLAN_FEW,
// never below the caller's. The list a launch sees was made with the threshold of the launch before, which is kept
// beside it (word 2), and the correction starts from THAT value: even and odd launches steer themselves apart, and
// neither reacts to a list its own threshold did not make.
constexpr uint32_t PLAN_MANY = 512, PLAN_FEW = 128;
__device__ inline uint32_t plan_threshold(const BlockPlan& p, const PlanReader& plan) {
    const uint32_t steered = plan.at(4);
    return steered > p.threshold ? steered : p.threshold;
}
__device__ inline void steer_plan(const BlockPlan& p, const PlanReader& plan) {  // one thread of the launch
    BlockPlan::word(p.after, 0) = 0;
    BlockPlan::word(p.after, 128) = 0;
    // the list this launch follows, and the threshold it was made with (word 2; the launch before wrote both)
    const uint32_t n = plan.at(0), made_with = plan.at(8);
    uint32_t t = made_with > p.threshold ? made_with : p.threshold;
    if (n >= p.cap_heavy)
        t = t < 0x40000000u ? 2 * t : t;
    else if (n > PLAN_MANY)
        t = t < 0x60000000u ? t + t / 4 : t;
    else if (n < PLAN_FEW)
        t -= t / 16;
    BlockPlan::word(p.next, 4) = t;  // for the next launch; plan_threshold() keeps it at or above the caller's
    BlockPlan::word(p.next, 8) = plan_threshold(p, plan);  // what this launch makes the next list with
}


// k_trace_primary for triangle-only scenes without counters (the common case), with the plan above. The grid is
// cap_heavy workgroups for the heavy list, cap_early / 4 for the early list (those beyond a list's length exit at
// once), then the plain kernel's 16x16-pixel workgroups. One traversal body serves both shapes (the lanes-per-ray count
// is a run-time scalar): two bodies in one kernel need 76 vector registers, two kernels need two streams per frame in
// flight, and the four hardware queues of a process are then shared between them and frames no longer overlap.
// The first two arguments are read again late (late_shadow_dir at offsetof(PrimaryArgs, f)): keep (DevScene, DevFrame) first.
template <bool WGSL, bool SHADOW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96))) void k_trace_primary_planned(
    DevScene s, DevFrame f, ShardMap m, BlockPlan p, uint2* __restrict__ out_p, uint2* __restrict__ out_s) {
    const uint32_t t_start = (uint32_t)__builtin_amdgcn_s_memrealtime();
    // the wave's number inside the workgroup is the same in all its lanes: say so, and everything derived from it
    // (the block number, its tile, the plan look-up) stays in scalar registers
    const uint32_t lane = threadIdx.x & 63, w4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t blk_x = (m.sub_x + 1) / 2, blk_y = (m.sub_y + 1) / 2;  // 16x16 blocks per tile
    const uint32_t blk_per_tile = blk_x * blk_y;
    const uint32_t wg_early = p.cap_heavy, wg_rest = p.cap_heavy + p.cap_early / 4;
    const bool quad = blockIdx.x < wg_early;
    const PlanReader plan(p);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) steer_plan(p, plan);
    uint32_t gw;  // the 8x8 block this wave (or, four lanes per ray, this workgroup) traces
    if (quad) {
        if (blockIdx.x >= plan.at(0)) return;
        gw = plan.at(p.at_heavy(blockIdx.x));
    } else if (blockIdx.x < wg_rest) {
        const uint32_t j = (blockIdx.x - wg_early) * 4 + w4;
        if (j >= plan.at(128)) return;
        gw = plan.at(p.at_early(j));
    } else {
        gw = (blockIdx.x - wg_rest) * 4 + w4;
        // the class bytes of the workgroup's four blocks are one word; not zero: traced from a list
        if (((plan.at(p.at_class(gw & ~3u)) >> (8 * w4)) & 0xFFu) != 0) return;
    }
    const uint32_t b16 = gw >> 2, g4 = gw & 3;
    const uint32_t lt = b16 / blk_per_tile, blk = b16 % blk_per_tile;
    if (lt >= m.n_local) return;
    const uint32_t sub_x = (blk % blk_x) * 2 + (g4 & 1), sub_y = (blk / blk_x) * 2 + (g4 >> 1);
    // four lanes per ray: wave = 4x4 quadrant, four lanes per pixel; else: wave = the 8x8 block, a lane per pixel
    const uint32_t r = quad ? lane >> 2 : lane;
    const uint32_t lx = sub_x * 8 + (quad ? (w4 & 1) * 4 + (r & 3) : (r & 7));
    const uint32_t ly = sub_y * 8 + (quad ? (w4 >> 1) * 4 + (r >> 2) : (r >> 3));
    const uint32_t qstep = quad ? 4u : 1u;
    const uint32_t tile = shard_tile(m, lt);
    const uint32_t px = (tile % m.tiles_x) * m.tile_w + lx, py = (tile / m.tiles_x) * m.tile_h + ly;
    const bool in_tile = lx < m.tile_w && ly < m.tile_h;
    const bool live = in_tile && px < f.width && py < f.height;

    uint2 ph = make_uint2(MISS, 0u), sh = make_uint2(MISS, 0u);
    Counters ctr{};
    if (live) {
        const V3 o = v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
        const V3 d = primary_dir(f, px, py);
        const RayDir rd = make_ray_dir<WGSL>(d, s.cell_size, s.px, s.pxy);
        HitRec h;
        if (trace_ray<WGSL, false, 0, false, false, false, TRACE_QUAD>(s, o, rd, h, ctr, qstep)) {
            ph = make_uint2(h.id, __float_as_uint(h.t));
            if (SHADOW) {
                const V3 pos = o + d * h.t;
                V3 n = hit_normal(s, h, pos);
                if (dot(n, d) > 0.0f) n = neg(n);
                const V3 so = pos + n * f.shadow_offset;
                HitRec oh;
                const RayDir to_light = late_shadow_dir((uint32_t)offsetof(PrimaryArgs, f));
                if (trace_ray<WGSL, true, 0, false, false, false, TRACE_QUAD>(s, so, to_light, oh, ctr, qstep))
                    sh = make_uint2(oh.id, __float_as_uint(oh.t));
            }
        }
    }
    if (in_tile && (!quad || (lane & 3) == 0)) {
        const size_t oi = (size_t)lt * m.tile_w * m.tile_h + (size_t)ly * m.tile_w + lx;
        out_p[oi] = ph;
        if (SHADOW) out_s[oi] = sh;
    }
    // what this block cost, for the next launch's plan
    const uint32_t ticks = (uint32_t)__builtin_amdgcn_s_memrealtime() - t_start;
    if (quad) {
        // the slowest of the four waves; one lane per ray it would have taken about three times as long: say so, or
        // the block would leave the list again because the list made it fast
        __shared__ uint32_t wg_ticks;
        if (threadIdx.x == 0) wg_ticks = 0;
        __syncthreads();
        if (lane == 0) atomicMax(&wg_ticks, ticks);
        __syncthreads();
        if (threadIdx.x == 0) plan_block(p, gw, 3u * wg_ticks, plan_threshold(p, plan));
    } else if (lane == 0) {
        plan_block(p, gw, ticks, plan_threshold(p, plan));
    }
#ifdef RT_EXPERIMENTS
    if (gw < 65536 && (quad ? threadIdx.x == 0 : lane == 0)) {
        g_wave_t[2 * gw] = __builtin_amdgcn_s_memrealtime() - ticks;
        g_wave_t[2 * gw + 1] = __builtin_amdgcn_s_memrealtime();
        g_wave_chain[gw] = ticks | (quad ? 0x80000000u : blockIdx.x < wg_rest ? 0x40000000u : 0u);
    }
#endif
}

// Explicit ray list: one lane per ray.
template <bool WGSL, bool ANY, int STATS, bool BOXES, bool TEX = false, bool WIDE = false>
__global__ __launch_bounds__(256) void k_trace_rays(DevScene s, const float* __restrict__ origins,
                                                    const float* __restrict__ dirs, uint32_t n,
                                                    uint2* __restrict__ out, unsigned long long* stats) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Counters ctr{};
    unsigned long long n_rays = 0, n_hit = 0;
    if (i < n) {
        HitRec h;
        n_rays = 1;
        const RayDir rd = make_ray_dir<WGSL>(ld3(dirs + 3 * (size_t)i), s.cell_size, s.px, s.pxy);
        bool hit = trace_ray<WGSL, ANY, STATS, BOXES, TEX, WIDE>(s, ld3(origins + 3 * (size_t)i), rd, h, ctr);
        n_hit = hit ? 1 : 0;
        out[i] = hit ? make_uint2(h.id, __float_as_uint(h.t)) : make_uint2(MISS, 0u);
    }
    if (STATS) flush_stats(stats, n_rays, n_hit, 0, 0, ctr);
}

// k_trace_rays for lists too short to fill the chip (triangle-only scenes, no counters): four lanes to a ray. Every
// lane walks the grid; a cell's list is dealt to the ray's four lanes and the four results meet under the key
// (t, list position), which is what a scan in order with a strict '<' yields (trace_ray<X = TRACE_QUAD>, the form the
// plan of expensive blocks uses) - same records bit for bit, a ray's chain of candidate tests a quarter as long.
template <bool WGSL, bool ANY>
__global__ __launch_bounds__(256) void k_trace_rays_quad(DevScene s, const float* __restrict__ origins, const float* __restrict__ dirs,
                                                         uint32_t n, uint2* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = (uint32_t)(t >> 2);
    if (i >= n) return;  // a ray's four lanes leave together
    Counters ctr{};
    HitRec h;
    const RayDir rd = make_ray_dir<WGSL>(ld3(dirs + 3 * (size_t)i), s.cell_size, s.px, s.pxy);
    const bool hit = trace_ray<WGSL, ANY, 0, false, false, false, TRACE_QUAD>(s, ld3(origins + 3 * (size_t)i), rd, h, ctr, 4u);
    if ((threadIdx.x & 3u) == 0) out[i] = hit ? make_uint2(h.id, __float_as_uint(h.t)) : make_uint2(MISS, 0u);
}

// k_trace_rays with trace_ray_repack as the traversal (rt_params.repack_candidates): a caller's ray list is rarely
// ordered so that neighbouring rays share their cells. Lanes past the end of the list take part without a ray.
template <bool WGSL, bool ANY>
__global__ __launch_bounds__(256, 6) void k_trace_rays_repack(DevScene s, const float* __restrict__ origins,
                                                              const float* __restrict__ dirs, uint32_t n, uint2* __restrict__ out) {
    __shared__ RepackLds repack[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool active = i < n;
    const size_t at = active ? i : 0u;
    const RayDir rd = make_ray_dir<WGSL>(ld3(dirs + 3 * at), s.cell_size, s.px, s.pxy);
    HitRec h;
    const bool hit = trace_ray_repack<WGSL, ANY, 1, false, false, 75u>(s, ld3(origins + 3 * at), rd, active, h, repack[(threadIdx.x >> 6) & 3]);
    if (active) out[i] = hit ? make_uint2(h.id, __float_as_uint(h.t)) : make_uint2(MISS, 0u);
}

// One pixel, one lane: what the shader writes for its debug pixel (:695-709), geometry fields.
template <bool WGSL, bool WIDE = false>
__global__ void k_debug_pixel(DevScene s, DevFrame f, uint32_t px, uint32_t py, float* __restrict__ info20) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const V3 o = v3(f.cam_pos[0], f.cam_pos[1], f.cam_pos[2]);
    const V3 d = primary_dir(f, px, py);
    HitRec h;
    Counters ctr{};
    const RayDir rd = make_ray_dir<WGSL>(d, s.cell_size, s.px, s.pxy);
    const bool hit = trace_ray<WGSL, false, 1, true, true, WIDE>(s, o, rd, h, ctr);
    for (int k = 0; k < 20; ++k) info20[k] = 0.0f;
    info20[0] = o.x, info20[1] = o.y, info20[2] = o.z;
    info20[3] = hit ? 1.0f : 0.0f;
    info20[4] = d.x, info20[5] = d.y, info20[6] = d.z;
    info20[7] = hit ? h.t : 1e10f;
    info20[11] = hit ? (float)h.id : -1.0f;
    info20[15] = (float)ctr.steps;
    if (hit) {
        const V3 pos = o + d * h.t;
        const V3 n = hit_normal(s, h, pos);
        info20[8] = pos.x, info20[9] = pos.y, info20[10] = pos.z;
        info20[12] = n.x, info20[13] = n.y, info20[14] = n.z;
    }
}

}  // namespace rt
