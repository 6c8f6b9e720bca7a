// C ABI of librt_mi355.so (include/rt_mi355.h): context, scene/grid upload, trace launches.
// Host code here only moves bytes, evaluates the per-frame constants and launches kernels;
// everything a ray does happens in rt_trace.hpp, grid binning in rt_build.hip.
#include <cmath>
#include <mutex>
#include <vector>

#include "rt_internal.hpp"
#ifdef RT_EXPERIMENTS
#include <cstdlib>
#include <type_traits>

#include "rt_trace_variants.hpp"
#endif
#include "rt_cull.hpp"
#include "rt_render.hpp"

using namespace rt;

namespace {

thread_local std::string g_create_error;

// ---------------------------------------------------------------- scratch of the entry point that allocates nothing
// rt_trace_primary_noalloc keeps the one buffer a trace needs beyond its inputs and outputs - the per-frame box
// records k_box_frame writes and the trace kernels read - in this array instead of a hipMalloc'ed one. Every GPU the
// library runs on holds its own copy (512 KiB) at its own address until the process ends. In host code the array's
// name is only a handle for hipGetSymbolAddress: rt_create asks once per device, with that device current, and
// everything below passes the address it got (g_scratch_ptr[device]) - never the name - to kernels.
constexpr uint32_t NOALLOC_MAX_BOXES = 8192;
static __device__ __align__(256) float4 scratch[4 * NOALLOC_MAX_BOXES];
constexpr int NOALLOC_MAX_DEVICES = 64;
float4* g_scratch_ptr[NOALLOC_MAX_DEVICES];  // written under g_scratch_lock, read after rt_create returned
std::mutex g_scratch_lock;
// Every context on a GPU shares that GPU's array. Inside one context the stream switch orders its users (scratch_busy /
// ev_order in rt_set_stream); between contexts this does: the context whose launches used the array last, and the event
// it recorded behind them (its own, made in rt_create). A call from ANOTHER context makes its stream wait for that event
// before its k_box_frame overwrites the records, then takes the entry over. Nothing here allocates.
struct ScratchUser {
    rt_ctx* ctx = nullptr;
    hipEvent_t done = nullptr;
};
ScratchUser g_scratch_user[NOALLOC_MAX_DEVICES];  // under g_scratch_lock

hipError_t scratch_for_device(int device) {  // the device must be current
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return hipSuccess;  // rt_trace_primary_noalloc reports it
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (g_scratch_ptr[device]) return hipSuccess;
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e == hipSuccess) g_scratch_ptr[device] = static_cast<float4*>(p);
    return e;
}

// ---------------------------------------------------------------- small kernels

// (first, count) per cell, in the bordered layout the trace kernels walk: a one-cell shell of sentinel records
// around the gx x gy x gz grid.
// The count is stored already cut to `cap`, the number of entries of a list the traversal looks at
// (raytracer_unified.wgsl:531 min(count, MAX_OBJECTS_PER_CELL)), so the trace kernels use it as it is.
__global__ void k_emit_cells(const uint32_t* __restrict__ cell_start, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t cap,
                             uint2* __restrict__ cells) {
    const uint32_t px = gx + 2, py = gy + 2, pz = gz + 2;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)px * py * pz) return;
    const uint32_t x = (uint32_t)(p % px), y = (uint32_t)((p / px) % py), z = (uint32_t)(p / ((size_t)px * py));
    uint2 rec = make_uint2(0u, CELL_SENTINEL);
    if (x >= 1 && x <= gx && y >= 1 && y <= gy && z >= 1 && z <= gz) {
        const size_t c = (size_t)(x - 1) + (size_t)(y - 1) * gx + (size_t)(z - 1) * gx * gy;
        const uint32_t a = cell_start[c], b = cell_start[c + 1];
        rec = make_uint2(a, b - a < cap ? b - a : cap);
    }
    cells[p] = rec;
}

// reference: raytracer_unified.wgsl:147-155
__device__ inline bool lod_culled(V3 cam, V3 center, V3 size, float lod_factor, float min_pixel_size) {
    float distance = length(cam - center);
    if (distance > 200.0f) return true;
    float max_size = fmaxf(fmaxf(size.x, size.y), size.z);
    float apparent = (max_size / distance) * lod_factor;
    return apparent < min_pixel_size;
}

// Per-frame box records: interpolated bounds (raytracer_unified.wgsl:200-206) and both
// LOD-cull outcomes (grid path :536-539, moving-box pre-pass :449-453).
__global__ void k_box_frame(const BoxData* __restrict__ boxes, uint32_t nb, float t_lerp, V3 cam, float lod_factor,
                            float min_pixel_size, uint32_t lod_cull, float4* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const BoxData b = boxes[i];
    V3 center = mix(ld3(b.center0), ld3(b.center1), t_lerp);
    V3 half = ld3(b.half_size);
    V3 lo = center - half, hi = center + half;
    uint32_t flags = 0;
    if (lod_cull) {
        V3 gc = (ld3(b.min) + ld3(b.max)) * 0.5f, gs = ld3(b.max) - ld3(b.min);
        if (lod_culled(cam, gc, gs, lod_factor, min_pixel_size)) flags |= BOX_CULLED_GRID;
        if (lod_culled(cam, center, half * 2.0f, lod_factor, min_pixel_size)) flags |= BOX_CULLED_PREPASS;
    }
    out[4 * (size_t)i] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(flags));
    out[4 * (size_t)i + 1] = make_float4(hi.x, hi.y, hi.z, 0.f);
    out[4 * (size_t)i + 2] = make_float4(center.x, center.y, center.z, 0.f);
    out[4 * (size_t)i + 3] = make_float4(half.x, half.y, half.z, 0.f);
}

// Linear-blend skinning of a triangle's corners: S = ((w0*M[j0] + w1*M[j1]) + w2*M[j2]) + w3*M[j3] element
// by element, then the point transform ((S.x*p.x + S.y*p.y) + S.z*p.z) + S.w (columns of S; glam's
// transform_point3 order). Matrices are column-major, one per joint. Writes the trace's vertex array directly,
// so the binning pass that follows reads what this kernel wrote without a trip through the host.
// One thread = one triangle: its three corners, then its geometric normal (normalize(cross(v1 - v0, v2 - v0)),
// src/core/triangle_intersection.rs:79) - one launch per frame instead of one per corner array and one per normal array: a
// frame of an animated model is a chain of small launches, and each costs about as much as this one's work.
__global__ void k_skin_triangles(const float* __restrict__ pos, const ushort4* __restrict__ joints, const float4* __restrict__ weights,
                                 const uint32_t* __restrict__ tri_idx, uint32_t nt, const float4* __restrict__ mats,
                                 float* __restrict__ tri_v, float4* __restrict__ tri_n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    V3 corner[3];
#pragma unroll
    for (uint32_t k = 0; k < 3; ++k) {
        const uint32_t v = tri_idx[3 * (size_t)t + k];
        const V3 p = ld3(pos + 3 * (size_t)v);
        const ushort4 j = joints[v];
        const float4 w = weights[v];
        V3 col[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 a = mats[4 * j.x + c], b = mats[4 * j.y + c], d = mats[4 * j.z + c], e = mats[4 * j.w + c];
            col[c] = ((v3(a.x, a.y, a.z) * w.x + v3(b.x, b.y, b.z) * w.y) + v3(d.x, d.y, d.z) * w.z) + v3(e.x, e.y, e.z) * w.w;
        }
        corner[k] = ((col[0] * p.x + col[1] * p.y) + col[2] * p.z) + col[3];
        float* o = tri_v + 9 * (size_t)t + 3 * k;
        o[0] = corner[k].x, o[1] = corner[k].y, o[2] = corner[k].z;
    }
    const V3 n = normalize(cross(corner[1] - corner[0], corner[2] - corner[0]));
    tri_n[t] = make_float4(n.x, n.y, n.z, 0.f);
}

// Geometric normal per triangle: normalize(cross(v1 - v0, v2 - v0)) (src/core/triangle_intersection.rs:79).
__global__ void k_tri_normals(const float* __restrict__ tri_v, uint32_t first, uint32_t count, float4* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float* v = tri_v + 9 * (size_t)(first + i);
    const V3 v0 = ld3(v), n = normalize(cross(ld3(v + 3) - v0, ld3(v + 6) - v0));
    out[first + i] = make_float4(n.x, n.y, n.z, 0.f);
}

__global__ void k_tri_unpack(const TriangleData* __restrict__ tris, uint32_t nt, float* __restrict__ tri_v,
                             uint32_t* __restrict__ tri_mat, float* __restrict__ tri_uv) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    const TriangleData t = tris[i];
    tri_mat[i] = f2u_sat(t.material_id);  // u32(tri.material_id), raytracer_unified.wgsl:280
    float* o = tri_v + 9 * (size_t)i;
    o[0] = t.v0[0], o[1] = t.v0[1], o[2] = t.v0[2];
    o[3] = t.v1[0], o[4] = t.v1[1], o[5] = t.v1[2];
    o[6] = t.v2[0], o[7] = t.v2[1], o[8] = t.v2[2];
    float* uv = tri_uv + 6 * (size_t)i;
    uv[0] = t.uv0[0], uv[1] = t.uv0[1], uv[2] = t.uv1[0], uv[3] = t.uv1[1], uv[4] = t.uv2[0], uv[5] = t.uv2[1];
}

// Every (ray, box) pair. A thread keeps four consecutive boxes in registers and walks down the rays (blockIdx.y, then in
// steps of gridDim.y): the ray and its safe inverse direction are the same for the whole workgroup and are formed once
// per four pairs, the four results leave as one 16-byte store when every row of `out` starts on a 16-byte boundary
// (WIDE: n_prims a multiple of four and `out` aligned). Per pair the arithmetic is safe_inv and slab, as before.
template <bool WGSL, bool WIDE>
__global__ __launch_bounds__(256) void k_aabb_pairs(const float* __restrict__ o, const float* __restrict__ d, uint32_t n_rays,
                                                    const float* __restrict__ bmin, const float* __restrict__ bmax, uint32_t n_prims,
                                                    float* __restrict__ out) {
    const uint32_t p0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (p0 >= n_prims) return;
    const uint32_t np = n_prims - p0 < 4u ? n_prims - p0 : 4u;
    V3 mn[4], mx[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t p = p0 + (k < np ? k : 0u);
        mn[k] = ld3(bmin + 3 * (size_t)p);
        mx[k] = ld3(bmax + 3 * (size_t)p);
    }
    for (uint32_t r = blockIdx.y; r < n_rays; r += gridDim.y) {
        const V3 oo = ld3(o + 3 * (size_t)r), dd = ld3(d + 3 * (size_t)r);
        const V3 inv = v3(safe_inv<WGSL>(dd.x), safe_inv<WGSL>(dd.y), safe_inv<WGSL>(dd.z));
        float t[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) t[k] = slab(oo, inv, mn[k], mx[k]);
        float* row = out + (size_t)r * n_prims + p0;
        if (WIDE) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(f4{t[0], t[1], t[2], t[3]}, reinterpret_cast<f4*>(row));  // written once, never read here
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k)
                if (k < np) row[k] = t[k];
        }
    }
}

template <bool WGSL>
__global__ void k_tri_pairs(const float* __restrict__ o, const float* __restrict__ d, uint32_t n_rays,
                            const float* __restrict__ v012, uint32_t n_prims, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_rays * n_prims) return;
    uint32_t r = (uint32_t)(i / n_prims), p = (uint32_t)(i % n_prims);
    const float* v = v012 + 9 * (size_t)p;
    V3 v0 = ld3(v);
    const V3 e1 = ld3(v + 3) - v0, e2 = ld3(v + 6) - v0;
    // the form the trace kernels use (packed two-lane arithmetic), so this entry point checks it pair by pair
    out[i] = moller_trumbore_pk<WGSL>(make_ray_pk(ld3(o + 3 * (size_t)r), ld3(d + 3 * (size_t)r)), make_float4(v0.x, v0.y, v0.z, 0.f),
                                      make_float4(e1.x, e1.y, e1.z, 0.f), make_float4(e2.x, e2.y, e2.z, 0.f));
}

__global__ void k_tri_pairs_watertight(const float* __restrict__ o, const float* __restrict__ d, uint32_t n_rays,
                                       const float* __restrict__ v012, uint32_t n_prims, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_rays * n_prims) return;
    uint32_t r = (uint32_t)(i / n_prims), p = (uint32_t)(i % n_prims);
    const float* v = v012 + 9 * (size_t)p;
    out[i] = watertight(ld3(o + 3 * (size_t)r), ld3(d + 3 * (size_t)r), ld3(v), ld3(v + 3), ld3(v + 6));
}

// Which rank traced a frame tile, and where it sits in that rank's block: without a table rank = tile % n_ranks;
// with one, three words per place o of a round: (rank, how many places of that rank come before o, places of that rank).
struct AssemblePlan {
    uint32_t n_ranks, round;
    const uint32_t* places;
};
constexpr uint32_t OWNER_SKY = 0xFFFFFFFFu;  // RT_OWNER_SKY: a tile nobody rendered; the assemble kernels fill it from the camera

// What the sky fill needs of a frame's camera (primary_dir reads nothing else), for up to 64 frames of a group.
constexpr uint32_t SKY_MAX_FRAMES = 64;
struct SkyFrames {
    float basis[SKY_MAX_FRAMES][9];  // fwd, right, up
    float fov_scale;
    uint32_t enabled;
};

// The shader's sky colour for pixel (px, py) of frame j as k_render writes it for a ray that meets nothing
// (rt_render.hpp: acc = 0 + mix(...) * 1, then the 8-bit store); alpha 255.
__device__ inline uint32_t sky_pixel(const SkyFrames& sky, uint32_t j, uint32_t width, uint32_t height, uint32_t px, uint32_t py) {
    DevFrame f;
    for (int k = 0; k < 3; ++k) f.fwd[k] = sky.basis[j][k], f.right[k] = sky.basis[j][3 + k], f.up[k] = sky.basis[j][6 + k];
    f.fov_scale = sky.fov_scale;
    f.width = width, f.height = height;
    const V3 d = primary_dir(f, px, py);
    const float t = (d.y + 1.0f) * 0.5f;
    const V3 acc = v3(0.0f, 0.0f, 0.0f) + mix(v3(0.3f, 0.5f, 0.7f), v3(0.5f, 0.7f, 1.0f), t) * 1.0f;
    return unorm8(acc.x) | (unorm8(acc.y) << 8) | (unorm8(acc.z) << 16) | 0xFF000000u;
}
__device__ inline void locate_tile(const AssemblePlan& p, uint32_t tile, uint32_t& rank, uint32_t& lt) {
    if (!p.places) {
        rank = tile % p.n_ranks;
        lt = tile / p.n_ranks;
        return;
    }
    const uint32_t q = tile / p.round, o = tile - q * p.round;
    rank = p.places[3 * o];
    lt = q * p.places[3 * o + 2] + p.places[3 * o + 1];
}

// One pixel per thread; RGB8: 3-byte pixels in, 4-byte pixels (alpha 255) out.
template <class PIXEL, bool RGB8>
__global__ void k_assemble(const void* __restrict__ gathered, uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h,
                           uint32_t tiles_x, AssemblePlan plan, size_t per_rank, size_t frame_stride, PIXEL* __restrict__ image, SkyFrames sky) {
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= width || y >= height) return;
    uint32_t rank, lt;
    locate_tile(plan, (y / tile_h) * tiles_x + (x / tile_w), rank, lt);
    if constexpr (sizeof(PIXEL) == 4) {
        if (sky.enabled && rank == OWNER_SKY) {
            image[(size_t)blockIdx.z * width * height + (size_t)y * width + x] = sky_pixel(sky, blockIdx.z, width, height, x, y);
            return;
        }
    }
    const size_t src = (size_t)blockIdx.z * frame_stride + (size_t)rank * per_rank + (size_t)lt * tile_w * tile_h +
                       (size_t)(y % tile_h) * tile_w + (x % tile_w);
    const size_t dst = (size_t)blockIdx.z * width * height + (size_t)y * width + x;
    if constexpr (RGB8) {
        const uint8_t* p = static_cast<const uint8_t*>(gathered) + 3 * src;
        image[dst] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xFF000000u;
    } else {
        image[dst] = static_cast<const PIXEL*>(gathered)[src];
    }
}

// The same for layouts where 16 bytes of an output row (PER pixels) never straddle a tile or the image edge: one
// workgroup per tile, a thread moves PER pixels at a time - 16 bytes in (12 for RGB8: three dwords), one 16-byte store
// out - and the tile -> (rank, block) arithmetic is done once per workgroup.
template <uint32_t PER, bool RGB8>
__global__ __launch_bounds__(256) void k_assemble_x16(const uint32_t* __restrict__ gathered, uint32_t width, uint32_t height,
                                                      uint32_t tile_w, uint32_t tile_h, uint32_t tiles_x, AssemblePlan plan,
                                                      size_t per_rank, size_t frame_stride, uint4* __restrict__ image, SkyFrames sky) {
    uint32_t rank, lt;
    locate_tile(plan, blockIdx.y * tiles_x + blockIdx.x, rank, lt);
    const uint32_t x0 = blockIdx.x * tile_w, y0 = blockIdx.y * tile_h;
    const uint32_t vec_x = tile_w / PER;
    const uint32_t cols = min(vec_x, (width - x0) / PER), rows = min(tile_h, height - y0);
    if constexpr (PER == 4) {  // 4-byte pixels: a tile nobody rendered is the sky seen through this frame's camera
        if (sky.enabled && rank == OWNER_SKY) {
            uint4* dst = image + ((size_t)blockIdx.z * width * height + (size_t)y0 * width + x0) / PER;
            for (uint32_t q = threadIdx.x; q < vec_x * rows; q += blockDim.x) {
                const uint32_t ly = q / vec_x, lq = q - ly * vec_x;
                if (lq >= cols) continue;
                const uint32_t px = x0 + lq * PER, py = y0 + ly;
                dst[(size_t)ly * (width / PER) + lq] = make_uint4(sky_pixel(sky, blockIdx.z, width, height, px, py), sky_pixel(sky, blockIdx.z, width, height, px + 1, py),
                                                                 sky_pixel(sky, blockIdx.z, width, height, px + 2, py), sky_pixel(sky, blockIdx.z, width, height, px + 3, py));
            }
            return;
        }
    }
    // first record of the tile, in vectors; RGB8: a multiple of 4 pixels is a whole number of dwords
    const size_t first = ((size_t)blockIdx.z * frame_stride + (size_t)rank * per_rank + (size_t)lt * tile_w * tile_h) / PER;
    const uint32_t* src = gathered + first * (RGB8 ? 3 : 4);
    uint4* dst = image + ((size_t)blockIdx.z * width * height + (size_t)y0 * width + x0) / PER;
    const uint32_t n = vec_x * rows;
    for (uint32_t q = threadIdx.x; q < n; q += blockDim.x) {
        const uint32_t ly = q / vec_x, lq = q - ly * vec_x;
        if (lq >= cols) continue;
        uint4 o;
        if constexpr (RGB8) {
            const uint32_t* p = src + 3 * (size_t)q;
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            o.x = (d0 & 0x00FFFFFFu) | 0xFF000000u;
            o.y = (d0 >> 24) | ((d1 & 0x0000FFFFu) << 8) | 0xFF000000u;
            o.z = (d1 >> 16) | ((d2 & 0x000000FFu) << 16) | 0xFF000000u;
            o.w = (d2 >> 8) | 0xFF000000u;
        } else {
            o = reinterpret_cast<const uint4*>(src)[q];
        }
        dst[(size_t)ly * (width / PER) + lq] = o;
    }
}

// ---------------------------------------------------------------- host helpers

int use(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    c->err.clear();
    RT_HIP(c, hipSetDevice(c->device));
    return RT_OK;
}

// Before the scene or the grid is replaced: nothing queued on any stream may still read the old one.
int scene_barrier(rt_ctx* c) {
    if (c->other_streams_busy) {
        RT_HIP(c, hipDeviceSynchronize());
        c->other_streams_busy = false;
        c->scratch_busy = false;
        c->traced_on.clear();
    }
    return RT_OK;
}

DevScene make_scene(const rt_ctx* c) {
    DevScene s;
    memset(&s, 0, sizeof(s));
    s.refs = c->refs.as<float4>();
    s.cells = c->cells.as<uint2>();
    s.tri_normal = c->tri_n.as<float4>();
    s.box_frame = c->box_frame.as<float4>();
    s.nb = c->grid_nb;
    s.nt = c->nt;
    s.n_cells = c->n_cells;
    s.n_refs = c->n_refs_t;
    s.gx = c->meta.grid_sizes[RT_GRID_LEVELS - 1][0];
    s.gy = c->meta.grid_sizes[RT_GRID_LEVELS - 1][1];
    s.gz = c->meta.grid_sizes[RT_GRID_LEVELS - 1][2];
    s.px = s.gx + 2;
    s.pxy = s.px * (s.gy + 2);
    s.n_bordered = s.pxy * (s.gz + 2);
    s.max_steps = c->params.max_dda_steps;
    s.moving_start = s.nb >= c->params.moving_box_count ? s.nb - c->params.moving_box_count : 0u;
    for (int k = 0; k < 3; ++k) {
        s.bmin[k] = c->meta.bounds_min[k];
        s.bmax[k] = c->meta.bounds_max[k];
    }
    s.cell_size = c->meta.finest_cell_size;
    s.inv_cell_size = 1.0f / fmaxf(s.cell_size, 1e-8f);  // raytracer_unified.wgsl:384
    s.tex.texels = c->texels.as<uchar4>();
    s.tex.w = c->tex_w;
    s.tex.h = c->tex_h;
    s.tex.layers = c->tex_layers;
    s.tex.tri_uv = c->tri_uv.as<float>();
    s.tex.tri_mat = c->tri_mat.as<uint32_t>();
    s.tex.materials = c->materials.as<float4>();
    s.tex.n_materials = c->n_materials;
    return s;
}

// Per-frame constants the shader evaluates per pixel; evaluated once here with the host libm.
DevFrame make_frame(const rt_ctx* c, const CameraUniform* cam, uint32_t w, uint32_t h) {
    const uint32_t px = c->meta.grid_sizes[RT_GRID_LEVELS - 1][0] + 2;
    const uint32_t pxy = px * (c->meta.grid_sizes[RT_GRID_LEVELS - 1][1] + 2);
    DevFrame f;
    memset(&f, 0, sizeof(f));
    if (cam) {
        for (int k = 0; k < 3; ++k) {
            f.cam_pos[k] = cam->position[k];
            f.fwd[k] = cam->forward[k];
            f.right[k] = cam->right[k];
            f.up[k] = cam->up[k];
        }
    }
    f.fov_scale = tanf(c->params.fov);                       // raytracer_unified.wgsl:643
    const V3 sd = neg(normalize(ld3(c->params.light_direction)));  // raytracer_unified.wgsl:599-600
    f.shadow = c->params.wgsl_compat ? make_ray_dir<true>(sd, c->meta.finest_cell_size, px, pxy)
                                     : make_ray_dir<false>(sd, c->meta.finest_cell_size, px, pxy);
    f.shadow_offset = c->params.shadow_offset;
    f.width = w;
    f.height = h;
    return f;
}

// Per-frame box records for n consecutive frames (frame j at record 4 * nb * j).
int update_box_frames(rt_ctx* c, const CameraUniform* cams, uint32_t n) {
    if (c->grid_nb == 0) return RT_OK;
    if (!cams) return fail(c, RT_ERR_INVALID, "camera is required when the scene has boxes (LOD cull, box motion)");
    RT_HIP(c, c->box_frame.reserve((size_t)c->nb * 4 * sizeof(float4) * n));
    c->scratch_busy = true;
    for (uint32_t j = 0; j < n; ++j) {
        const CameraUniform* cam = cams + j;
        float t_lerp = (sinf(cam->time * 2.0f) + 1.0f) * 0.5f;  // raytracer_unified.wgsl:201,449
        hipLaunchKernelGGL(k_box_frame, dim3((c->nb + 255) / 256), dim3(256), 0, c->stream, c->boxes.as<BoxData>(), c->nb,
                           t_lerp, ld3(cam->position), cam->lod_factor, cam->min_pixel_size, c->params.lod_cull,
                           c->box_frame.as<float4>() + (size_t)c->nb * 4 * j);
    }
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int update_box_frame(rt_ctx* c, const CameraUniform* cam) { return update_box_frames(c, cam, 1); }

constexpr uint32_t MAX_OWNER_TABLE = 65536;  // a table over every 32x32 tile of a 7680x4320 frame has 32 400 entries

// Device copy of `content`, uploaded the first time `key` is seen.
int table_for(rt_ctx* c, std::vector<uint32_t>&& key, const std::vector<uint32_t>& content, const uint32_t** dev,
              bool may_upload = true) {
    for (rt_table* t : c->tables)
        if (t->key == key) {
            *dev = t->buf.as<uint32_t>();
            return RT_OK;
        }
    if (!may_upload)
        return fail(c, RT_ERR_STATE, "this owner table is not on the device yet: make one allocating call with this shard first");
    if (c->tables.size() >= 256) {  // a caller that keeps changing its tables: wait for every reader, start over
        RT_HIP(c, hipDeviceSynchronize());
        for (rt_table* t : c->tables) {
            t->buf.release();
            delete t;
        }
        c->tables.clear();
    }
    rt_table* t = new rt_table;
    c->tables.push_back(t);
    RT_HIP(c, t->buf.reserve(content.size() * sizeof(uint32_t)));
    RT_HIP(c, hipMemcpy(t->buf.p, content.data(), content.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    t->key = std::move(key);  // only now: a failed upload must not be found again
    *dev = t->buf.as<uint32_t>();
    return RT_OK;
}

int make_shard(rt_ctx* c, uint32_t w, uint32_t h, const rt_shard* sh, ShardMap* out, const char* who, bool may_upload = true) {
    ShardMap m;
    memset(&m, 0, sizeof(m));
    if (sh && (!sh->tile_w || !sh->tile_h || !sh->stride))
        return fail(c, RT_ERR_INVALID, "%s: shard tile size and stride must be non-zero", who);
    m.tile_w = sh ? sh->tile_w : w;
    m.tile_h = sh ? sh->tile_h : h;
    m.first = sh ? sh->first : 0;
    m.stride = sh ? sh->stride : 1;
    m.tiles_x = (w + m.tile_w - 1) / m.tile_w;
    m.tiles_y = (h + m.tile_h - 1) / m.tile_h;
    m.n_tiles = m.tiles_x * m.tiles_y;
    m.count = 1;
    m.n_local = m.first < m.n_tiles ? (m.n_tiles - m.first + m.stride - 1) / m.stride : 0;
    if (sh && sh->owner) {
        if (sh->stride > MAX_OWNER_TABLE)
            return fail(c, RT_ERR_LIMIT, "%s: owner table of %u entries (max %u)", who, sh->stride, MAX_OWNER_TABLE);
        std::vector<uint32_t> key{0u, sh->first, sh->stride}, mine;
        key.insert(key.end(), sh->owner, sh->owner + sh->stride);
        const uint32_t rem = m.n_tiles % m.stride;  // tiles of the last, partial round
        uint32_t in_rem = 0;
        for (uint32_t o = 0; o < sh->stride; ++o)
            if (sh->owner[o] == sh->first) {
                mine.push_back(o);
                in_rem += o < rem ? 1u : 0u;
            }
        m.count = (uint32_t)mine.size();
        m.n_local = m.n_tiles / m.stride * m.count + in_rem;
        if (m.count == 0) {
            m.count = 1;  // this rank owns nothing: an empty call
            m.n_local = 0;
        } else if (int r = table_for(c, std::move(key), mine, &m.offsets, may_upload)) {
            return r;
        }
    }
    m.sub_x = (m.tile_w + 7) / 8;
    m.sub_y = (m.tile_h + 7) / 8;
    *out = m;
    return RT_OK;
}

int zero_stats(rt_ctx* c) {
    RT_HIP(c, c->stats.reserve(16 * sizeof(unsigned long long)));
    RT_HIP(c, hipMemsetAsync(c->stats.p, 0, 16 * sizeof(unsigned long long), c->stream));
    return RT_OK;
}

int fetch_stats(rt_ctx* c) {
    unsigned long long v[16];
    RT_HIP(c, hipMemcpyAsync(v, c->stats.p, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->last_stats.rays = v[0];
    c->last_stats.primary_hits = v[1];
    c->last_stats.shadow_rays = v[2];
    c->last_stats.shadow_hits = v[3];
    c->last_stats.cells_visited = v[4];
    c->last_stats.tri_tests = v[5];
    c->last_stats.box_tests = v[6];
    c->last_stats.cell_records_fetched = v[7];
    c->last_stats.ref_records_fetched = v[8];
    c->last_stats.ref_records_staged = v[9];
    c->last_stats.cell_records_uniform = v[10];
    return RT_OK;
}

// True when the trace kernels cannot address the grid through buffer descriptors (32-bit byte offsets).
bool wide_grid(const DevScene& s) {
    return ((uint64_t)s.n_refs + 1) * 48 > 0xFFFFFFFFull || (uint64_t)s.n_bordered * 8 > 0xFFFFFFFFull;
}

constexpr uint32_t PLAN_CAP_HEAVY = 1024;  // most 8x8 blocks a plan traces four lanes per ray: that many workgroups
constexpr uint32_t PLAN_CAP_EARLY = 4096;  // most it moves to the front as they are: a quarter as many workgroups
constexpr uint32_t PLAN_LEAD_WGS = PLAN_CAP_HEAVY + PLAN_CAP_EARLY / 4;  // lead every planned launch
constexpr size_t PLAN_MAX_SLOTS = 8;       // (stream, frame shape) pairs remembered per context
constexpr size_t CULL_MAX_SLOTS = 8;       // streams with a block mask of their own, per context

// The plan for this stream and frame shape, made on first use (which allocates: the entry point that must not allocate
// never comes here). *plan stays null when there is none to be had; the caller then launches the plain kernel.
int plan_for(rt_ctx* c, const ShardMap& m, const DevFrame& f, uint32_t n_blocks, bool shadow, rt_plan** plan) {
    *plan = nullptr;
    const uint32_t key[8] = {f.width, f.height, m.tile_w, m.tile_h, m.first, m.stride, m.n_local, shadow ? 1u : 0u};
    rt_plan* hit = nullptr;
    for (rt_plan* p : c->plans)
        if (p->stream == c->stream && p->offsets == m.offsets && p->n_blocks == n_blocks && !memcmp(p->key, key, sizeof(key))) hit = p;
    if (!hit) {
        if (c->plans.size() < PLAN_MAX_SLOTS) {
            rt_plan* fresh = new rt_plan;
            hit = fresh;
            c->plans.push_back(hit);
        } else {  // the slot used longest ago; nothing may still be reading it
            for (rt_plan* p : c->plans)
                if (!hit || p->last_use < hit->last_use) hit = p;
            RT_HIP(c, hipDeviceSynchronize());
        }
        hit->stream = c->stream;
        memcpy(hit->key, key, sizeof(key));
        hit->offsets = m.offsets;
        hit->n_blocks = n_blocks;
        hit->cur = 0;
        const size_t bytes = BlockPlan::HEADER + ((size_t)PLAN_CAP_HEAVY + PLAN_CAP_EARLY) * sizeof(uint32_t) + n_blocks;
        for (rt::DevBuf& b : hit->buf) {
            RT_HIP(c, b.reserve(bytes));
            RT_HIP(c, hipMemsetAsync(b.p, 0, bytes, c->stream));  // an empty plan: no block listed
        }
    }
    hit->last_use = ++c->plan_clock;
    c->last_plan = hit;
    *plan = hit;
    return RT_OK;
}

BlockPlan make_block_plan(const rt_plan* p, uint32_t threshold_us) {
    BlockPlan b;
    b.cur = static_cast<const uint8_t*>(p->buf[p->cur].p);
    b.next = static_cast<uint8_t*>(p->buf[(p->cur + 1) % 3].p);
    b.after = static_cast<uint8_t*>(p->buf[(p->cur + 2) % 3].p);
    b.cap_heavy = PLAN_CAP_HEAVY;
    b.cap_early = PLAN_CAP_EARLY;
    b.bytes = (uint32_t)(BlockPlan::HEADER + ((size_t)PLAN_CAP_HEAVY + PLAN_CAP_EARLY) * sizeof(uint32_t) + p->n_blocks);
    b.threshold = threshold_us < 40000000u ? threshold_us * 100u : 0xFFFFFFFFu;  // microseconds -> 10 ns ticks
    return b;
}

// mask words a wave of k_block_mask can hold in LDS beside its list of occupied cells (64 KB a workgroup): frames of up
// to 491 520 blocks of 8x8 pixels, e.g. 7680 x 4096; a 7680 x 4320 frame (16 200 words) gets no mask
constexpr size_t CULL_MAX_WORDS = 15360;
constexpr uint32_t CULL_MAX_CELLS = 1u << 21;  // beyond this many grid cells a frame's mask costs more than it saves

// Inverse of the 3x3 matrix with columns c0, c1, c2, as rows; false when it is singular (or not finite).
static bool invert_columns(const double c0[3], const double c1[3], const double c2[3], double r0[3], double r1[3], double r2[3]) {
    auto cross3 = [](const double a[3], const double b[3], double o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1], o[1] = a[2] * b[0] - a[0] * b[2], o[2] = a[0] * b[1] - a[1] * b[0];
    };
    cross3(c1, c2, r0), cross3(c2, c0, r1), cross3(c0, c1, r2);
    const double det = c0[0] * r0[0] + c0[1] * r0[1] + c0[2] * r0[2];
    auto len = [](const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); };
    const double scale = len(c0) * len(c1) * len(c2);
    if (!(fabs(det) > 1e-9 * scale) || !(scale < 1e300)) return false;
    for (int k = 0; k < 3; ++k) r0[k] /= det, r1[k] /= det, r2[k] /= det;
    return true;
}

// How many OTHER streams of the context have been given a trace since the last rt_sync (or scene change): the frames
// the one being queued can overlap. Host state only - the same call sequence takes the same decisions on every run.
static uint32_t other_streams_tracing(const rt_ctx* c) {
    uint32_t n = 0;
    for (hipStream_t s : c->traced_on)
        if (s != c->stream) ++n;
    return n;
}

// After the host has waited for everything queued on `s`: no frame is in flight there any more.
static void stream_drained(rt_ctx* c, hipStream_t s) {
    for (size_t k = 0; k < c->traced_on.size(); ++k)
        if (c->traced_on[k] == s) {
            c->traced_on.erase(c->traced_on.begin() + (long)k);
            return;
        }
}

// After a trace launch on the context's stream.
static int mark_in_flight(rt_ctx* c) {
    for (hipStream_t s : c->traced_on)
        if (s == c->stream) return RT_OK;
    if (c->traced_on.size() < 64) c->traced_on.push_back(c->stream);
    return RT_OK;
}

// Whether this frame can have a block mask (see cull_masks_for), and what k_block_mask needs for it.
static bool cull_frame(const DevScene& s, const DevFrame& f, const ShardMap& m, CullFrame* cf) {
    if ((m.tile_w & 7u) && !(m.tiles_x == 1)) return false;
    if ((m.tile_h & 7u) && !(m.tiles_y == 1)) return false;
    if (s.n_cells == 0 || s.n_cells > CULL_MAX_CELLS || !(s.cell_size > 0.0f)) return false;
    const double cs = s.cell_size;
    // the whole grid and the camera on a scale where float positions resolve far below a cell
    double coord = 0.0, far2 = 0.0;
    for (int corner = 0; corner < 8; ++corner) {
        double d2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double q = (corner >> k) & 1 ? s.bmax[k] : s.bmin[k];
            coord = fmax(coord, fabs(q));
            d2 += (q - f.cam_pos[k]) * (q - f.cam_pos[k]);
        }
        far2 = fmax(far2, d2);
    }
    for (int k = 0; k < 3; ++k) coord = fmax(coord, fabs((double)f.cam_pos[k]));
    const double far = sqrt(far2) + 2.0 * cs;
    const double steps = fmin((double)s.max_steps, (double)s.gx + s.gy + s.gz) + 8.0;
    const double ulp = 1.0 / 16777216.0;
    // drift of the DDA's boundary crossings, and of the positions they are computed from, in cells
    if (!(steps * ulp * far / cs < 0.05) || !(16.0 * ulp * coord / cs < 0.01)) return false;
    // A start cell outside the grid sends the traversal down its literal path, whose wrapped cell index follows no ray
    // (rt_trace.hpp). An entering ray starts at most at fl(bmax - 1e-4) (raytracer_unified.wgsl:483-488), and the start
    // coordinate is monotonic in the position: if even that one is inside the grid, every entering ray's is.
    {
        const uint32_t g[3] = {s.gx, s.gy, s.gz};
        for (int k = 0; k < 3; ++k) {
            const float top = s.bmax[k] - 0.0001f;
            const float rel = top - s.bmin[k];
            if (f2u_sat(fmaxf(0.0f, floorf(rel * s.inv_cell_size))) >= g[k]) return false;
        }
    }
    // ... and so does an origin inside the bounds whose own start cell is outside the grid
    bool inside = true;
    for (int k = 0; k < 3; ++k) inside = inside && !(f.cam_pos[k] < s.bmin[k] || f.cam_pos[k] > s.bmax[k]);
    if (inside) {
        const uint32_t g[3] = {s.gx, s.gy, s.gz};
        for (int k = 0; k < 3; ++k) {
            const float rel = f.cam_pos[k] - s.bmin[k];
            if (f2u_sat(fmaxf(0.0f, floorf(rel * s.inv_cell_size))) >= g[k]) return false;
        }
    }
    const double aspect = (double)((float)f.width / (float)f.height), fov = (double)f.fov_scale;
    double c0[3], c1[3], c2[3];
    for (int k = 0; k < 3; ++k) {
        cf->o[k] = f.cam_pos[k];
        c0[k] = f.fwd[k];
        c1[k] = (double)f.right[k] * aspect * fov;
        c2[k] = -(double)f.up[k] * fov;
        if (!std::isfinite(cf->o[k]) || !std::isfinite(c0[k]) || !std::isfinite(c1[k]) || !std::isfinite(c2[k])) return false;
    }
    if (!invert_columns(c0, c1, c2, cf->r0, cf->r1, cf->r2)) return false;
    double rn = 0.0;
    for (int k = 0; k < 3; ++k) rn = fmax(rn, fmax(fabs(cf->r1[k]), fabs(cf->r2[k])));
    cf->tol = 1e-9 * (far + coord) * 3.0 * rn;
    cf->near_r = 1e-6 * cs;
    cf->margin = 0.25 * cs + 0.002;
    cf->width = f.width, cf->height = f.height;
    cf->blocks_x = (f.width + 7u) >> 3, cf->blocks_y = (f.height + 7u) >> 3;
    cf->words_per_row = (cf->blocks_x + 31u) >> 5;
    if ((size_t)cf->blocks_y * cf->words_per_row > CULL_MAX_WORDS) return false;
    return true;
}

// The block masks for the n frames of a launch (see rt_cull.hpp), `*words` mask words apart, queued on the context's
// stream ahead of the launch; *mask stays null when they get none: a parameter says so, nothing else of the context is
// running (mode 1), the scene has boxes, the tiles do not line up with the frame's 8x8 blocks, the float DDA's drift
// cannot be bounded tightly enough for a camera and the grid, a camera basis is degenerate, an origin is one that starts
// the traversal outside the grid, or the grid is so large that making the masks costs more than it saves.
int cull_masks_for(rt_ctx* c, const DevScene& s, const DevFrame* fs, uint32_t n, const ShardMap& m, const uint32_t** mask,
                   uint32_t* words_out, bool always) {
    *mask = nullptr;
    if (!c->params.cull_empty_blocks || s.nb != 0 || n == 0 || n > CULL_MAX_FRAMES) return RT_OK;
    // 1: only where it pays. The mask is one more launch the trace has to wait for, 9 us alone and 20-30 us beside other
    // streams' work. A group of frames in one launch shares that wait; with frames given to two or more other streams
    // it is hidden and the traversals saved are time gained (three in flight: 67 -> 62 us per frame); with one other
    // stream it is break-even for hit records and a loss for the shaded picture, and a frame traced on an idle GPU
    // waits for its few longest waves whatever the others do, and would only start later.
    if (c->params.cull_empty_blocks == 1 && !always && n < 4 && other_streams_tracing(c) < 2) return RT_OK;
    CullFrames cfs;
    for (uint32_t j = 0; j < n; ++j)
        if (!cull_frame(s, fs[j], m, &cfs.f[j])) return RT_OK;
    const CullFrame& cf = cfs.f[0];
    const size_t words = (size_t)cf.blocks_y * cf.words_per_row, total = words * n;

    rt_cull* hit = nullptr;
    for (rt_cull* k : c->culls)
        if (k->stream == c->stream) hit = k;
    if (!hit) {
        if (c->culls.size() < CULL_MAX_SLOTS) {
            hit = new rt_cull;
            c->culls.push_back(hit);
        } else {  // the slot used longest ago; nothing may still be reading it
            for (rt_cull* k : c->culls)
                if (!hit || k->last_use < hit->last_use) hit = k;
            RT_HIP(c, hipDeviceSynchronize());
        }
        hit->stream = c->stream;
    }
    if (total * sizeof(uint32_t) > hit->buf[0].cap) {
        // a larger frame on this stream: the old buffers may still be read by a launch queued before this one
        RT_HIP(c, hipStreamSynchronize(c->stream));
        for (int b = 0; b < 2; ++b) {
            RT_HIP(c, hit->buf[b].reserve(total * sizeof(uint32_t)));
            RT_HIP(c, hipMemsetAsync(hit->buf[b].p, 0, hit->buf[b].cap, c->stream));
            hit->dirty[b] = 0;
        }
    }
    const int fill = hit->cur ^ 1;
    if (hit->dirty[fill]) {  // the trace launch that would have zeroed it did not happen (a frame without a mask in between)
        RT_HIP(c, hipMemsetAsync(hit->buf[fill].p, 0, (size_t)hit->dirty[fill] * sizeof(uint32_t), c->stream));
        hit->dirty[fill] = 0;
    }
    hit->last_use = ++c->plan_clock;
    hit->words = (uint32_t)words;  // of the first frame of the launch: what rt_get_cull_stats reports
    hit->blocks = cf.blocks_x * cf.blocks_y;
    const uint32_t waves = (s.n_cells + CULL_CELLS_PER_WAVE - 1u) / CULL_CELLS_PER_WAVE;
    if (n == 1)
        hipLaunchKernelGGL(k_block_mask, dim3(waves), dim3(64), words * sizeof(uint32_t), c->stream, s.cells, s.gx, s.gy, s.gz, s.px,
                           s.pxy, s.bmin[0], s.bmin[1], s.bmin[2], s.cell_size, cf, hit->buf[fill].as<uint32_t>());
    else
        hipLaunchKernelGGL(k_block_masks, dim3(waves, n), dim3(64), words * sizeof(uint32_t), c->stream, s.cells, s.gx, s.gy, s.gz, s.px,
                           s.pxy, s.bmin[0], s.bmin[1], s.bmin[2], s.cell_size, cfs, hit->buf[fill].as<uint32_t>(), (uint32_t)words);
    RT_HIP(c, hipGetLastError());
    hit->dirty[fill] = (uint32_t)total;
    hit->cur = fill;
    c->last_cull = hit;
    *mask = hit->buf[fill].as<uint32_t>();
    if (words_out) *words_out = (uint32_t)words;
    return RT_OK;
}

int cull_mask_for(rt_ctx* c, const DevScene& s, const DevFrame& f, const ShardMap& m, const uint32_t** mask, bool always) {
    return cull_masks_for(c, s, &f, 1, m, mask, nullptr, always);
}

// The block-queue counter of the current stream (rt_params.persistent_waves); *queue stays null when eight other streams
// hold the context's counters already: the caller then launches the plain grid.
int queue_for(rt_ctx* c, uint32_t** queue) {
    *queue = nullptr;
    rt_queues& q = c->queues;
    if (!q.counters.p) {
        RT_HIP(c, q.counters.reserve(8 * 128));
        RT_HIP(c, hipMemsetAsync(q.counters.p, 0, 8 * 128, c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));  // once per context: every stream's counter starts at zero
        hipDeviceProp_t prop;
        RT_HIP(c, hipGetDeviceProperties(&prop, c->device));
        q.resident_wgs = (uint32_t)prop.multiProcessorCount * 6u;  // 4 SIMDs x 6 waves (the kernel holds up to 75 vector registers) = 6 workgroups of 4 waves per CU
    }
    int i = 0;
    while (i < q.used && q.streams[i] != c->stream) ++i;
    if (i == q.used) {
        if (q.used == 8) return RT_OK;
        q.streams[q.used++] = c->stream;
    }
    *queue = q.counters.as<uint32_t>() + 32 * i;
    return RT_OK;
}

// Kernel variants: WGSL (constant set), SHADOW / ANY, BOXES (scene has boxes), TEX (hits decided by a texture's
// alpha). The counting build is the general one (with the box path); so is the build for grids of 4 GiB and more.
template <bool WGSL, bool SHADOW>
int launch_primary(rt_ctx* c, int stats, dim3 grid, const DevScene& s, const DevFrame& f, const ShardMap& m, uint2* op,
                   uint2* os, bool may_alloc) {
    unsigned long long* st = c->stats.as<unsigned long long>();
    c->last_cull = nullptr;
#ifdef RT_EXPERIMENTS
    if (const char* v = getenv("RT_VARIANT")) {
        if (!WGSL && SHADOW && !stats && s.nb == 0) {
            const int vn = atoi(v);
            if (vn >= 40 && vn < 60) {  // persistent waves: one wave per wave slot of the chip
                void* q0 = nullptr;
                if (hipGetSymbolAddress(&q0, HIP_SYMBOL(rt::g_queue)) != hipSuccess) return RT_OK;
                static hipStream_t q_streams[8];  // a counter per launching stream: frames in flight must not share one
                static int q_used = 0;
                int qi = 0;
                while (qi < q_used && q_streams[qi] != c->stream) ++qi;
                if (qi == q_used) {
                    if (q_used == 8) return fail(c, RT_ERR_LIMIT, "persistent variants: more than 8 streams");
                    q_streams[q_used++] = c->stream;
                }
                uint32_t* q = static_cast<uint32_t*>(q0) + 32 * qi;
                (void)hipMemsetAsync(q, 0, sizeof(uint32_t), c->stream);
                const uint32_t n_waves = grid.x * 4;
                switch (vn) {
                    case 41: hipLaunchKernelGGL((k_trace_primary_persist<41>), dim3(2048), dim3(256), 0, c->stream, s, f, m, op, os, n_waves, q); break;
                    case 44: hipLaunchKernelGGL((k_trace_primary_persist<44>), dim3(2048), dim3(256), 0, c->stream, s, f, m, op, os, n_waves, q); break;
                    case 40: hipLaunchKernelGGL((k_trace_primary_persist<40>), dim3(2048), dim3(256), 0, c->stream, s, f, m, op, os, n_waves, q); break;
                    case 54: hipLaunchKernelGGL((k_trace_primary_persist<54>), dim3(2048), dim3(256), 0, c->stream, s, f, m, op, os, n_waves, q); break;
                    default: hipLaunchKernelGGL((k_trace_primary_persist<50>), dim3(2048), dim3(256), 0, c->stream, s, f, m, op, os, n_waves, q); break;
                }
                return RT_OK;
            }
            if (vn == 80) {  // four lanes per ray, every block
                hipLaunchKernelGGL(k_trace_primary_quad_all, dim3(grid.x * 4), dim3(256), 0, c->stream, s, f, m, op, os);
                return RT_OK;
            }
            if (vn == 72) {  // the shipped kernel with the frame's block mask, whatever the parameter says
                const uint32_t* mask = nullptr;
                if (int r = cull_mask_for(c, s, f, m, &mask, true)) return r;
                hipLaunchKernelGGL((k_trace_primary_lean<WGSL, SHADOW>), grid, dim3(256), 0, c->stream, s, f, m, op, os, mask, (uint32_t*)nullptr, 0u);
                return RT_OK;
            }
            if (vn == 71) {  // the shipped kernel without a plan of expensive blocks
                const uint32_t* no_mask = nullptr;
                hipLaunchKernelGGL((k_trace_primary_lean<WGSL, SHADOW>), grid, dim3(256), 0, c->stream, s, f, m, op, os, no_mask, (uint32_t*)nullptr, 0u);
                return RT_OK;
            }
            if (vn == 73) {  // the general kernel's instantiation for the same scenes: 86 scalar registers, seven waves per SIMD
                hipLaunchKernelGGL((k_trace_primary<WGSL, SHADOW, 0, false>), grid, dim3(256), 0, c->stream, s, f, m, op, os, st, (const uint32_t*)nullptr);
                return RT_OK;
            }
            switch (vn) {
#define RT_VAR(N) case N: hipLaunchKernelGGL((k_trace_primary_var<N>), dim3(grid.x * (256 / VarCfg<N>::threads)), dim3(VarCfg<N>::threads), 0, c->stream, s, f, m, op, os); return RT_OK;
                RT_VARIANT_LIST
#undef RT_VAR
            }
        }
    }
#endif
    // the two counting builds (1: the reference's work, 2: this library's fetches) are the general kernels; the second
    // follows the frame's block mask as the timed kernel does (a block that is not traced fetches nothing)
    const uint32_t* nm = nullptr;
    const bool count_repack = stats == 2 && c->params.repack_candidates && s.nb == 0 && c->tex_mode != 2 && !c->params.heavy_block_threshold && !wide_grid(s);
    if (stats == 2 && may_alloc && c->tex_mode != 2 && !wide_grid(s) && !count_repack)
        if (int r = cull_mask_for(c, s, f, m, &nm, true)) return r;
    if (wide_grid(s)) {
        if (stats == 2)
            hipLaunchKernelGGL((k_trace_primary<WGSL, SHADOW, 2, true, true, true>), grid, dim3(256), 0, c->stream, s, f, m, op, os, st, nm);
        else if (stats)
            hipLaunchKernelGGL((k_trace_primary<WGSL, SHADOW, 1, true, true, true>), grid, dim3(256), 0, c->stream, s, f, m, op, os, st, nm);
        else
            hipLaunchKernelGGL((k_trace_primary<WGSL, SHADOW, 0, true, true, true>), grid, dim3(256), 0, c->stream, s, f, m, op, os, st, nm);
    } else if (count_repack) {
        // the counting build of the kernel such a frame is traced with (no mask there, no staging: pairs are dealt)
        hipLaunchKernelGGL((k_trace_primary_repack<WGSL, SHADOW, true>), grid, dim3(256), 0, c->stream, s, f, m, op, os, st);
    } else if (stats == 2) {
        hipLaunchKernelGGL((k_trace_primary<WGSL, SHADOW, 2, true, true>), grid, dim3(256), 0, c->stream, s, f, m, op, os, st, nm);
    } else if (stats) {
        hipLaunchKernelGGL((k_trace_primary<WGSL, SHADOW, 1, true, true>), grid, dim3(256), 0, c->stream, s, f, m, op, os, st, nm);
    } else if (c->tex_mode == 2) {  // hits decided by a texture's alpha: the general kernel with the material path
        if (s.nb != 0)
            hipLaunchKernelGGL((k_trace_primary<WGSL, SHADOW, 0, true, true>), grid, dim3(256), 0, c->stream, s, f, m, op, os, st, nm);
        else
            hipLaunchKernelGGL((k_trace_primary<WGSL, SHADOW, 0, false, true>), grid, dim3(256), 0, c->stream, s, f, m, op, os, st, nm);
    } else if (s.nb != 0)
        hipLaunchKernelGGL((k_trace_primary<WGSL, SHADOW, 0, true>), grid, dim3(256), 0, c->stream, s, f, m, op, os, st, nm);
    else {
        // triangles only, no counters: the kernel that keeps a plan of expensive pixel blocks, when it may
        rt_plan* plan = nullptr;
        if (may_alloc && c->params.heavy_block_threshold && (uint64_t)grid.x * 4 + PLAN_LEAD_WGS < 0x7FFFFFFFull)
            if (int r = plan_for(c, m, f, grid.x * 4, SHADOW, &plan)) return r;
        uint32_t* queue = nullptr;
        if (!plan && !c->params.repack_candidates && c->params.persistent_waves && may_alloc)
            if (int r = queue_for(c, &queue)) return r;
        if (queue) {
            // persistent waves: as many workgroups as fit the device at once work the block queue off; the launch leaves
            // the queue's counter at zero for the next one on this stream
            const uint32_t* mask = nullptr;
            if (int r = cull_mask_for(c, s, f, m, &mask, false)) return r;
            QueueArgs a;
            a.s = s, a.f = f, a.m = m, a.out_p = op, a.out_s = os, a.cull_mask = mask, a.queue = queue, a.n_waves = grid.x * 4u;
            a.cull_zero = nullptr, a.cull_zero_words = 0;
            if (mask) {  // the other buffer of the mask pair, for the next frame's mask: this launch zeroes it
                rt_cull* k = c->last_cull;
                a.cull_zero = k->buf[k->cur ^ 1].as<uint32_t>();
                a.cull_zero_words = k->dirty[k->cur ^ 1];
                k->dirty[k->cur ^ 1] = 0;
            }
            const uint32_t wgs = grid.x < c->queues.resident_wgs ? grid.x : c->queues.resident_wgs;
            hipLaunchKernelGGL((k_trace_primary_queue<WGSL, SHADOW>), dim3(wgs), dim3(256), 0, c->stream, a);
            if (c->params.cull_empty_blocks == 1)
                if (int r = mark_in_flight(c)) return r;
        } else if (!plan && c->params.repack_candidates) {
            // rays of a wave in different cells: (ray, candidate) pairs dealt out again to the lanes (trace_ray_repack)
            hipLaunchKernelGGL((k_trace_primary_repack<WGSL, SHADOW>), grid, dim3(256), 0, c->stream, s, f, m, op, os, (unsigned long long*)nullptr);
        } else if (plan) {
            const BlockPlan bp = make_block_plan(plan, c->params.heavy_block_threshold);
            hipLaunchKernelGGL((k_trace_primary_planned<WGSL, SHADOW>), dim3(grid.x + PLAN_LEAD_WGS), dim3(256), 0, c->stream, s, f, m, bp, op, os);
            plan->cur = (plan->cur + 1) % 3;
        } else {
            // the frame's mask of pixel blocks that can meet geometry, when one can be made (and may be allocated)
            const uint32_t* mask = nullptr;
            if (may_alloc)
                if (int r = cull_mask_for(c, s, f, m, &mask, false)) return r;
            uint32_t* zero = nullptr;  // the other buffer of the pair, for the next frame's mask: this launch zeroes it
            uint32_t zero_words = 0;
            if (mask) {
                rt_cull* k = c->last_cull;
                zero = k->buf[k->cur ^ 1].as<uint32_t>();
                zero_words = k->dirty[k->cur ^ 1];
                k->dirty[k->cur ^ 1] = 0;
            }
            hipLaunchKernelGGL((k_trace_primary_lean<WGSL, SHADOW>), grid, dim3(256), 0, c->stream, s, f, m, op, os, mask, zero, zero_words);
            if (may_alloc && c->params.cull_empty_blocks == 1)
                if (int r = mark_in_flight(c)) return r;
        }
    }
    return RT_OK;
}

// Ray lists of up to 32 768 rays (a quarter of the chip's wave slots at four lanes each) are traced four lanes to a ray:
// with so few waves the four walks of the grid cost nothing, and where lists are long the chain is what the call waits
// for (bundled static model, 1 000 / 20 000 rays: 99 -> 50 us / 132 -> 63 us; the generator's scene on its tuned grid,
// short lists: 87 -> 84 / 99 -> 100). Beyond that it depends on the scene (100 000 rays: 134 -> 70 us on the model,
// 96 -> 119 on the generator's scene), so a lane to a ray stays. RT_RAY_LANES=1 / 4 (development, tests): that form
// whatever the count.
static bool rays_take_four_lanes(uint32_t n) {
    const char* lanes = getenv("RT_RAY_LANES");
    return n != 0 && (lanes ? atoi(lanes) == 4 : n <= 32768u);
}

template <bool WGSL, bool ANY>
void launch_rays(rt_ctx* c, int stats, dim3 grid, const DevScene& s, const float* o, const float* d, uint32_t n,
                 uint2* out) {
    unsigned long long* st = c->stats.as<unsigned long long>();
    if (wide_grid(s)) {
        if (stats == 2)
            hipLaunchKernelGGL((k_trace_rays<WGSL, ANY, 2, true, true, true>), grid, dim3(256), 0, c->stream, s, o, d, n, out, st);
        else if (stats)
            hipLaunchKernelGGL((k_trace_rays<WGSL, ANY, 1, true, true, true>), grid, dim3(256), 0, c->stream, s, o, d, n, out, st);
        else
            hipLaunchKernelGGL((k_trace_rays<WGSL, ANY, 0, true, true, true>), grid, dim3(256), 0, c->stream, s, o, d, n, out, st);
    } else if (stats == 2) {
        hipLaunchKernelGGL((k_trace_rays<WGSL, ANY, 2, true, true>), grid, dim3(256), 0, c->stream, s, o, d, n, out, st);
    } else if (stats) {
        hipLaunchKernelGGL((k_trace_rays<WGSL, ANY, 1, true, true>), grid, dim3(256), 0, c->stream, s, o, d, n, out, st);
    } else if (c->tex_mode == 2) {
        if (s.nb != 0)
            hipLaunchKernelGGL((k_trace_rays<WGSL, ANY, 0, true, true>), grid, dim3(256), 0, c->stream, s, o, d, n, out, st);
        else
            hipLaunchKernelGGL((k_trace_rays<WGSL, ANY, 0, false, true>), grid, dim3(256), 0, c->stream, s, o, d, n, out, st);
    } else if (s.nb != 0)
        hipLaunchKernelGGL((k_trace_rays<WGSL, ANY, 0, true>), grid, dim3(256), 0, c->stream, s, o, d, n, out, st);
    else if (c->params.repack_candidates && n != 0)
        hipLaunchKernelGGL((k_trace_rays_repack<WGSL, ANY>), grid, dim3(256), 0, c->stream, s, o, d, n, out);
    else if (rays_take_four_lanes(n))  // a list that leaves the chip idle behind its rays' dependent chains
        hipLaunchKernelGGL((k_trace_rays_quad<WGSL, ANY>), dim3((uint32_t)(((size_t)n * 4 + 255) / 256)), dim3(256), 0, c->stream, s, o, d,
                           n, out);
    else
        hipLaunchKernelGGL((k_trace_rays<WGSL, ANY, 0, false>), grid, dim3(256), 0, c->stream, s, o, d, n, out, st);
}

// Stage a host array on the device (or pass a device pointer through).
int stage_in(rt_ctx* c, DevBuf& buf, const void* src, size_t bytes, const void** dev) {
    if (is_device_ptr(src)) {
        *dev = src;
        return RT_OK;
    }
    RT_HIP(c, buf.reserve(bytes));
    RT_HIP(c, hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    *dev = buf.p;
    return RT_OK;
}

}  // namespace

namespace rt {

// (first, count) records of the bordered cell array from the trace lists; reserves the reference records.
int emit_cells(rt_ctx* c) {
    const uint32_t* g = c->meta.grid_sizes[RT_GRID_LEVELS - 1];
    const unsigned long long bordered = (g[0] + 2ull) * (g[1] + 2ull) * (g[2] + 2ull);
    if (bordered > 0xFFFFFFFFull)
        return fail(c, RT_ERR_LIMIT, "grid of %u x %u x %u cells: the bordered cell array has %llu entries (max 2^32-1)",
                    g[0], g[1], g[2], bordered);
    RT_HIP(c, c->cells.reserve((size_t)bordered * sizeof(uint2)));
    RT_HIP(c, c->refs.reserve(((size_t)c->n_refs_t * 3 + 3) * sizeof(float4)));
    hipLaunchKernelGGL(k_emit_cells, dim3((unsigned)((bordered + 255) / 256)), dim3(256), 0, c->stream,
                       c->t_cell_start, g[0], g[1], g[2], c->params.max_objects_per_cell, c->cells.as<uint2>());
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

// A new scene in three steps, shared by rt_set_scene (host arrays) and rt_broadcast_scene (arrays that arrive from
// another GPU): begin_scene waits for readers of the old scene and forgets what depended on it; reserve_scene sizes the
// buffers - boxes as they are, triangles as 80-byte wire records in `misc`; finish_scene unpacks the triangle records
// on the device (36-byte vertex records kept: the rest - uv, material - is not geometry) and waits.
int begin_scene(rt_ctx* c) {
    if (int r = scene_barrier(c)) return r;
    c->has_grid = false;
    c->dup_state = -1;
    c->has_skin = false;  // a skin belongs to the triangles it was set for
    c->has_bvh = false;
    c->fast_ready = false;
    return RT_OK;
}

int reserve_scene(rt_ctx* c, uint32_t nb, uint32_t nt) {
    RT_HIP(c, c->boxes.reserve((size_t)nb * sizeof(BoxData) + 16));
    RT_HIP(c, c->tri_v.reserve((size_t)nt * 9 * sizeof(float) + 16));
    if (nt) {
        RT_HIP(c, c->misc.reserve((size_t)nt * sizeof(TriangleData)));
        RT_HIP(c, c->tri_mat.reserve((size_t)nt * sizeof(uint32_t)));
        RT_HIP(c, c->tri_uv.reserve((size_t)nt * 6 * sizeof(float)));
        RT_HIP(c, c->tri_n.reserve((size_t)nt * sizeof(float4)));
    }
    return RT_OK;
}

int finish_scene(rt_ctx* c, uint32_t nb, uint32_t nt) {
    c->nb = nb;
    c->nt = nt;
    if (nt) {
        hipLaunchKernelGGL(k_tri_unpack, dim3((nt + 255) / 256), dim3(256), 0, c->stream, c->misc.as<TriangleData>(), nt,
                           c->tri_v.as<float>(), c->tri_mat.as<uint32_t>(), c->tri_uv.as<float>());
        RT_HIP(c, hipGetLastError());
        hipLaunchKernelGGL(k_tri_normals, dim3((nt + 255) / 256), dim3(256), 0, c->stream, c->tri_v.as<float>(), 0u, nt,
                           c->tri_n.as<float4>());
        RT_HIP(c, hipGetLastError());
    }
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

}  // namespace rt

// ================================================================== lifecycle

extern "C" void rt_default_params(rt_params* p) {
    if (!p) return;
    p->finest_cell_size = RT_FINEST_CELL_SIZE;
    p->max_objects_per_cell = RT_MAX_OBJECTS_PER_CELL;
    p->max_dda_steps = 200;
    p->wgsl_compat = 0;
    p->lod_cull = 1;
    p->moving_box_count = 3;
    p->fov = 0.785398f;
    p->light_direction[0] = 0.5f;
    p->light_direction[1] = -1.0f;
    p->light_direction[2] = 0.3f;
    p->shadow_offset = 0.001f;
    p->collect_stats = 0;
    p->skip_duplicate_triangles = 1;
    p->heavy_block_threshold = 0;
    p->cull_empty_blocks = 1;
    p->repack_candidates = 0;
    p->persistent_waves = 0;
}

extern "C" const char* rt_create_error(void) { return g_create_error.c_str(); }

extern "C" int rt_create(rt_ctx** out, int device_id, const rt_params* params) {
    g_create_error.clear();
    if (!out) {
        g_create_error = "rt_create: out is NULL";
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        g_create_error = std::string("rt_create: no HIP device (") + hipGetErrorString(e) + ")";
        return RT_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        g_create_error = "rt_create: device_id out of range";
        return RT_ERR_INVALID;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
        g_create_error = std::string("rt_create: hipGetDeviceProperties: ") + hipGetErrorString(e);
        return RT_ERR_NO_DEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("rt_create: device is ") + prop.gcnArchName + ", this library holds gfx950 code only";
        return RT_ERR_NO_DEVICE;
    }
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        g_create_error = std::string("rt_create: hipSetDevice: ") + hipGetErrorString(e);
        return RT_ERR_NO_DEVICE;
    }
    if ((e = scratch_for_device(device_id)) != hipSuccess) {
        g_create_error = std::string("rt_create: hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
        return RT_ERR_NO_DEVICE;
    }
    rt_ctx* c = new rt_ctx();
    c->device = device_id;
    if (params)
        c->params = *params;
    else
        rt_default_params(&c->params);
    memset(&c->meta, 0, sizeof(c->meta));
    memset(&c->last_stats, 0, sizeof(c->last_stats));
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_scratch, hipEventDisableTiming)) != hipSuccess) {
        g_create_error = std::string("rt_create: stream/event creation: ") + hipGetErrorString(e);
        delete c;
        return RT_ERR_NO_DEVICE;
    }
    c->stream = c->own_stream;
    *out = c;
    return RT_OK;
}

extern "C" void rt_destroy(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)rt_comm_destroy(c);
    c->queues.counters.release();
    DevBuf* bufs[] = {&c->boxes,     &c->tri_v,     &c->tri_n,     &c->tri_mat,   &c->materials, &c->skin_pos, &c->skin_joints,
                      &c->skin_weights, &c->skin_tri, &c->skin_mats, &c->bvh_nodes, &c->bvh_leaf_first, &c->bvh_prims, &c->bvh_recs, &c->bvh_work, &c->bvh_parts,     &c->cell_start, &c->indices,   &c->cells,       &c->refs,
                      &c->box_frame, &c->b_partial, &c->b_counts,   &c->b_tmp_start, &c->b_cursor,  &c->b_tmp_ids,
                      &c->b_tmp_sorted, &c->b_tmp_cell, &c->b_scan_tmp, &c->b_flags, &c->b_obj_n, &c->b_obj_off, &c->b_obj_sum, &c->out_p,     &c->out_s,
                      &c->in_o,      &c->in_d,      &c->stats,      &c->misc,      &c->tri_uv,    &c->texels,    &c->tc_start,  &c->tc_indices, &c->tc_pos, &c->dd_keys, &c->dd_keys2,
                      &c->dd_vals, &c->dd_vals2, &c->dd_flag, &c->dd_head, &c->dd_dup, &c->ref_cell, &c->tc_ref_cell, &c->fast_meta, &c->fast_scan};
    for (DevBuf* b : bufs) b->release();
    for (rt_table* t : c->tables) {
        t->buf.release();
        delete t;
    }
    for (rt_plan* p : c->plans) {
        for (DevBuf& b : p->buf) b.release();
        delete p;
    }
    for (rt_cull* k : c->culls) {
        k->buf[0].release();
        k->buf[1].release();
        delete k;
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    if (c->ev_scratch) {
        {  // this context's launches are done (synchronised above): nobody has to wait for its event any more
            std::lock_guard<std::mutex> hold(g_scratch_lock);
            if (c->device >= 0 && c->device < NOALLOC_MAX_DEVICES && g_scratch_user[c->device].ctx == c) g_scratch_user[c->device] = ScratchUser{};
        }
        (void)hipEventDestroy(c->ev_scratch);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->fast_landing) (void)hipHostFree(c->fast_landing);
    delete c;
}

extern "C" const char* rt_last_error(const rt_ctx* c) { return c ? c->err.c_str() : "null context"; }

// Switching streams does not block the host, so a caller can keep several frames in flight by alternating
// streams (the last waves of a frame - a few pixel blocks over the densest cells - then overlap the next frame's
// bulk). Ordering rules the context keeps by itself:
//   - work that uses the context's per-frame scratch (box records of scenes with boxes) is ordered across the
//     switch on the GPU: the new stream waits for it,
//   - calls that replace the scene or the grid first wait for every stream used before (scene_barrier).
// Output buffers are the caller's: a frame in flight must not share them with another.
extern "C" int rt_set_stream(rt_ctx* c, void* hip_stream) {
    if (int r = use(c)) return r;
    hipStream_t next = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    if (next == c->stream) return RT_OK;
    if (c->scratch_busy) {
        RT_HIP(c, hipEventRecord(c->ev_order, c->stream));
        RT_HIP(c, hipStreamWaitEvent(next, c->ev_order, 0));
        c->scratch_busy = false;
    }
    c->other_streams_busy = true;
    c->stream = next;
    return RT_OK;
}

extern "C" int rt_sync(rt_ctx* c) {
    if (int r = use(c)) return r;
    if (c->other_streams_busy) {  // frames may be in flight on streams used before the current one
        RT_HIP(c, hipDeviceSynchronize());
        c->other_streams_busy = false;
    } else {
        RT_HIP(c, wait_short(c->stream));  // a frame is a hundred microseconds: polled first, the driver's sleep after that
        (void)hipGetLastError();
    }
    c->traced_on.clear();  // nothing is in flight any more: the next frame is a lone one until others join it
    c->scratch_busy = false;
    return RT_OK;
}

extern "C" int rt_set_collect_stats(rt_ctx* c, int enabled) {
    if (int r = use(c)) return r;
    if (enabled < 0 || enabled > 2) return fail(c, RT_ERR_INVALID, "rt_set_collect_stats: mode %d (0, 1 or 2)", enabled);
    const uint32_t v = (uint32_t)enabled;
    if (v == c->params.collect_stats) return RT_OK;
    const bool full_lists_before = c->params.collect_stats == 1, full_lists_now = v == 1;
    c->params.collect_stats = v;
    // mode 1 walks the reference's full lists, modes 0 and 2 the lists with duplicate triangles left out
    if (full_lists_before != full_lists_now && c->has_grid && c->grid_built && c->params.skip_duplicate_triangles &&
        c->dup_state != 0) {
        if (int r = scene_barrier(c)) return r;
        RT_HIP(c, hipStreamSynchronize(c->stream));
        if (int r = rebuild_trace_records(c)) return r;
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}

extern "C" int rt_get_stats(rt_ctx* c, rt_stats* stats) {
    if (int r = use(c)) return r;
    if (!stats) return fail(c, RT_ERR_INVALID, "rt_get_stats: stats is NULL");
    *stats = c->last_stats;
    return RT_OK;
}

// reference: src/camera.rs:51-54,76-122
extern "C" void rt_camera_to_uniform(const float position[3], float yaw, float pitch, float time, float screen_height,
                                     float fov, int show_grid, CameraUniform* out) {
    if (!out) return;
    V3 fwd = normalize(v3(sinf(yaw) * cosf(pitch), sinf(pitch), cosf(yaw) * cosf(pitch)));
    V3 right = normalize(cross(fwd, v3(0.0f, 1.0f, 0.0f)));
    memset(out, 0, sizeof(*out));
    for (int k = 0; k < 3; ++k) out->position[k] = position[k];
    out->forward[0] = fwd.x, out->forward[1] = fwd.y, out->forward[2] = fwd.z;
    out->right[0] = right.x, out->right[1] = right.y, out->right[2] = right.z;
    out->up[0] = 0.0f, out->up[1] = 1.0f, out->up[2] = 0.0f;
    out->time = time;
    out->lod_factor = screen_height / (2.0f * tanf(fov / 2.0f));
    out->min_pixel_size = 2.0f;
    out->show_grid = show_grid ? 1.0f : 0.0f;
}

// ================================================================== scene and grid

extern "C" int rt_set_scene(rt_ctx* c, const BoxData* boxes, uint32_t nb, const TriangleData* tris, uint32_t nt) {
    if (int r = use(c)) return r;
    if ((nb && !boxes) || (nt && !tris)) return fail(c, RT_ERR_INVALID, "rt_set_scene: NULL array with non-zero count");
    if ((uint64_t)nb + nt >= 0xFFFFFFF0ull) return fail(c, RT_ERR_LIMIT, "rt_set_scene: more than 2^32-16 objects");
    if (int r = begin_scene(c)) return r;
    if (int r = reserve_scene(c, nb, nt)) return r;
    if (nb) RT_HIP(c, hipMemcpyAsync(c->boxes.p, boxes, (size_t)nb * sizeof(BoxData), hipMemcpyHostToDevice, c->stream));
    if (nt) RT_HIP(c, hipMemcpyAsync(c->misc.p, tris, (size_t)nt * sizeof(TriangleData), hipMemcpyHostToDevice, c->stream));
    return finish_scene(c, nb, nt);
}

extern "C" int rt_get_scene(rt_ctx* c, uint32_t* nb, uint32_t* nt) {
    if (int r = use(c)) return r;
    if (nb) *nb = c->nb;
    if (nt) *nt = c->nt;
    return RT_OK;
}

static int check_meta(rt_ctx* c, const GridMetadata* meta, size_t n_cells, const char* who) {
    if (!meta) return fail(c, RT_ERR_INVALID, "%s: meta is NULL", who);
    const uint32_t* g = meta->grid_sizes[RT_GRID_LEVELS - 1];
    uint64_t want = (uint64_t)g[0] * g[1] * g[2];
    if (want == 0 || want > 0x7FFFFFFFull) return fail(c, RT_ERR_LIMIT, "%s: fine grid of %llu cells", who, (unsigned long long)want);
    if (n_cells != (size_t)want)
        return fail(c, RT_ERR_INVALID, "%s: n_cells=%zu but grid_sizes says %llu", who, n_cells, (unsigned long long)want);
    if (!(meta->finest_cell_size > 0.0f)) return fail(c, RT_ERR_INVALID, "%s: finest_cell_size must be > 0", who);
    return RT_OK;
}

static int install_csr(rt_ctx* c, const GridMetadata* meta, const uint32_t* cell_start, const uint32_t* indices,
                       size_t n_cells) {
    uint32_t n_refs = cell_start[n_cells];
    for (size_t k = 0; k < n_cells; ++k)
        if (cell_start[k + 1] < cell_start[k]) return fail(c, RT_ERR_INVALID, "cell_start is not non-decreasing at %zu", k);
    if (int r = scene_barrier(c)) return r;
    c->meta = *meta;
    c->grid_nb = c->nb;
    c->n_cells = (uint32_t)n_cells;
    c->n_refs = n_refs;
    RT_HIP(c, c->cell_start.reserve((n_cells + 1) * sizeof(uint32_t)));
    RT_HIP(c, c->indices.reserve((size_t)n_refs * sizeof(uint32_t) + 16));
    RT_HIP(c, hipMemcpyAsync(c->cell_start.p, cell_start, (n_cells + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    if (n_refs)
        RT_HIP(c, hipMemcpyAsync(c->indices.p, indices, (size_t)n_refs * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    c->grid_built = false;  // the caller's lists: any order, any content - every entry is tested, none skipped
    if (int r = rebuild_trace_records(c)) return r;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->has_grid = true;
    return RT_OK;
}

extern "C" int rt_set_grid_csr(rt_ctx* c, const GridMetadata* meta, const uint32_t* cell_start, const uint32_t* indices) {
    if (int r = use(c)) return r;
    if (!meta || !cell_start) return fail(c, RT_ERR_INVALID, "rt_set_grid_csr: NULL argument");
    const uint32_t* g = meta->grid_sizes[RT_GRID_LEVELS - 1];
    size_t n_cells = (size_t)g[0] * g[1] * g[2];
    if (int r = check_meta(c, meta, n_cells, "rt_set_grid_csr")) return r;
    if (cell_start[n_cells] && !indices) return fail(c, RT_ERR_INVALID, "rt_set_grid_csr: indices is NULL");
    return install_csr(c, meta, cell_start, indices, n_cells);
}

// Dense per-cell records -> compact list. Only the count word and the first `count` ids of each
// record are read, so the untouched tail pages of a multi-gigabyte dense array are never faulted in.
template <class Cell, uint32_t CAP>
static int set_grid_dense(rt_ctx* c, const GridMetadata* meta, const Cell* cells, size_t n_cells, const char* who) {
    if (int r = use(c)) return r;
    if (!cells) return fail(c, RT_ERR_INVALID, "%s: fine_cells is NULL", who);
    if (int r = check_meta(c, meta, n_cells, who)) return r;
    std::vector<uint32_t> start(n_cells + 1);
    uint64_t acc = 0;
    for (size_t k = 0; k < n_cells; ++k) {
        start[k] = (uint32_t)acc;
        acc += cells[k].count < CAP ? cells[k].count : CAP;  // the shader stops at min(count, cap)
        if (acc > 0xFFFFFFF0ull) return fail(c, RT_ERR_LIMIT, "%s: more than 2^32-16 references", who);
    }
    start[n_cells] = (uint32_t)acc;
    std::vector<uint32_t> idx((size_t)acc);
    for (size_t k = 0; k < n_cells; ++k)
        memcpy(idx.data() + start[k], cells[k].object_indices, (size_t)(start[k + 1] - start[k]) * sizeof(uint32_t));
    return install_csr(c, meta, start.data(), idx.data(), n_cells);
}

extern "C" int rt_set_grid_dense(rt_ctx* c, const GridMetadata* meta, const FineCellData* fine_cells, size_t n_cells) {
    return set_grid_dense<FineCellData, RT_MAX_OBJECTS_PER_CELL>(c, meta, fine_cells, n_cells, "rt_set_grid_dense");
}

extern "C" int rt_set_grid_dense_tri(rt_ctx* c, const GridMetadata* meta, const TriFineCellData* fine_cells, size_t n_cells) {
    return set_grid_dense<TriFineCellData, RT_TRI_MAX_TRIANGLES_PER_CELL>(c, meta, fine_cells, n_cells,
                                                                         "rt_set_grid_dense_tri");
}

extern "C" int rt_build_grid(rt_ctx* c, float cell_size, uint32_t max_per_cell, int mode) {
    if (int r = use(c)) return r;
    if (mode != RT_GRID_HIERARCHICAL && mode != RT_GRID_TRIANGLES) return fail(c, RT_ERR_INVALID, "rt_build_grid: bad mode");
    if (!(cell_size > 0.0f)) cell_size = c->params.finest_cell_size;
    if (max_per_cell == 0)
        max_per_cell = mode == RT_GRID_TRIANGLES ? RT_TRI_MAX_TRIANGLES_PER_CELL : c->params.max_objects_per_cell;
    if (int r = scene_barrier(c)) return r;
    c->has_grid = false;
    if (int r = build_grid_device(c, cell_size, max_per_cell, mode)) return r;
    c->has_grid = true;
    return RT_OK;
}

extern "C" int rt_get_grid(rt_ctx* c, GridMetadata* meta, uint32_t* n_cells, uint32_t* n_refs) {
    if (int r = use(c)) return r;
    if (!c->has_grid) return fail(c, RT_ERR_STATE, "rt_get_grid: no grid");
    if (meta) *meta = c->meta;
    if (n_cells) *n_cells = c->n_cells;
    if (n_refs) *n_refs = c->n_refs;
    return RT_OK;
}

// Of the frame blocks (8x8 pixels) of the most recent rt_trace_primary launch on this context, how many its block mask
// kept; both zero when that launch had no mask. Waits for the stream.
extern "C" int rt_get_cull_stats(rt_ctx* c, uint32_t* blocks_total, uint32_t* blocks_kept) {
    if (int r = use(c)) return r;
    if (blocks_total) *blocks_total = 0;
    if (blocks_kept) *blocks_kept = 0;
    const rt_cull* k = c->last_cull;
    if (!k) return RT_OK;
    std::vector<uint32_t> host(k->words);
    RT_HIP(c, hipMemcpyAsync(host.data(), k->buf[k->cur].p, (size_t)k->words * sizeof(uint32_t), hipMemcpyDeviceToHost, k->stream));
    RT_HIP(c, hipStreamSynchronize(k->stream));
    stream_drained(c, k->stream);
    uint32_t kept = 0;
    for (uint32_t w : host) kept += (uint32_t)__builtin_popcount(w);
    if (blocks_total) *blocks_total = k->blocks;
    if (blocks_kept) *blocks_kept = kept;
    return RT_OK;
}

extern "C" int rt_checked_violations(rt_ctx* c, uint32_t* count, uint32_t first3[3]) {
    if (int r = use(c)) return r;
    if (!count) return fail(c, RT_ERR_INVALID, "rt_checked_violations: count is NULL");
#ifdef RT_CHECKED
    uint32_t v[4] = {0, 0, 0, 0};
    RT_HIP(c, hipDeviceSynchronize());
    RT_HIP(c, hipMemcpyFromSymbol(v, HIP_SYMBOL(rt::g_check), sizeof(v)));
    *count = v[0];
    if (first3) first3[0] = v[1], first3[1] = v[2], first3[2] = v[3];
    return RT_OK;
#else
    *count = 0;
    return fail(c, RT_ERR_STATE, "rt_checked_violations: this library was built without index checks (make checked)");
#endif
}

extern "C" int rt_get_heavy_blocks(rt_ctx* c, uint32_t* n_heavy, uint32_t* n_early, uint32_t* threshold_us) {
    if (int r = use(c)) return r;
    if (!n_heavy) return fail(c, RT_ERR_INVALID, "rt_get_heavy_blocks: NULL argument");
    *n_heavy = 0;
    if (n_early) *n_early = 0;
    if (threshold_us) *threshold_us = c->params.heavy_block_threshold;
    if (!c->last_plan) return RT_OK;
    const rt_plan* p = c->last_plan;
    uint32_t head[64] = {};  // the plan the NEXT launch of that shape will use: its two list lengths, its threshold
    RT_HIP(c, hipMemcpyAsync(head, p->buf[p->cur].p, sizeof(head), hipMemcpyDeviceToHost, p->stream));
    RT_HIP(c, hipStreamSynchronize(p->stream));
    *n_heavy = head[0] < PLAN_CAP_HEAVY ? head[0] : PLAN_CAP_HEAVY;
    if (n_early) *n_early = head[32] < PLAN_CAP_EARLY ? head[32] : PLAN_CAP_EARLY;
    if (threshold_us) *threshold_us = (head[1] > c->params.heavy_block_threshold * 100u ? head[1] : c->params.heavy_block_threshold * 100u) / 100u;
    return RT_OK;
}

extern "C" int rt_get_build_path(rt_ctx* c, int* path) {
    if (int r = use(c)) return r;
    if (!c->has_grid) return fail(c, RT_ERR_STATE, "rt_get_build_path: no grid");
    if (!path) return fail(c, RT_ERR_INVALID, "rt_get_build_path: NULL argument");
    *path = c->grid_built ? (c->last_build_fast ? 1 : 0) : -1;
    return RT_OK;
}

extern "C" int rt_get_trace_refs(rt_ctx* c, uint32_t* n_trace_refs) {
    if (int r = use(c)) return r;
    if (!c->has_grid) return fail(c, RT_ERR_STATE, "rt_get_trace_refs: no grid");
    if (!n_trace_refs) return fail(c, RT_ERR_INVALID, "rt_get_trace_refs: NULL argument");
    *n_trace_refs = c->n_refs_t;
    return RT_OK;
}

extern "C" int rt_read_grid_csr(rt_ctx* c, uint32_t* cell_start, uint32_t* indices) {
    if (int r = use(c)) return r;
    if (!c->has_grid) return fail(c, RT_ERR_STATE, "rt_read_grid_csr: no grid");
    if (cell_start)
        RT_HIP(c, hipMemcpyAsync(cell_start, c->cell_start.p, ((size_t)c->n_cells + 1) * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost, c->stream));
    if (indices && c->n_refs)
        RT_HIP(c, hipMemcpyAsync(indices, c->indices.p, (size_t)c->n_refs * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                 c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_update_vertices(rt_ctx* c, const float* xyz9, uint32_t first, uint32_t count) {
    if (int r = use(c)) return r;
    if (!xyz9 && count) return fail(c, RT_ERR_INVALID, "rt_update_vertices: xyz9 is NULL");
    if ((uint64_t)first + count > c->nt) return fail(c, RT_ERR_INVALID, "rt_update_vertices: range beyond %u triangles", c->nt);
    if (!count) return RT_OK;
    if (int r = scene_barrier(c)) return r;
    const bool dev = is_device_ptr(xyz9);
    RT_HIP(c, hipMemcpyAsync(c->tri_v.as<float>() + 9 * (size_t)first, xyz9, (size_t)count * 9 * sizeof(float),
                             dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_tri_normals, dim3((count + 255) / 256), dim3(256), 0, c->stream, c->tri_v.as<float>(), first,
                       count, c->tri_n.as<float4>());
    RT_HIP(c, hipGetLastError());
    if (!dev) RT_HIP(c, hipStreamSynchronize(c->stream));
    c->has_grid = false;  // cell lists and reference records are stale now
    c->has_bvh = false;
    return RT_OK;
}

// ================================================================== tracing

// NOALLOC: the form that allocates nothing (rt_trace_primary_noalloc): device outputs only, no counters, owner tables
// only if already resident, per-frame box records in the static scratch array of this device.
template <bool NOALLOC>
static int trace_primary(rt_ctx* c, const CameraUniform* cam, uint32_t width, uint32_t height, const rt_shard* shard,
                         rt_hit* out_primary, rt_hit* out_shadow, const char* who) {
    if (int r = use(c)) return r;
    if (!c->has_grid) return fail(c, RT_ERR_STATE, "%s: no grid (call rt_build_grid or rt_set_grid_*)", who);
    if (!cam || !out_primary) return fail(c, RT_ERR_INVALID, "%s: NULL camera or output", who);
    if (!width || !height) return fail(c, RT_ERR_INVALID, "%s: empty frame", who);
    float4* static_boxes = nullptr;
    if (NOALLOC) {
        if (c->params.collect_stats) return fail(c, RT_ERR_STATE, "%s: counters are not available here (collect_stats is on)", who);
        if (c->grid_nb > NOALLOC_MAX_BOXES)
            return fail(c, RT_ERR_LIMIT, "%s: the scene has %u boxes, the static scratch holds the per-frame records of %u", who,
                        c->grid_nb, NOALLOC_MAX_BOXES);
        if (c->device >= NOALLOC_MAX_DEVICES || !(static_boxes = g_scratch_ptr[c->device]))
            return fail(c, RT_ERR_LIMIT, "%s: no static scratch on device %d (devices 0 to %d have one)", who, c->device,
                        NOALLOC_MAX_DEVICES - 1);
    }
    ShardMap m;
    if (int r = make_shard(c, width, height, shard, &m, who, !NOALLOC)) return r;
    const size_t n_out = (size_t)m.n_local * m.tile_w * m.tile_h;
    if (n_out == 0) return RT_OK;
    const uint64_t blocks = (uint64_t)m.n_local * ((m.sub_x + 1) / 2) * ((m.sub_y + 1) / 2);  // 16x16 pixels each
    if (blocks > 0x7FFFFFFFull) return fail(c, RT_ERR_LIMIT, "%s: frame too large", who);

    const bool host_p = !is_device_ptr(out_primary);
    const bool host_s = out_shadow && !is_device_ptr(out_shadow);
    if (NOALLOC && (host_p || host_s)) return fail(c, RT_ERR_INVALID, "%s: outputs must be device memory", who);
    uint2* op = reinterpret_cast<uint2*>(out_primary);
    uint2* os = reinterpret_cast<uint2*>(out_shadow);
    if (host_p) {
        RT_HIP(c, c->out_p.reserve(n_out * sizeof(uint2)));
        op = c->out_p.as<uint2>();
    }
    if (host_s) {
        RT_HIP(c, c->out_s.reserve(n_out * sizeof(uint2)));
        os = c->out_s.as<uint2>();
    }
    std::unique_lock<std::mutex> scratch_hold(g_scratch_lock, std::defer_lock);
    if (NOALLOC) {
        if (c->grid_nb) {  // every record of boxes 0 .. nb-1 is rewritten: nothing in the array has to start at zero
            scratch_hold.lock();  // until this call's launches are queued and its event recorded (end of the function)
            ScratchUser& last = g_scratch_user[c->device];
            if (last.ctx && last.ctx != c) RT_HIP(c, hipStreamWaitEvent(c->stream, last.done, 0));
            const float t_lerp = (sinf(cam->time * 2.0f) + 1.0f) * 0.5f;  // raytracer_unified.wgsl:201,449
            hipLaunchKernelGGL(k_box_frame, dim3((c->nb + 255) / 256), dim3(256), 0, c->stream, c->boxes.as<BoxData>(), c->nb,
                               t_lerp, ld3(cam->position), cam->lod_factor, cam->min_pixel_size, c->params.lod_cull, static_boxes);
            RT_HIP(c, hipGetLastError());
            c->scratch_busy = true;  // a later call on another stream of this context is ordered behind this one
        }
    } else if (int r = update_box_frame(c, cam)) {
        return r;
    }
    const int stats = (int)c->params.collect_stats;
    if (stats)
        if (int r = zero_stats(c)) return r;

    DevScene s = make_scene(c);
    if (NOALLOC) s.box_frame = static_boxes;
    const DevFrame f = make_frame(c, cam, width, height);
    const dim3 grid((uint32_t)blocks);
    const bool wgsl = c->params.wgsl_compat != 0;
    int lr;
    if (out_shadow)
        lr = wgsl ? launch_primary<true, true>(c, stats, grid, s, f, m, op, os, !NOALLOC)
                  : launch_primary<false, true>(c, stats, grid, s, f, m, op, os, !NOALLOC);
    else
        lr = wgsl ? launch_primary<true, false>(c, stats, grid, s, f, m, op, os, !NOALLOC)
                  : launch_primary<false, false>(c, stats, grid, s, f, m, op, os, !NOALLOC);
    if (lr) return lr;
    RT_HIP(c, hipGetLastError());
    if (NOALLOC && c->grid_nb) {  // behind the launches that read the static array: what another context's call waits for
        RT_HIP(c, hipEventRecord(c->ev_scratch, c->stream));
        g_scratch_user[c->device] = ScratchUser{c, c->ev_scratch};
    }
    if (host_p) RT_HIP(c, hipMemcpyAsync(out_primary, op, n_out * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
    if (host_s) RT_HIP(c, hipMemcpyAsync(out_shadow, os, n_out * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
    if (stats) {
        if (int r = fetch_stats(c)) return r;
        stream_drained(c, c->stream);
    } else if (host_p || host_s) {
        RT_HIP(c, hipStreamSynchronize(c->stream));
        stream_drained(c, c->stream);
    }
    return RT_OK;
}

extern "C" int rt_trace_primary(rt_ctx* c, const CameraUniform* cam, uint32_t width, uint32_t height,
                                const rt_shard* shard, rt_hit* out_primary, rt_hit* out_shadow) {
    return trace_primary<false>(c, cam, width, height, shard, out_primary, out_shadow, "rt_trace_primary");
}

// Up to MAX_FRAMES_PER_LAUNCH frames of a camera path in one launch. Frame j's records start frame_stride records after
// frame j-1's (0 = packed). Triangle-only scenes without counters, textured alpha or a plan of expensive blocks take
// k_trace_primary_frames; anything else is traced frame by frame through rt_trace_primary, with the same records.
extern "C" int rt_trace_primary_frames(rt_ctx* c, const CameraUniform* cams, uint32_t n_frames, uint32_t width, uint32_t height,
                                       const rt_shard* shard, rt_hit* out_primary, rt_hit* out_shadow, size_t frame_stride) {
    const char* who = "rt_trace_primary_frames";
    if (int r = use(c)) return r;
    if (!c->has_grid) return fail(c, RT_ERR_STATE, "%s: no grid (call rt_build_grid or rt_set_grid_*)", who);
    if (!cams || !out_primary) return fail(c, RT_ERR_INVALID, "%s: NULL camera or output", who);
    if (!width || !height) return fail(c, RT_ERR_INVALID, "%s: empty frame", who);
    if (n_frames == 0) return RT_OK;
    if (n_frames > MAX_FRAMES_PER_LAUNCH)
        return fail(c, RT_ERR_LIMIT, "%s: %u frames in one call (max %u)", who, n_frames, MAX_FRAMES_PER_LAUNCH);
    ShardMap m;
    if (int r = make_shard(c, width, height, shard, &m, who)) return r;
    const size_t n_frame = (size_t)m.n_local * m.tile_w * m.tile_h;
    if (n_frame == 0) return RT_OK;
    if (frame_stride && frame_stride < n_frame)
        return fail(c, RT_ERR_INVALID, "%s: frame_stride %zu smaller than the %zu records of a frame", who, frame_stride, n_frame);
    const size_t stride = frame_stride ? frame_stride : n_frame;
    const uint64_t blocks = (uint64_t)m.n_local * ((m.sub_x + 1) / 2) * ((m.sub_y + 1) / 2);
    if (blocks > 0x7FFFFFFFull) return fail(c, RT_ERR_LIMIT, "%s: frame too large", who);
    const DevScene s = make_scene(c);
    const bool host_out = !is_device_ptr(out_primary) || (out_shadow && !is_device_ptr(out_shadow));
    const bool grouped = !c->params.collect_stats && c->grid_nb == 0 && c->tex_mode != 2 && !wide_grid(s) &&
                         !c->params.heavy_block_threshold && !c->params.repack_candidates && !c->params.persistent_waves && !host_out && n_frames > 1;
    if (!grouped) {
        for (uint32_t j = 0; j < n_frames; ++j)
            if (int r = trace_primary<false>(c, cams + j, width, height, shard, out_primary + j * stride,
                                             out_shadow ? out_shadow + j * stride : nullptr, who))
                return r;
        return RT_OK;
    }
    DevFrames fs;
    memset(&fs, 0, sizeof(fs));
    for (uint32_t j = 0; j < n_frames; ++j) fs.f[j] = make_frame(c, cams + j, width, height);
    const uint32_t* mask = nullptr;
    uint32_t words = 0;
    c->last_cull = nullptr;
    if (int r = cull_masks_for(c, s, fs.f, n_frames, m, &mask, &words, false)) return r;
    uint32_t* zero = nullptr;
    uint32_t zero_words = 0;
    if (mask) {
        rt_cull* k = c->last_cull;
        zero = k->buf[k->cur ^ 1].as<uint32_t>();
        zero_words = k->dirty[k->cur ^ 1];
        k->dirty[k->cur ^ 1] = 0;
    }
    uint2* op = reinterpret_cast<uint2*>(out_primary);
    uint2* os = reinterpret_cast<uint2*>(out_shadow);
    const dim3 grid((uint32_t)blocks, n_frames);
    const unsigned long long fstride = stride;
    const bool wgsl = c->params.wgsl_compat != 0;
    if (out_shadow) {
        if (wgsl) hipLaunchKernelGGL((k_trace_primary_frames<true, true>), grid, dim3(256), 0, c->stream, s, fs, m, op, os, mask, zero, zero_words, words, fstride);
        else hipLaunchKernelGGL((k_trace_primary_frames<false, true>), grid, dim3(256), 0, c->stream, s, fs, m, op, os, mask, zero, zero_words, words, fstride);
    } else {
        if (wgsl) hipLaunchKernelGGL((k_trace_primary_frames<true, false>), grid, dim3(256), 0, c->stream, s, fs, m, op, os, mask, zero, zero_words, words, fstride);
        else hipLaunchKernelGGL((k_trace_primary_frames<false, false>), grid, dim3(256), 0, c->stream, s, fs, m, op, os, mask, zero, zero_words, words, fstride);
    }
    RT_HIP(c, hipGetLastError());
    if (c->params.cull_empty_blocks == 1)
        if (int r = mark_in_flight(c)) return r;
    return RT_OK;
}

// Calls no hipMalloc / hipFree and no routine that allocates: see the scratch array at the top of this file.
extern "C" int rt_trace_primary_noalloc(rt_ctx* c, const CameraUniform* cam, uint32_t width, uint32_t height,
                                        const rt_shard* shard, rt_hit* out_primary, rt_hit* out_shadow) {
    return trace_primary<true>(c, cam, width, height, shard, out_primary, out_shadow, "rt_trace_primary_noalloc");
}

extern "C" int rt_trace_rays(rt_ctx* c, const CameraUniform* cam, const float* origins, const float* dirs, uint32_t n,
                             rt_hit* out, int any_hit) {
    if (int r = use(c)) return r;
    if (!c->has_grid) return fail(c, RT_ERR_STATE, "rt_trace_rays: no grid (call rt_build_grid or rt_set_grid_*)");
    if (n == 0) return RT_OK;
    if (!origins || !dirs || !out) return fail(c, RT_ERR_INVALID, "rt_trace_rays: NULL argument");
    const void *d_o, *d_d;
    if (int r = stage_in(c, c->in_o, origins, (size_t)n * 3 * sizeof(float), &d_o)) return r;
    if (int r = stage_in(c, c->in_d, dirs, (size_t)n * 3 * sizeof(float), &d_d)) return r;
    const bool host_out = !is_device_ptr(out);
    uint2* d_out = reinterpret_cast<uint2*>(out);
    if (host_out) {
        RT_HIP(c, c->out_p.reserve((size_t)n * sizeof(uint2)));
        d_out = c->out_p.as<uint2>();
    }
    if (int r = update_box_frame(c, cam)) return r;
    const int stats = (int)c->params.collect_stats;
    if (stats)
        if (int r = zero_stats(c)) return r;
    const DevScene s = make_scene(c);
    const dim3 grid((n + 255) / 256);
    const bool wgsl = c->params.wgsl_compat != 0;
    const float* fo = static_cast<const float*>(d_o);
    const float* fd = static_cast<const float*>(d_d);
    if (any_hit) {
        if (wgsl)
            launch_rays<true, true>(c, stats, grid, s, fo, fd, n, d_out);
        else
            launch_rays<false, true>(c, stats, grid, s, fo, fd, n, d_out);
    } else {
        if (wgsl)
            launch_rays<true, false>(c, stats, grid, s, fo, fd, n, d_out);
        else
            launch_rays<false, false>(c, stats, grid, s, fo, fd, n, d_out);
    }
    RT_HIP(c, hipGetLastError());
    if (host_out) RT_HIP(c, hipMemcpyAsync(out, d_out, (size_t)n * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
    if (stats) {
        if (int r = fetch_stats(c)) return r;
    } else {
        // staged inputs must stay untouched until the kernel has read them
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}

// PIXEL: the output record (uint2 hit record, uint32_t RGBA8 pixel); RGB8: the gathered pixels are 3 bytes.
// owner: the rt_shard.owner table the ranks rendered with (`round` entries), NULL = one tile each per round of n_ranks.
template <class PIXEL, bool RGB8>
static int assemble_tiles(rt_ctx* c, const void* gathered, uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h,
                          uint32_t n_ranks, const uint32_t* owner, uint32_t round, size_t rank_stride, void* image, const char* who,
                          uint32_t n_frames = 1, size_t frame_stride = 0, const CameraUniform* sky_cams = nullptr) {
    if (int r = use(c)) return r;
    SkyFrames sky;
    memset(&sky, 0, sizeof(sky));
    if (sky_cams) {  // owner entries of RT_OWNER_SKY are filled from these cameras
        if (sizeof(PIXEL) != 4 || !owner) return fail(c, RT_ERR_INVALID, "%s: sky tiles need an owner table and 4-byte pixels", who);
        if (n_frames > SKY_MAX_FRAMES) return fail(c, RT_ERR_LIMIT, "%s: %u frames with sky tiles (max %u)", who, n_frames, SKY_MAX_FRAMES);
        for (uint32_t j = 0; j < n_frames; ++j)
            for (int k = 0; k < 3; ++k)
                sky.basis[j][k] = sky_cams[j].forward[k], sky.basis[j][3 + k] = sky_cams[j].right[k], sky.basis[j][6 + k] = sky_cams[j].up[k];
        sky.fov_scale = tanf(c->params.fov);  // as make_frame evaluates it
        sky.enabled = 1;
    }
    if (!gathered || !image || !width || !height || !tile_w || !tile_h || !n_ranks)
        return fail(c, RT_ERR_INVALID, "%s: NULL or zero argument", who);
    if (!is_device_ptr(gathered) || !is_device_ptr(image))
        return fail(c, RT_ERR_INVALID, "%s: both buffers must be device memory", who);
    const uint32_t tiles_x = (width + tile_w - 1) / tile_w, tiles_y = (height + tile_h - 1) / tile_h;
    const uint32_t n_tiles = tiles_x * tiles_y;
    AssemblePlan plan;
    memset(&plan, 0, sizeof(plan));
    plan.n_ranks = plan.round = n_ranks;
    uint32_t most = (n_tiles + n_ranks - 1) / n_ranks;  // tiles of the rank that owns most
    if (owner) {
        if (!round || round > MAX_OWNER_TABLE)
            return fail(c, RT_ERR_LIMIT, "%s: owner table of %u entries (1 to %u)", who, round, MAX_OWNER_TABLE);
        std::vector<uint32_t> key{1u, n_ranks, round}, places(3 * (size_t)round), count(n_ranks, 0u), tiles(n_ranks, 0u);
        key.insert(key.end(), owner, owner + round);
        const uint32_t rem = n_tiles % round;
        for (uint32_t o = 0; o < round; ++o) {
            if (sky.enabled && owner[o] == OWNER_SKY) {
                places[3 * o] = OWNER_SKY;
                continue;
            }
            if (owner[o] >= n_ranks) return fail(c, RT_ERR_INVALID, "%s: owner[%u] = %u, but there are %u ranks", who, o, owner[o], n_ranks);
            places[3 * o] = owner[o];
            places[3 * o + 1] = count[owner[o]]++;
            tiles[owner[o]] += o < rem ? 1u : 0u;
        }
        most = 0;
        for (uint32_t o = 0; o < round; ++o)
            if (places[3 * o] != OWNER_SKY) places[3 * o + 2] = count[owner[o]];
        for (uint32_t r = 0; r < n_ranks; ++r) most = std::max(most, n_tiles / round * count[r] + tiles[r]);
        plan.round = round;
        if (int r = table_for(c, std::move(key), places, &plan.places)) return r;
    }
    const size_t min_stride = (size_t)most * tile_w * tile_h;
    const size_t per_rank = rank_stride ? rank_stride : min_stride;
    if (n_frames == 0) return RT_OK;
    if (n_frames > 65535u) return fail(c, RT_ERR_LIMIT, "%s: more than 65535 frames", who);
    if (n_frames > 1 && !rank_stride) return fail(c, RT_ERR_INVALID, "%s: rank_stride is required", who);
    // the last frame of a rank's block must end inside the block: the kernels do no bounds checking
    if (per_rank < min_stride ||
        (n_frames > 1 && (frame_stride < min_stride || (size_t)(n_frames - 1) * frame_stride + min_stride > per_rank)))
        return fail(c, RT_ERR_INVALID, "%s: frame_stride / rank_stride smaller than the blocks they hold", who);
    constexpr uint32_t PER = 16 / sizeof(PIXEL);
    const bool by_vectors = tile_w % PER == 0 && width % PER == 0 && per_rank % PER == 0 &&
                            (n_frames == 1 || frame_stride % PER == 0) && reinterpret_cast<uintptr_t>(gathered) % 16 == 0 &&
                            reinterpret_cast<uintptr_t>(image) % 16 == 0 && tiles_y <= 65535u &&
                            (uint64_t)tile_w * tile_h / PER <= 0xFFFFFFFFull;
    if (by_vectors)
        hipLaunchKernelGGL((k_assemble_x16<PER, RGB8>), dim3(tiles_x, tiles_y, n_frames), dim3(256), 0, c->stream,
                           reinterpret_cast<const uint32_t*>(gathered), width, height, tile_w, tile_h, tiles_x, plan, per_rank,
                           frame_stride, reinterpret_cast<uint4*>(image), sky);
    else
        hipLaunchKernelGGL((k_assemble<PIXEL, RGB8>), dim3((width + 255) / 256, height, n_frames), dim3(256), 0, c->stream, gathered,
                           width, height, tile_w, tile_h, tiles_x, plan, per_rank, frame_stride, reinterpret_cast<PIXEL*>(image), sky);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

extern "C" int rt_assemble_tiles(rt_ctx* c, const rt_hit* gathered, uint32_t width, uint32_t height, uint32_t tile_w,
                                 uint32_t tile_h, uint32_t n_ranks, size_t rank_stride, rt_hit* image) {
    return assemble_tiles<uint2, false>(c, gathered, width, height, tile_w, tile_h, n_ranks, nullptr, 0, rank_stride, image,
                                        "rt_assemble_tiles");
}

extern "C" int rt_assemble_tiles_rgba8(rt_ctx* c, const uint8_t* gathered, uint32_t width, uint32_t height, uint32_t tile_w,
                                       uint32_t tile_h, uint32_t n_ranks, size_t rank_stride, uint8_t* image) {
    return assemble_tiles<uint32_t, false>(c, gathered, width, height, tile_w, tile_h, n_ranks, nullptr, 0, rank_stride, image,
                                           "rt_assemble_tiles_rgba8");
}

extern "C" int rt_assemble_frames(rt_ctx* c, const uint8_t* gathered, uint32_t pixel_format, uint32_t width, uint32_t height,
                                  uint32_t tile_w, uint32_t tile_h, uint32_t n_ranks, const uint32_t* owner, uint32_t round,
                                  uint32_t n_frames, size_t rank_stride, size_t frame_stride, uint8_t* images) {
    if (pixel_format == RT_PIXEL_RGB8)
        return assemble_tiles<uint32_t, true>(c, gathered, width, height, tile_w, tile_h, n_ranks, owner, round, rank_stride, images,
                                              "rt_assemble_frames", n_frames, frame_stride);
    if (pixel_format == RT_PIXEL_RGBA8)
        return assemble_tiles<uint32_t, false>(c, gathered, width, height, tile_w, tile_h, n_ranks, owner, round, rank_stride, images,
                                               "rt_assemble_frames", n_frames, frame_stride);
    if (int r = use(c)) return r;
    return fail(c, RT_ERR_INVALID, "rt_assemble_frames: unknown pixel_format %u", pixel_format);
}

extern "C" int rt_assemble_frames_sky(rt_ctx* c, const uint8_t* gathered, uint32_t pixel_format, uint32_t width, uint32_t height,
                                      uint32_t tile_w, uint32_t tile_h, uint32_t n_ranks, const uint32_t* owner, uint32_t round,
                                      const CameraUniform* cameras, uint32_t n_frames, size_t rank_stride, size_t frame_stride,
                                      uint8_t* images) {
    if (!cameras) {
        if (int r = use(c)) return r;
        return fail(c, RT_ERR_INVALID, "rt_assemble_frames_sky: cameras is NULL");
    }
    if (pixel_format == RT_PIXEL_RGB8)
        return assemble_tiles<uint32_t, true>(c, gathered, width, height, tile_w, tile_h, n_ranks, owner, round, rank_stride, images,
                                              "rt_assemble_frames_sky", n_frames, frame_stride, cameras);
    if (pixel_format == RT_PIXEL_RGBA8)
        return assemble_tiles<uint32_t, false>(c, gathered, width, height, tile_w, tile_h, n_ranks, owner, round, rank_stride, images,
                                               "rt_assemble_frames_sky", n_frames, frame_stride, cameras);
    if (int r = use(c)) return r;
    return fail(c, RT_ERR_INVALID, "rt_assemble_frames_sky: unknown pixel_format %u", pixel_format);
}

// Which tiles of the frame can show geometry from this camera: the frame's block mask, made now, read back, reduced.
extern "C" int rt_tile_flags(rt_ctx* c, const CameraUniform* cam, uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h,
                             uint8_t* flags) {
    if (int r = use(c)) return r;
    if (!c->has_grid) return fail(c, RT_ERR_STATE, "rt_tile_flags: no grid (call rt_build_grid or rt_set_grid_*)");
    if (!cam || !flags || !width || !height || !tile_w || !tile_h) return fail(c, RT_ERR_INVALID, "rt_tile_flags: NULL or zero argument");
    const uint32_t tiles_x = (width + tile_w - 1) / tile_w, tiles_y = (height + tile_h - 1) / tile_h;
    memset(flags, 1, (size_t)tiles_x * tiles_y);
    if ((tile_w & 7u) || (tile_h & 7u) || !c->params.cull_empty_blocks) return RT_OK;  // no mask wanted or possible: every tile is rendered
    ShardMap m;
    if (int r = make_shard(c, width, height, nullptr, &m, "rt_tile_flags")) return r;
    const DevScene s = make_scene(c);
    const DevFrame f = make_frame(c, cam, width, height);
    const uint32_t* mask = nullptr;
    c->last_cull = nullptr;
    if (int r = cull_mask_for(c, s, f, m, &mask, true)) return r;
    if (!mask) return RT_OK;
    rt_cull* k = c->last_cull;
    std::vector<uint32_t> host(k->words);
    RT_HIP(c, hipMemcpyAsync(host.data(), mask, (size_t)k->words * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    const uint32_t blocks_x = (width + 7u) >> 3, blocks_y = (height + 7u) >> 3, words_per_row = (blocks_x + 31u) >> 5;
    memset(flags, 0, (size_t)tiles_x * tiles_y);
    for (uint32_t by = 0; by < blocks_y; ++by)
        for (uint32_t bx = 0; bx < blocks_x; ++bx)
            if ((host[(size_t)by * words_per_row + (bx >> 5)] >> (bx & 31u)) & 1u) flags[(size_t)(by * 8u / tile_h) * tiles_x + bx * 8u / tile_w] = 1;
    return RT_OK;
}

extern "C" int rt_assemble_frames_rgba8(rt_ctx* c, const uint8_t* gathered, uint32_t width, uint32_t height, uint32_t tile_w,
                                        uint32_t tile_h, uint32_t n_ranks, uint32_t n_frames, size_t rank_stride,
                                        size_t frame_stride, uint8_t* images) {
    return assemble_tiles<uint32_t, false>(c, gathered, width, height, tile_w, tile_h, n_ranks, nullptr, 0, rank_stride, images,
                                           "rt_assemble_frames_rgba8", n_frames, frame_stride);
}

extern "C" int rt_assemble_frames_rgb8(rt_ctx* c, const uint8_t* gathered, uint32_t width, uint32_t height, uint32_t tile_w,
                                       uint32_t tile_h, uint32_t n_ranks, uint32_t n_frames, size_t rank_stride,
                                       size_t frame_stride, uint8_t* images) {
    return assemble_tiles<uint32_t, true>(c, gathered, width, height, tile_w, tile_h, n_ranks, nullptr, 0, rank_stride, images,
                                          "rt_assemble_frames_rgb8", n_frames, frame_stride);
}

// ================================================================== skinning

extern "C" int rt_set_skin(rt_ctx* c, const float* positions, const uint16_t* joints, const float* weights, uint32_t n_vertices,
                           const uint32_t* tri_indices, uint32_t n_triangles) {
    if (int r = use(c)) return r;
    if (!positions || !joints || !weights || !tri_indices || !n_vertices)
        return fail(c, RT_ERR_INVALID, "rt_set_skin: NULL or empty argument");
    if (n_triangles != c->nt)
        return fail(c, RT_ERR_INVALID, "rt_set_skin: %u triangles, the scene has %u (call rt_set_scene first)", n_triangles, c->nt);
    for (size_t k = 0; k < (size_t)n_triangles * 3; ++k)
        if (tri_indices[k] >= n_vertices) return fail(c, RT_ERR_INVALID, "rt_set_skin: vertex index %u of %u at %zu", tri_indices[k], n_vertices, k);
    uint32_t max_joint = 0;
    for (size_t k = 0; k < (size_t)n_vertices * 4; ++k) max_joint = std::max<uint32_t>(max_joint, joints[k]);
    if (int r = scene_barrier(c)) return r;
    RT_HIP(c, c->skin_pos.reserve((size_t)n_vertices * 12));
    RT_HIP(c, c->skin_joints.reserve((size_t)n_vertices * 8));
    RT_HIP(c, c->skin_weights.reserve((size_t)n_vertices * 16));
    RT_HIP(c, c->skin_tri.reserve((size_t)n_triangles * 12 + 16));
    RT_HIP(c, hipMemcpyAsync(c->skin_pos.p, positions, (size_t)n_vertices * 12, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync(c->skin_joints.p, joints, (size_t)n_vertices * 8, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync(c->skin_weights.p, weights, (size_t)n_vertices * 16, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipMemcpyAsync(c->skin_tri.p, tri_indices, (size_t)n_triangles * 12, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->skin_nv = n_vertices;
    c->skin_max_joint = max_joint;
    c->has_skin = true;
    return RT_OK;
}

extern "C" int rt_skin_vertices(rt_ctx* c, const float* joint_matrices, uint32_t n_joints) {
    if (int r = use(c)) return r;
    if (!c->has_skin) return fail(c, RT_ERR_STATE, "rt_skin_vertices: no skin (call rt_set_skin)");
    if (!joint_matrices) return fail(c, RT_ERR_INVALID, "rt_skin_vertices: NULL matrices");
    if (n_joints <= c->skin_max_joint)
        return fail(c, RT_ERR_INVALID, "rt_skin_vertices: %u matrices, the skin uses joint %u", n_joints, c->skin_max_joint);
    if (!c->nt) return RT_OK;
    if (int r = scene_barrier(c)) return r;
    const float4* mats = reinterpret_cast<const float4*>(joint_matrices);
    const bool dev = is_device_ptr(joint_matrices);
    if (!dev) {
        RT_HIP(c, c->skin_mats.reserve((size_t)n_joints * 64));
        RT_HIP(c, hipMemcpyAsync(c->skin_mats.p, joint_matrices, (size_t)n_joints * 64, hipMemcpyHostToDevice, c->stream));
        mats = c->skin_mats.as<float4>();
    }
    hipLaunchKernelGGL(k_skin_triangles, dim3((c->nt + 255) / 256), dim3(256), 0, c->stream, c->skin_pos.as<float>(),
                       c->skin_joints.as<ushort4>(), c->skin_weights.as<float4>(), c->skin_tri.as<uint32_t>(), c->nt, mats,
                       c->tri_v.as<float>(), c->tri_n.as<float4>());
    RT_HIP(c, hipGetLastError());
    if (!dev) RT_HIP(c, hipStreamSynchronize(c->stream));  // the caller may reuse its matrix array
    c->has_grid = false;  // cell lists and reference records are stale now
    c->has_bvh = false;
    return RT_OK;
}

extern "C" int rt_read_vertices(rt_ctx* c, float* xyz9) {
    if (int r = use(c)) return r;
    if (!xyz9) return fail(c, RT_ERR_INVALID, "rt_read_vertices: NULL output");
    if (c->nt) {
        RT_HIP(c, hipMemcpyAsync(xyz9, c->tri_v.p, (size_t)c->nt * 36, hipMemcpyDefault, c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}

// ================================================================== shaded picture

extern "C" int rt_set_materials(rt_ctx* c, const MaterialData* materials, uint32_t n) {
    if (int r = use(c)) return r;
    if (n && !materials) return fail(c, RT_ERR_INVALID, "rt_set_materials: NULL array with non-zero count");
    if (n > 0x7FFFFFFFu) return fail(c, RT_ERR_LIMIT, "rt_set_materials: more than 2^31-1 materials");
    if (int r = scene_barrier(c)) return r;
    RT_HIP(c, c->materials.reserve((size_t)n * sizeof(MaterialData) + 16));
    if (n) RT_HIP(c, hipMemcpyAsync(c->materials.p, materials, (size_t)n * sizeof(MaterialData), hipMemcpyHostToDevice, c->stream));
    c->n_materials = n;
    c->dup_state = -1;  // which copies may be left out depends on the originals' alpha rules
    c->tex_mode = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const MaterialData& m = materials[i];
        if (m.texture_index >= 0 && (m.alpha_mode == 1u || m.alpha_mode == 2u)) c->tex_mode = 2;
        if (c->tex_mode < 1 && (m.texture_index >= 0 || m.normal_texture_index >= 0 || m.emissive_texture_index >= 0))
            c->tex_mode = 1;
    }
    if (c->has_grid) {
        // which duplicates may be left out, and the id / material words of the records, depend on the materials
        if (int r = rebuild_trace_records(c)) return r;
    }
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_set_textures(rt_ctx* c, const rt_texture* textures, uint32_t n) {
    if (int r = use(c)) return r;
    if (n && !textures) return fail(c, RT_ERR_INVALID, "rt_set_textures: NULL array with non-zero count");
    uint32_t w = 0, h = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (!textures[i].rgba8 || !textures[i].width || !textures[i].height)
            return fail(c, RT_ERR_INVALID, "rt_set_textures: texture %u is empty", i);
        w = textures[i].width > w ? textures[i].width : w;
        h = textures[i].height > h ? textures[i].height : h;
    }
    if (n && ((uint64_t)w * h * n > 0x40000000ull || w > 16384u || h > 16384u))
        return fail(c, RT_ERR_LIMIT, "rt_set_textures: %u layers of %u x %u texels (max 16384 a side, 2^30 texels in all)", n, w, h);
    if (int r = scene_barrier(c)) return r;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    // the array as the renderer builds it: every layer max_w x max_h, zero, texture in its top-left corner
    const size_t layer = (size_t)w * h * 4;
    RT_HIP(c, c->texels.reserve(layer * n + 16));
    if (n) RT_HIP(c, hipMemsetAsync(c->texels.p, 0, layer * n, c->stream));
    for (uint32_t i = 0; i < n; ++i)
        RT_HIP(c, hipMemcpy2DAsync(static_cast<uint8_t*>(c->texels.p) + layer * i, (size_t)w * 4, textures[i].rgba8,
                                   (size_t)textures[i].width * 4, (size_t)textures[i].width * 4, textures[i].height,
                                   hipMemcpyHostToDevice, c->stream));
    c->tex_w = w;
    c->tex_h = h;
    c->tex_layers = n;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" void rt_default_render_opts(rt_render_opts* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->max_bounces = 8;  // MAX_BOUNCES, raytracer_unified.wgsl:656
}

extern "C" int rt_render_frames(rt_ctx* c, const CameraUniform* cams, uint32_t n_frames, uint32_t width, uint32_t height,
                                const rt_shard* shard, const rt_render_opts* opts, uint8_t* out_rgba8) {
    if (int r = use(c)) return r;
    if (!c->has_grid) return fail(c, RT_ERR_STATE, "rt_render: no grid (call rt_build_grid or rt_set_grid_*)");
    if (!cams || !out_rgba8) return fail(c, RT_ERR_INVALID, "rt_render: NULL camera or output");
    if (!width || !height) return fail(c, RT_ERR_INVALID, "rt_render: empty frame");
    if (n_frames == 0) return RT_OK;
    if (n_frames > MAX_FRAMES_PER_LAUNCH)
        return fail(c, RT_ERR_LIMIT, "rt_render_frames: %u frames in one call (max %u)", n_frames, MAX_FRAMES_PER_LAUNCH);
    rt_render_opts o;
    rt_default_render_opts(&o);
    if (opts) o = *opts;
    ShardMap m;
    if (int r = make_shard(c, width, height, shard, &m, "rt_render")) return r;
    const size_t n_frame = (size_t)m.n_local * m.tile_w * m.tile_h, n_out = n_frame * n_frames;
    if (n_out == 0) return RT_OK;
    const uint64_t blocks = (uint64_t)m.n_local * ((m.sub_x + 1) / 2) * ((m.sub_y + 1) / 2);
    if (blocks > 0x7FFFFFFFull) return fail(c, RT_ERR_LIMIT, "rt_render: frame too large");
    if (o.pixel_format != RT_PIXEL_RGBA8 && o.pixel_format != RT_PIXEL_RGB8)
        return fail(c, RT_ERR_INVALID, "rt_render: unknown pixel_format %u", o.pixel_format);
    if (o.frame_stride && o.frame_stride < n_frame)
        return fail(c, RT_ERR_INVALID, "rt_render_frames: frame_stride %u smaller than the %zu pixels of a frame", o.frame_stride, n_frame);
    const size_t stride = o.frame_stride ? o.frame_stride : n_frame;
    const size_t out_bytes = ((n_frames - 1) * stride + n_frame) * (o.pixel_format == RT_PIXEL_RGB8 ? 3 : 4);
    const bool host_out = !is_device_ptr(out_rgba8);
    uint32_t* out = reinterpret_cast<uint32_t*>(out_rgba8);
    if (host_out) {
        RT_HIP(c, c->out_p.reserve(out_bytes + 16));
        out = c->out_p.as<uint32_t>();
    }
    if (int r = update_box_frames(c, cams, n_frames)) return r;
    const bool stats = c->params.collect_stats != 0;
    if (stats)
        if (int r = zero_stats(c)) return r;
    const DevScene s = make_scene(c);
    DevFrames fs;
    memset(&fs, 0, sizeof(fs));
    for (uint32_t j = 0; j < n_frames; ++j) {
        fs.f[j] = make_frame(c, cams + j, width, height);
        fs.show_grid[j] = cams[j].show_grid;
    }
    DevShade sh;
    memset(&sh, 0, sizeof(sh));
    sh.boxes = c->boxes.as<float4>();
    sh.tri_mat = c->tri_mat.as<uint32_t>();
    sh.materials = c->materials.as<float4>();
    sh.n_materials = c->n_materials;
    sh.max_bounces = o.max_bounces;
    sh.shadows = o.shadows ? 1u : 0u;
    sh.debug_enabled = o.debug_enabled;
    sh.debug_x = o.debug_pixel[0];
    sh.debug_y = o.debug_pixel[1];
    sh.rgb8 = o.pixel_format == RT_PIXEL_RGB8 ? 1u : 0u;
    sh.box_frame_stride = c->nb * 4;
    sh.frame_pixels = stride;
    unsigned long long* st = stats ? c->stats.as<unsigned long long>() : nullptr;
    // the frames' masks of pixel blocks that can meet geometry: the others are sky (rt_params.cull_empty_blocks)
    const uint32_t* rmask = nullptr;
    uint32_t rwords = 0;
    c->last_cull = nullptr;
    if (!stats && !wide_grid(s))
        if (int r = cull_masks_for(c, s, fs.f, n_frames, m, &rmask, &rwords, false)) return r;
    const dim3 grid((uint32_t)blocks, n_frames);
    const bool wgsl = c->params.wgsl_compat != 0, has_boxes = s.nb != 0;
    // Measured on the bundled static model, 1080p, two frames in flight: without texture-decided alpha 120.4 / 134.3 /
    // 147.5 us for 2 / 4 / 8 bounces against k_render's 125.5 / 144.3 / 165.8; with it (128 registers and 40 bytes of
    // scratch per lane) 138.4 / 153.6 / 166.8 against 128.8 / 150.6 / 178.7: there only long paths gain.
    const bool deal = !stats && !has_boxes && !wide_grid(s) && o.max_bounces > (c->tex_mode == 2 ? 5u : 1u) &&
                      !o.plain_bounces;
    if (deal) {
        // reflection bounces of a triangle-only scene: the scattered rays of a wave share their candidate lists
        if (c->tex_mode == 2) {
            if (wgsl) hipLaunchKernelGGL((k_render_deal<true, true, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, rmask, rwords);
            else hipLaunchKernelGGL((k_render_deal<false, true, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, rmask, rwords);
        } else if (c->tex_mode != 0) {
            if (wgsl) hipLaunchKernelGGL((k_render_deal<true, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, rmask, rwords);
            else hipLaunchKernelGGL((k_render_deal<false, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, rmask, rwords);
        } else {
            if (wgsl) hipLaunchKernelGGL((k_render_deal<true, false>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, rmask, rwords);
            else hipLaunchKernelGGL((k_render_deal<false, false>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, rmask, rwords);
        }
    } else if (wide_grid(s)) {
        if (wgsl) hipLaunchKernelGGL((k_render<true, true, true, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
        else hipLaunchKernelGGL((k_render<false, true, true, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
    } else if (c->tex_mode != 0 && o.max_bounces == 1 && !has_boxes) {  // textured, direct lighting only: no bounce loop to carry state through
        if (wgsl) hipLaunchKernelGGL((k_render<true, false, true, false, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
        else hipLaunchKernelGGL((k_render<false, false, true, false, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
    } else if (c->tex_mode != 0) {  // materials name textures: the general kernel with the material path
        if (wgsl && has_boxes) hipLaunchKernelGGL((k_render<true, true, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
        else if (wgsl) hipLaunchKernelGGL((k_render<true, false, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
        else if (has_boxes) hipLaunchKernelGGL((k_render<false, true, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
        else hipLaunchKernelGGL((k_render<false, false, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
    } else if (o.max_bounces == 1) {  // direct lighting only
        if (wgsl && has_boxes) hipLaunchKernelGGL((k_render<true, true, false, false, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
        else if (wgsl) hipLaunchKernelGGL((k_render<true, false, false, false, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
        else if (has_boxes) hipLaunchKernelGGL((k_render<false, true, false, false, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
        else hipLaunchKernelGGL((k_render<false, false, false, false, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
    } else if (wgsl && has_boxes) hipLaunchKernelGGL((k_render<true, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
    else if (wgsl) hipLaunchKernelGGL((k_render<true, false>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
    else if (has_boxes) hipLaunchKernelGGL((k_render<false, true>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
    else hipLaunchKernelGGL((k_render<false, false>), grid, dim3(256), 0, c->stream, s, fs, m, sh, out, st, rmask, rwords);
    RT_HIP(c, hipGetLastError());
    if (c->params.cull_empty_blocks == 1 && s.nb == 0)
        if (int r = mark_in_flight(c)) return r;
    if (host_out) RT_HIP(c, hipMemcpyAsync(out_rgba8, out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    if (stats) {
        if (int r = fetch_stats(c)) return r;
        stream_drained(c, c->stream);
    } else if (host_out) {
        RT_HIP(c, hipStreamSynchronize(c->stream));
        stream_drained(c, c->stream);
    }
    return RT_OK;
}

extern "C" int rt_render(rt_ctx* c, const CameraUniform* cam, uint32_t width, uint32_t height, const rt_shard* shard,
                         const rt_render_opts* opts, uint8_t* out_rgba8) {
    return rt_render_frames(c, cam, 1, width, height, shard, opts, out_rgba8);
}

extern "C" int rt_debug_pixel(rt_ctx* c, const CameraUniform* cam, uint32_t width, uint32_t height, uint32_t px,
                              uint32_t py, RayDebugInfo* info) {
    if (int r = use(c)) return r;
    if (!c->has_grid) return fail(c, RT_ERR_STATE, "rt_debug_pixel: no grid");
    if (!cam || !info || px >= width || py >= height) return fail(c, RT_ERR_INVALID, "rt_debug_pixel: bad argument");
    if (int r = update_box_frame(c, cam)) return r;
    RT_HIP(c, c->misc.reserve(sizeof(RayDebugInfo)));
    const DevScene s = make_scene(c);
    const DevFrame f = make_frame(c, cam, width, height);
    const bool wide = wide_grid(s);
    if (c->params.wgsl_compat && wide)
        hipLaunchKernelGGL((k_debug_pixel<true, true>), dim3(1), dim3(64), 0, c->stream, s, f, px, py, c->misc.as<float>());
    else if (c->params.wgsl_compat)
        hipLaunchKernelGGL((k_debug_pixel<true>), dim3(1), dim3(64), 0, c->stream, s, f, px, py, c->misc.as<float>());
    else if (wide)
        hipLaunchKernelGGL((k_debug_pixel<false, true>), dim3(1), dim3(64), 0, c->stream, s, f, px, py, c->misc.as<float>());
    else
        hipLaunchKernelGGL((k_debug_pixel<false>), dim3(1), dim3(64), 0, c->stream, s, f, px, py, c->misc.as<float>());
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipMemcpyAsync(info, c->misc.p, sizeof(RayDebugInfo), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

// ================================================================== primitive routines, all pairs

extern "C" int rt_intersect_aabb_pairs(rt_ctx* c, const float* origins, const float* dirs, uint32_t n_rays,
                                       const float* box_min, const float* box_max, uint32_t n_prims, float* out_t) {
    if (int r = use(c)) return r;
    const size_t n = (size_t)n_rays * n_prims;
    if (n == 0) return RT_OK;
    if (!origins || !dirs || !box_min || !box_max || !out_t) return fail(c, RT_ERR_INVALID, "rt_intersect_aabb_pairs: NULL argument");
    if (n > 0x7FFFFFFFull * 256) return fail(c, RT_ERR_LIMIT, "rt_intersect_aabb_pairs: too many pairs");
    // all arguments in device memory: run in place on the context's stream, no copies, no host synchronisation
    const bool dev = is_device_ptr(out_t);
    if (dev != is_device_ptr(origins) || dev != is_device_ptr(dirs) || dev != is_device_ptr(box_min) || dev != is_device_ptr(box_max))
        return fail(c, RT_ERR_INVALID, "rt_intersect_aabb_pairs: arguments must be all host or all device memory");
    const size_t rb = (size_t)n_rays * 12, pb = (size_t)n_prims * 12;
    const float *d_o = origins, *d_d = dirs, *d_mn = box_min, *d_mx = box_max;
    float* d_out = out_t;
    if (!dev) {
        RT_HIP(c, c->misc.reserve(2 * rb + 2 * pb + n * sizeof(float) + 64));
        char* base = c->misc.as<char>();
        d_o = (float*)base, d_d = (float*)(base + rb), d_mn = (float*)(base + 2 * rb), d_mx = (float*)(base + 2 * rb + pb);
        d_out = (float*)(base + ((2 * rb + 2 * pb + 15) & ~(size_t)15));  // rows on 16-byte boundaries when n_prims allows
        RT_HIP(c, hipMemcpyAsync((void*)d_o, origins, rb, hipMemcpyHostToDevice, c->stream));
        RT_HIP(c, hipMemcpyAsync((void*)d_d, dirs, rb, hipMemcpyHostToDevice, c->stream));
        RT_HIP(c, hipMemcpyAsync((void*)d_mn, box_min, pb, hipMemcpyHostToDevice, c->stream));
        RT_HIP(c, hipMemcpyAsync((void*)d_mx, box_max, pb, hipMemcpyHostToDevice, c->stream));
    }
    // about 16 K workgroups in all: enough to fill the chip many times over, few enough that a thread's boxes serve many rays
    const uint32_t gx = (n_prims + 1023u) / 1024u;
    const uint32_t gy = std::min<uint32_t>(std::min<uint32_t>(n_rays, 65535u), std::max<uint32_t>(1u, 16384u / gx));
    const dim3 grid(gx, gy);
    const bool wide = (n_prims & 3u) == 0 && ((uintptr_t)d_out & 15u) == 0;
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(256), 0, c->stream, d_o, d_d, n_rays, d_mn, d_mx, n_prims, d_out);
    };
    if (c->params.wgsl_compat)
        wide ? launch(k_aabb_pairs<true, true>) : launch(k_aabb_pairs<true, false>);
    else
        wide ? launch(k_aabb_pairs<false, true>) : launch(k_aabb_pairs<false, false>);
    RT_HIP(c, hipGetLastError());
    if (!dev) {
        RT_HIP(c, hipMemcpyAsync(out_t, d_out, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}

static int tri_pairs(rt_ctx* c, const float* origins, const float* dirs, uint32_t n_rays, const float* v012,
                     uint32_t n_prims, float* out_t, bool use_watertight) {
    if (int r = use(c)) return r;
    const size_t n = (size_t)n_rays * n_prims;
    if (n == 0) return RT_OK;
    if (!origins || !dirs || !v012 || !out_t) return fail(c, RT_ERR_INVALID, "rt_intersect_tri_pairs: NULL argument");
    if (n > 0x7FFFFFFFull * 256) return fail(c, RT_ERR_LIMIT, "rt_intersect_tri_pairs: too many pairs");
    const bool dev = is_device_ptr(out_t);
    if (dev != is_device_ptr(origins) || dev != is_device_ptr(dirs) || dev != is_device_ptr(v012))
        return fail(c, RT_ERR_INVALID, "rt_intersect_tri_pairs: arguments must be all host or all device memory");
    const size_t rb = (size_t)n_rays * 12, pb = (size_t)n_prims * 36;
    const float *d_o = origins, *d_d = dirs, *d_v = v012;
    float* d_out = out_t;
    if (!dev) {
        RT_HIP(c, c->misc.reserve(2 * rb + pb + n * sizeof(float) + 64));
        char* base = c->misc.as<char>();
        d_o = (float*)base, d_d = (float*)(base + rb), d_v = (float*)(base + 2 * rb), d_out = (float*)(base + 2 * rb + pb);
        RT_HIP(c, hipMemcpyAsync((void*)d_o, origins, rb, hipMemcpyHostToDevice, c->stream));
        RT_HIP(c, hipMemcpyAsync((void*)d_d, dirs, rb, hipMemcpyHostToDevice, c->stream));
        RT_HIP(c, hipMemcpyAsync((void*)d_v, v012, pb, hipMemcpyHostToDevice, c->stream));
    }
    const dim3 grid((uint32_t)((n + 255) / 256));
    if (use_watertight)
        hipLaunchKernelGGL(k_tri_pairs_watertight, grid, dim3(256), 0, c->stream, d_o, d_d, n_rays, d_v, n_prims, d_out);
    else if (c->params.wgsl_compat)
        hipLaunchKernelGGL(k_tri_pairs<true>, grid, dim3(256), 0, c->stream, d_o, d_d, n_rays, d_v, n_prims, d_out);
    else
        hipLaunchKernelGGL(k_tri_pairs<false>, grid, dim3(256), 0, c->stream, d_o, d_d, n_rays, d_v, n_prims, d_out);
    RT_HIP(c, hipGetLastError());
    if (!dev) {
        RT_HIP(c, hipMemcpyAsync(out_t, d_out, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}

extern "C" int rt_intersect_tri_pairs(rt_ctx* c, const float* origins, const float* dirs, uint32_t n_rays,
                                      const float* v012, uint32_t n_prims, float* out_t) {
    return tri_pairs(c, origins, dirs, n_rays, v012, n_prims, out_t, false);
}

extern "C" int rt_intersect_tri_pairs_watertight(rt_ctx* c, const float* origins, const float* dirs, uint32_t n_rays,
                                                 const float* v012, uint32_t n_prims, float* out_t) {
    return tri_pairs(c, origins, dirs, n_rays, v012, n_prims, out_t, true);
}

#ifdef RT_EXPERIMENTS
// development only: the order in which the persistent variants 5x hand out 8x8 pixel blocks
extern "C" int rt_dbg_set_order(rt_ctx* c, const uint32_t* order, size_t n) {
    if (int r = use(c)) return r;
    if (n > 65536) return fail(c, RT_ERR_LIMIT, "rt_dbg_set_order: at most 65536 blocks");
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(rt::g_order), order, n * sizeof(uint32_t)));
    return RT_OK;
}
// development only: per-wave timestamps of the instrumented variant
extern "C" int rt_dbg_wave_times(rt_ctx* c, unsigned long long* out, size_t n) {
    if (int r = use(c)) return r;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(rt::g_wave_t), n * sizeof(unsigned long long)));
    return RT_OK;
}
// development only: per-block cost as the planned kernel reported it
extern "C" int rt_dbg_wave_chain(rt_ctx* c, uint32_t* out, size_t n) {
    if (int r = use(c)) return r;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(rt::g_wave_chain), n * sizeof(uint32_t)));
    return RT_OK;
}
#endif

// ================================================================== timing

extern "C" int rt_timer_begin(rt_ctx* c) {
    if (int r = use(c)) return r;
    RT_HIP(c, hipEventRecord(c->ev0, c->stream));
    return RT_OK;
}

// What this GPU's memory delivers on the plainest access there is: 16 bytes per lane, read once, written once, past the
// caches. One element per lane and a workgroup per 4 KB is the fastest shape measured on the MI355X (6.5 TB/s over
// 2 GiB; a grid sized to the chip whose lanes loop with 2-8 loads in flight: 4.6-5.6 TB/s; hipMemcpyAsync: 4.9).
typedef uint32_t probe_vec __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_probe_copy(const probe_vec* __restrict__ src, probe_vec* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

extern "C" int rt_probe_copy(rt_ctx* c, const void* src, void* dst, size_t bytes) {
    if (int r = use(c)) return r;
    if (!src || !dst) return fail(c, RT_ERR_INVALID, "rt_probe_copy: NULL buffer");
    if ((bytes & 15u) || ((uintptr_t)src & 15u) || ((uintptr_t)dst & 15u))
        return fail(c, RT_ERR_INVALID, "rt_probe_copy: size and addresses must be multiples of 16 bytes");
    if (!is_device_ptr(src) || !is_device_ptr(dst)) return fail(c, RT_ERR_INVALID, "rt_probe_copy: device memory only");
    if (bytes == 0) return RT_OK;
    const size_t n = bytes / 16, wgs = (n + 255) / 256;
    if (wgs > 0x7FFFFFFFull) return fail(c, RT_ERR_LIMIT, "rt_probe_copy: at most 8 TiB in one call");
    hipLaunchKernelGGL(k_probe_copy, dim3((uint32_t)wgs), dim3(256), 0, c->stream, static_cast<const probe_vec*>(src),
                       static_cast<probe_vec*>(dst), n);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

extern "C" int rt_timer_end(rt_ctx* c, float* elapsed_ms) {
    if (int r = use(c)) return r;
    RT_HIP(c, hipEventRecord(c->ev1, c->stream));
    RT_HIP(c, hipEventSynchronize(c->ev1));
    float ms = 0.0f;
    RT_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return RT_OK;
}
