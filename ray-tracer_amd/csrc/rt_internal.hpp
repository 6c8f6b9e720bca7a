// Library-internal context and helpers shared by rt_api.hip and rt_build.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_mi355.h"
#include "rt_trace.hpp"

namespace rt {

// Growable device allocation. Capacity only grows; contents are NOT preserved across a grow.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, bytes);
            want = bytes;
        }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

}  // namespace rt

// A small table uploaded once and kept for the context's lifetime (kernels in flight on any stream may be reading
// it): the places of a shard's tiles in a round, or the assemble kernels' owner records. Found again by its key.
struct rt_table {
    std::vector<uint32_t> key;
    rt::DevBuf buf;
};

// Plan of expensive pixel blocks (BlockPlan in rt_trace.hpp) for one stream and one frame shape: three buffers used
// in rotation (read, written, cleared for the launch after), each holding two list lengths, the two block lists and a
// class byte per 8x8 block.
// Launches on one stream are ordered, so the plan a launch reads is complete; frames in flight on other streams have
// plans of their own.
struct rt_plan {
    hipStream_t stream = nullptr;
    uint32_t key[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // width, height, tile_w, tile_h, first, stride, local tiles, shadow
    const uint32_t* offsets = nullptr;
    uint32_t n_blocks = 0;
    rt::DevBuf buf[3];
    int cur = 0;
    unsigned long long last_use = 0;
};

// Block mask of the frames traced on one stream (k_block_mask in rt_cull.hpp): the mask of a frame is made and read by
// consecutive launches on that stream, so frames in flight on other streams need masks of their own.
// Two buffers: a frame's mask kernel fills one (all zero when it starts), the trace kernel that follows reads it and
// zeroes the other, which the next frame's mask kernel then fills - no launch of its own for the zeroing.
struct rt_cull {
    hipStream_t stream = nullptr;
    rt::DevBuf buf[2];
    uint32_t dirty[2] = {0, 0};      // leading words of each buffer that may be non-zero
    int cur = 0;                     // the buffer of the most recent mask
    uint32_t words = 0, blocks = 0;  // of the most recent mask made here
    unsigned long long last_use = 0;
};

// Block queues of the persistent-wave launches (rt_params.persistent_waves): one counter per stream that has launched
// one, each in a 128-byte line of its own, zeroed on that stream ahead of every launch.
struct rt_queues {
    rt::DevBuf counters;
    hipStream_t streams[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int used = 0;
    uint32_t resident_wgs = 0;  // 256-thread workgroups of the queue kernel the device holds at once
};

struct rt_ctx {
    int device = 0;
    rt_queues queues;
    std::vector<hipStream_t> traced_on;  // streams given a trace since the last rt_sync: what params.cull_empty_blocks = 1 decides by
    std::vector<rt_cull*> culls;
    rt_cull* last_cull = nullptr;  // the mask of the most recent launch that had one (null: it had none)
    std::vector<rt_plan*> plans;
    unsigned long long plan_clock = 0;
    rt_plan* last_plan = nullptr;  // the plan of the most recent planned launch
    std::vector<rt_table*> tables;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // stream in use (own_stream or the caller's)
    rt_params params;
    std::string err;

    // scene
    uint32_t nb = 0, nt = 0;
    uint32_t grid_nb = 0;  // boxes numbered ahead of the triangles in the current grid's ids (0 for a triangle-only grid)
    rt::DevBuf boxes;  // BoxData[nb]
    rt::DevBuf tri_v;  // float[nt*9]: v0,v1,v2
    rt::DevBuf tri_n;  // float4[nt]: geometric normal
    rt::DevBuf tri_mat;    // u32[nt]: u32(TriangleData::material_id)
    rt::DevBuf tri_uv;     // float[nt*6]: uv0, uv1, uv2
    rt::DevBuf materials;  // MaterialData[n_materials]
    uint32_t n_materials = 0;
    // 0: no material names a texture; 1: some do (the shaded picture samples); 2: some decide a hit's
    // acceptance by a texture's alpha (every trace kernel samples)
    int tex_mode = 0;
    rt::DevBuf texels;     // uchar4[tex_layers][tex_h][tex_w]
    uint32_t tex_w = 0, tex_h = 0, tex_layers = 0;
    // skin (rt_set_skin): per-vertex bind position, joint indices, weights; per-triangle vertex indices
    rt::DevBuf skin_pos, skin_joints, skin_weights, skin_tri, skin_mats;
    uint32_t skin_nv = 0, skin_max_joint = 0;
    bool has_skin = false;
    // bounding volume hierarchy (rt_build_bvh)
    rt::DevBuf bvh_nodes, bvh_leaf_first, bvh_prims, bvh_recs, bvh_work, bvh_parts;
    rt_bvh_stats bvh_stats{};
    uint32_t bvh_n_nodes = 0;
    bool has_bvh = false;

    // grid
    bool has_grid = false;
    GridMetadata meta;
    uint32_t n_cells = 0, n_refs = 0;
    rt::DevBuf cell_start;  // u32[n_cells+1]   the reference's cell lists (what rt_read_grid_csr returns)
    rt::DevBuf indices;     // u32[n_refs]
    // The lists the trace records are made from: the reference's, or those with duplicate triangles left out.
    bool grid_built = false;     // produced by rt_build_grid (lists hold the lowest ids, re-test masks are valid)
    int grid_mode = 0;
    uint32_t build_cap = 0;
    uint32_t n_refs_t = 0;
    // -1: not looked for yet (new scene or materials); 0: the last search left nothing out, so a rebuild after a
    // vertex update does not search again (an animated mesh pays for one search, not one per frame); 1: found
    int dup_state = -1;
    float dup_share = 0.0f;  // share of the references the last search left out
    const uint32_t* t_cell_start = nullptr;  // = cell_start or tc_start
    const uint32_t* t_indices = nullptr;     // = indices or tc_indices
    const uint32_t* t_ref_cell = nullptr;    // cell of every entry of t_indices (grids built here only)
    rt::DevBuf ref_cell, tc_ref_cell;
    rt::DevBuf tc_start, tc_indices, tc_pos, dd_keys, dd_keys2, dd_vals, dd_vals2, dd_flag, dd_head, dd_dup;
    rt::DevBuf cells;       // uint2[n_cells]
    rt::DevBuf refs;        // float4[3*n_refs]
    rt::DevBuf box_frame;   // float4[4*nb]

    // state of the rebuild that asks the host for nothing (build_grid_fast in rt_build.hip): valid after a general build
    bool fast_ready = false;
    bool last_build_fast = false;
    uint32_t fast_total = 0, fast_pairs = 0, fast_bordered = 0;
    float fast_cell = 0.0f;
    rt::DevBuf fast_meta, fast_scan;
    void* fast_landing = nullptr;  // 256 bytes of pinned host memory: where the rebuild's sizes arrive

    // binning scratch
    rt::DevBuf b_partial, b_counts, b_tmp_start, b_cursor, b_tmp_ids, b_tmp_sorted, b_tmp_cell, b_scan_tmp, b_flags,
        b_obj_n, b_obj_off, b_obj_sum;

    // per-call scratch
    rt::DevBuf out_p, out_s, in_o, in_d, stats, misc;
    rt_stats last_stats;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Frames in flight on several streams (rt_set_stream): see the rules at rt_set_stream in rt_api.hip.
    hipEvent_t ev_order = nullptr;
    hipEvent_t ev_scratch = nullptr;  // recorded behind launches that use the GPU's static scratch array (rt_trace_primary_noalloc)
    // multi-GPU (rt_comm_init): an ncclComm_t, kept untyped so that only rt_comm.hip needs RCCL's header
    void* comm = nullptr;
    int comm_rank = 0, comm_size = 0;
    bool scratch_busy = false;        // work using the context's per-frame scratch is queued on the current stream
    bool other_streams_busy = false;  // a stream used earlier may still hold trace work that reads the scene
};

namespace rt {

inline int fail(rt_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define RT_HIP(c, expr)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return rt::fail((c), RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                                     \
    } while (0)

// Waits for a stream whose remaining work is short (a few small kernels and a copy of some bytes): polls it for a
// few hundred microseconds before going to sleep in the driver, whose wake-up costs more than such a wait is long.
inline hipError_t wait_short(hipStream_t st) {
    for (int k = 0; k < 600; ++k) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
    }
    (void)hipGetLastError();  // "not ready" is no error to report later
    return hipStreamSynchronize(st);
}

// true when p points to device-accessible memory allocated through HIP
inline bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// implemented in rt_api.hip: the steps of installing a new scene (see there)
int begin_scene(rt_ctx* c);
int reserve_scene(rt_ctx* c, uint32_t nb, uint32_t nt);
int finish_scene(rt_ctx* c, uint32_t nb, uint32_t nt);
// implemented in rt_build.hip
int build_grid_device(rt_ctx* c, float cell_size, uint32_t max_per_cell, int mode);
// implemented in rt_api.hip: t_cell_start on device -> bordered cell records; reserves the reference records
int emit_cells(rt_ctx* c);
// implemented in rt_build.hip: trace records (cells, refs, masks, visibility) from the current cell lists,
// vertices and materials; leaves duplicate triangles out when the grid and the parameters allow it
int rebuild_trace_records(rt_ctx* c);

}  // namespace rt
