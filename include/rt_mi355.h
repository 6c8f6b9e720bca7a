/*
 * rt_mi355.h - C ABI of librt_mi355.so, the MI355X (gfx950) trace core.
 *
 * The library replaces ONE path of the reference renderer: everything between
 * "the host has Vec<BoxData>/Vec<TriangleData>, a grid and a CameraUniform" and
 * "per pixel, which object was hit and how far away" - i.e. what the reference
 * does with HierarchicalGrid::build + queue.write_buffer + dispatch_workgroups of
 * raytracer_unified.wgsl (src/renderer.rs:97-156, 822-868;
 * src/core/ray_tracing_layer.rs:193-238, 446-467).
 *
 * Conventions
 *  - plain pointers and sizes only; all structs are the wire structs of rt_types.h
 *  - every function returns 0 on success or a negative rt_status; the message is
 *    available from rt_last_error() until the next call on the same context
 *  - input pointers are read before the call returns (the caller keeps ownership,
 *    as with bytemuck::cast_slice + create_buffer_init in the reference)
 *  - pointers documented "host or device" are classified with
 *    hipPointerGetAttributes; host outputs are filled synchronously, device
 *    outputs are written asynchronously on the context's stream (rt_sync waits)
 *  - a context owns one HIP stream and is not re-entrant; rt_trace_* calls do not
 *    change scene state (the reference's render(&self))
 *  - there is no CPU fallback: every entry point that computes fails with
 *    RT_ERR_NO_DEVICE when no gfx950 device is usable
 */
#ifndef RT_MI355_H
#define RT_MI355_H

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument */
    RT_ERR_NO_DEVICE = -2, /* no usable GPU / HIP runtime error at create */
    RT_ERR_HIP = -3,       /* HIP runtime error during a call */
    RT_ERR_STATE = -4,     /* scene or grid missing */
    RT_ERR_LIMIT = -5,     /* a size exceeds a documented limit */
    RT_ERR_COMM = -6       /* RCCL missing or an RCCL call failed */
} rt_status;

/* Object id reported on a miss. */
#define RT_MISS 0xFFFFFFFFu

/* One traced ray's answer. id is the reference's unified object id
 * (raytracer_unified.wgsl:533-553: box i -> i, triangle k -> num_boxes + k). */
typedef struct rt_hit {
    uint32_t id;
    float t;
} rt_hit;

/* Compile-time constants of the reference, made run-time so parity runs and tuned
 * runs use one binary. rt_default_params() fills in the reference's values. */
typedef struct rt_params {
    float finest_cell_size;        /* 16.0            src/grid.rs:6 */
    uint32_t max_objects_per_cell; /* 8192            src/grid.rs:7 */
    uint32_t max_dda_steps;        /* 200             raytracer_unified.wgsl:520 */
    uint32_t wgsl_compat;          /* 0: CPU routines (Moller-Trumbore eps 1e-6, reject t < eps,
                                         zero direction component -> 1/copysign(1e-8))
                                         src/core/triangle_intersection.rs:38,74; src/math/ray.rs:9-11
                                      1: shader (eps 1e-5, accept t > eps, sign(0) = 0)
                                         raytracer_unified.wgsl:3,252,273,162-166 */
    uint32_t lod_cull;             /* 1               raytracer_unified.wgsl:147-155,453,539 */
    uint32_t moving_box_count;     /* 3               raytracer_unified.wgsl:446 */
    float fov;                     /* 0.785398        raytracer_unified.wgsl:4,643 */
    float light_direction[3];      /* (0.5,-1,0.3)    raytracer_unified.wgsl:5 */
    float shadow_offset;           /* 0.001           raytracer_unified.wgsl:687 */
    uint32_t collect_stats;        /* 0; 1 = count the cells and candidates of the REFERENCE algorithm (every list
                                      entry, no re-test skipping: slower kernels); 2 = count what the shipped kernels
                                      fetch (re-test skipping, duplicate removal, look-ahead, LDS staging as timed):
                                      rays and hits plus the three *_records_* fields of rt_stats */
    uint32_t skip_duplicate_triangles; /* 1: grids built by rt_build_grid leave a triangle out of the records the
                                         trace walks when a triangle with a lower id has bit-identical vertices and
                                         no alpha rule of its own: the copy is met right after the original in every
                                         cell list, yields the same t, and under the shader's strict '<' can neither
                                         replace it nor be accepted where it was rejected, so no ray's answer changes.
                                         (The reference's benchmark scene repeats 10 000 triangle positions.)
                                         rt_read_grid_csr / rt_get_grid still report the reference's full lists, and
                                         the counting kernels (collect_stats) walk them. 0: every list entry is tested */
    uint32_t heavy_block_threshold; /* 0 = off. A wave run time in microseconds (64 is a good start). When set,
                                         rt_trace_primary on triangle-only scenes remembers, per stream and frame
                                         shape, how long each 8x8 pixel block's wave ran in the last frame. Blocks at
                                         or above a threshold are traced with four lanes per ray by the workgroups
                                         that start first, blocks above half of it are started right after them, the
                                         rest follows in picture order. The threshold starts at this value and never
                                         drops below it; from frame to frame the library raises or lowers it so that a
                                         few hundred blocks are listed. The same answers bit for bit. For frames
                                         that wait for a few dense spots of the model (a frame of the animated phoenix:
                                         168 -> 77 us); a frame whose cost is spread evenly gains nothing and pays
                                         about 8 % for the bookkeeping. No reference counterpart */
    uint32_t cull_empty_blocks;     /* Before rt_trace_primary traces a frame of a triangle-only scene, a small kernel can
                                         project the grid's occupied cells (each grown by a quarter of a cell, far more
                                         than the float DDA of raytracer_unified.wgsl:492-588 can drift) onto the frame
                                         and mark the 8x8 pixel blocks they can be seen through; a block left unmarked
                                         has rays that meet no candidate list, so its 64 records are written as misses
                                         without a traversal. The same answers bit for bit.
                                         0: never. 1 (default): where the extra launch is hidden - once two or more OTHER
                                         streams of this context have been given a frame since the last rt_sync (three
                                         frames in flight; decided from the call sequence alone, so every run of the same
                                         calls launches the same kernels), or for four or more frames in one
                                         rt_render_frames / rt_trace_primary_frames call; a lone frame on an
                                         idle GPU waits for its longest waves either way and would only start later,
                                         and with two frames in flight the launch costs what the mask saves. 2: always.
                                         Applies when tiles are multiples of 8 pixels, the grid has at most 2^21 cells, the
                                         frame's mask at most 15 360 words (e.g. 7680 x 4096 pixels), and no plan of expensive blocks is kept; otherwise
                                         every ray walks the grid. No reference counterpart */
    uint32_t repack_candidates;     /* 1: rt_trace_primary and rt_trace_rays deal the (ray, candidate) pairs of a wave out again to its
                                         64 lanes at the DDA steps where that keeps more of them busy than each lane
                                         scanning its own cell's list (results meet per ray through a minimum over
                                         (t, list position), which is the in-order scan's answer: the same bits). For
                                         scenes whose neighbouring rays sit in DIFFERENT cells - cells much smaller than
                                         an 8x8 pixel block's footprint, many short lists: 10 M random triangles at
                                         3840x2160 2.50 -> 1.96 ms a frame, 4 M rays of a list in no order 3.41 -> 2.60
                                         ms. Where the rays of a block share their cells
                                         (the bundled models) the plain loop with its shared list buffer is faster
                                         (104 against 130 us), hence 0 by default. Triangle-only scenes with up to 4 GiB
                                         of reference records, no counters, no texture-decided hits, no plan of
                                         expensive blocks; ignored elsewhere. No reference counterpart */
    uint32_t persistent_waves;      /* 1: rt_trace_primary launches as many workgroups as the GPU holds at once and each
                                         workgroup takes 8x8 pixel blocks from a queue (an atomic counter, sixteen blocks a turn)
                                         until the frame is done, instead of one workgroup per 16x16 pixels dealt out by
                                         the hardware dispatcher. The same records bit for bit. Measured slower than the
                                         plain launch wherever tried (the dispatcher already hands a workgroup to whichever
                                         slot frees first; the queue adds its atomics and, because the block body sits in
                                         a loop, seven resident waves per SIMD instead of eight), hence 0 by default; for
                                         callers that must bound the number of workgroups in flight. Triangle-only
                                         scenes, no counters, no texture-decided hits, no plan of expensive blocks, at most
                                         eight streams per context; ignored elsewhere. No reference counterpart */
} rt_params;

/* Which pixels of a W x H frame one call traces. The frame is cut into
 * tile_w x tile_h tiles numbered row-major, and the tiles into rounds of `stride`
 * consecutive tiles. With owner == NULL the call traces tile `first` of every round
 * (tiles first, first+stride, first+2*stride, ...). With an owner table - `stride`
 * entries of host memory (at most 65 536), owner[o] = the rank that traces tile o of every round, or RT_OWNER_SKY for
 * a tile nobody traces (rt_tile_flags) - it traces the tiles whose owner is `first`, so ranks can own unequal shares of the
 * frame, spread over it as finely as the table says: a rank that also gathers and
 * assembles can be given less to trace (rt_assemble_frames takes the same table).
 * Either way the tiles are packed in increasing tile order: local tile j occupies
 * out[j*tile_w*tile_h ...] in tile-local row-major order (pixels of an edge tile that
 * fall outside the frame are written as misses). A NULL shard means the whole frame,
 * row-major. Tables of up to 4096 entries. */
typedef struct rt_shard {
    uint32_t tile_w, tile_h;
    uint32_t first, stride;
    const uint32_t* owner;
} rt_shard;

/* Work counters of the most recent stats-enabled trace call. */
typedef struct rt_stats {
    uint64_t rays;          /* primary + shadow (or list) rays traced */
    uint64_t primary_hits;
    uint64_t shadow_rays;
    uint64_t shadow_hits;
    uint64_t cells_visited; /* DDA iterations (cell records read) */
    uint64_t tri_tests;     /* Moller-Trumbore candidates tested */
    uint64_t box_tests;     /* slab candidates tested (moving-box pre-pass included) */
    /* collect_stats = 2 only (0 otherwise): what the trace kernels asked the memory system for */
    uint64_t cell_records_fetched; /* 8-byte cell records loaded, summed over lanes */
    uint64_t ref_records_fetched;  /* 48-byte reference records loaded by a lane for itself (look-ahead included) */
    uint64_t ref_records_staged;   /* 48-byte reference records copied into LDS once for a whole wave */
    uint64_t cell_records_uniform; /* of cell_records_fetched: fetched in a step in which every active lane of the wave
                                      asked for the same record (the rays of a wave's 8x8 pixels walk together) */
} rt_stats;

/* ---- lifecycle ------------------------------------------------------------------ */

/* Fills *p with the reference's constants. */
void rt_default_params(rt_params* p);

/* Replaces wgpu Instance/Adapter/Device creation (src/renderer.rs:43-53).
 * params may be NULL (defaults). */
int rt_create(rt_ctx** out, int device_id, const rt_params* params);
void rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx);
/* Message of the most recent failed rt_create (no context exists then). */
const char* rt_create_error(void);

/* Run on a caller-owned hipStream_t (e.g. torch's current stream) instead of the
 * context's own; NULL restores the context's stream. Does not block the host: alternating
 * between streams keeps several frames in flight (each with its own output buffers), which
 * hides the long tail of a frame behind the next one. The context orders its own per-frame
 * scratch across the switch on the GPU, and rt_set_scene / rt_set_grid_* / rt_build_grid /
 * rt_update_vertices wait for every stream used before; rt_sync waits for all of them. */
int rt_set_stream(rt_ctx* ctx, void* hip_stream);
/* Blocks until everything queued by this context has finished. */
int rt_sync(rt_ctx* ctx);

/* Host helper mirroring Camera::forward/right/up/to_uniform (src/camera.rs:51-54,76-122) for
 * callers that hold a yaw/pitch camera rather than a ready CameraUniform. */
void rt_camera_to_uniform(const float position[3], float yaw, float pitch, float time, float screen_height,
                          float fov, int show_grid, CameraUniform* out);

/* ---- scene and grid --------------------------------------------------------------- */

/* Replaces the box/triangle create_buffer_init uploads (src/renderer.rs:119-145).
 * boxes/tris: host pointers, nb/nt may be 0. Invalidates any grid. */
int rt_set_scene(rt_ctx* ctx, const BoxData* boxes, uint32_t nb, const TriangleData* tris, uint32_t nt);

/* Number of boxes and triangles of the current scene (a rank that received its scene through rt_broadcast_scene
 * learns the sizes here). */
int rt_get_scene(rt_ctx* ctx, uint32_t* nb, uint32_t* nt);

/* Replaces the "Fine Cells" upload (src/renderer.rs:99,113-117) for a caller that
 * already ran HierarchicalGrid::to_gpu_buffers (src/grid.rs:301-358): fine_cells is
 * the dense FineCellData[n_cells] array (host). Converted to a compact cell list. */
int rt_set_grid_dense(rt_ctx* ctx, const GridMetadata* meta, const FineCellData* fine_cells, size_t n_cells);

/* Same for TriangleGrid::to_gpu_buffers (src/grid_triangles.rs:215-261): 1040-byte cells. */
int rt_set_grid_dense_tri(rt_ctx* ctx, const GridMetadata* meta, const TriFineCellData* fine_cells, size_t n_cells);

/* Compact form of the same information: cell_start[n_cells+1], indices[cell_start[n_cells]] (host). */
int rt_set_grid_csr(rt_ctx* ctx, const GridMetadata* meta, const uint32_t* cell_start, const uint32_t* indices);

typedef enum rt_grid_mode {
    RT_GRID_HIERARCHICAL = 0, /* HierarchicalGrid::build  src/grid.rs:110-216 */
    RT_GRID_TRIANGLES = 1     /* TriangleGrid::build      src/grid_triangles.rs:29-124 */
} rt_grid_mode;

/* Replaces HierarchicalGrid::build / TriangleGrid::build with an on-GPU binning pass
 * over the current scene: same bounds, dimensions, id order and per-cell cap.
 * cell_size <= 0 and max_per_cell == 0 select the context's params. */
int rt_build_grid(rt_ctx* ctx, float cell_size, uint32_t max_per_cell, int mode);

/* What the most recent rt_trace_primary launch that keeps a plan (params.heavy_block_threshold) decided for the next
 * launch of the same frame shape on the same stream: n_heavy 8x8 pixel blocks to be traced four lanes per ray, n_early
 * to be started ahead of picture order as they are, and the run time (microseconds) that launch will call heavy. The
 * last two may be NULL. Waits for that stream. Zeros (and the parameter) when no such launch has happened. No
 * reference counterpart. */
int rt_get_heavy_blocks(rt_ctx* ctx, uint32_t* n_heavy, uint32_t* n_early, uint32_t* threshold_us);

/* Of the 8x8 pixel blocks of the frame of the most recent rt_trace_primary launch on this context, how many its block
 * mask (params.cull_empty_blocks) kept for tracing; zeros when that launch had no mask. Waits for that stream. No
 * reference counterpart. */
int rt_get_cull_stats(rt_ctx* ctx, uint32_t* blocks_total, uint32_t* blocks_kept);

/* Checked builds only (make -C ray-tracer_amd/csrc checked -> librt_mi355_checked.so: the trace kernels compare every
 * record, cell and triangle index they form with the array it addresses, count violations instead of faulting, and go
 * on): how many violations the launches of this process have counted so far, and the first one's kind and two
 * operands (first3 may be NULL). Waits for the device. The shipped library compiles no checks and returns
 * RT_ERR_STATE. No reference counterpart (the reference relies on the shader compiler's robust buffer access). */
int rt_checked_violations(rt_ctx* ctx, uint32_t* count, uint32_t first3[3]);

/* How the current grid was made: 0 = rt_build_grid's general path (sizes read back by the host between passes,
 * duplicate search, any scene), 1 = its rebuild path for scenes of up to 65 536 objects whose grid has been built
 * before with the same mode, cell size and cap (rt_update_vertices / rt_skin_vertices + rt_build_grid, the animated
 * case): every size stays on the device, launches cover the capacity the previous build left behind, one 64-byte
 * read-back at the end; a frame that does not fit falls back to 0 by itself. -1 = the caller supplied the grid.
 * Same cell lists either way. No reference counterpart. */
int rt_get_build_path(rt_ctx* ctx, int* path);

/* Metadata and sizes of the current grid. */
int rt_get_grid(rt_ctx* ctx, GridMetadata* meta, uint32_t* n_cells, uint32_t* n_refs);
/* Number of reference records the trace kernels walk for the current grid: n_refs of rt_get_grid minus the
 * references left out by params.skip_duplicate_triangles. No reference counterpart. */
int rt_get_trace_refs(rt_ctx* ctx, uint32_t* n_trace_refs);
/* Copies the current grid's cell list back to the host: cell_start[n_cells+1], indices[n_refs]. */
int rt_read_grid_csr(rt_ctx* ctx, uint32_t* cell_start, uint32_t* indices);

/* Overwrites the vertices of triangles [first, first+count) with xyz9[count*9]
 * (v0,v1,v2 per triangle; host or device). The grid must be rebuilt afterwards
 * (rt_build_grid); the reference has no counterpart - it re-uploads whole scenes
 * (src/main.rs:152-168). */
int rt_update_vertices(rt_ctx* ctx, const float* xyz9, uint32_t first, uint32_t count);

/* ---- tracing ------------------------------------------------------------------------ */

/* Replaces queue.write_buffer(camera) + dispatch_workgroups(ceil(W/8),ceil(H/8),1)
 * (src/renderer.rs:822-868; src/core/ray_tracing_layer.rs:446-467), geometry part of
 * raytracer_unified.wgsl:625-653 (ray generation) and :430-588 (trace_ray).
 * out_primary (host or device, required): closest hit per pixel.
 * out_shadow (host or device, may be NULL): for every pixel whose primary ray hit,
 * one any-hit ray from hit_point + n*shadow_offset towards -normalize(light_direction)
 * (n = geometric normal turned against the primary ray); id/t of the first accepted
 * occluder in traversal order, or RT_MISS. Pixels without a primary hit get RT_MISS. */
int rt_trace_primary(rt_ctx* ctx, const CameraUniform* camera, uint32_t width, uint32_t height,
                     const rt_shard* shard, rt_hit* out_primary, rt_hit* out_shadow);

/* rt_trace_primary for up to 16 consecutive frames of a camera path in one call (the reference dispatches once per frame,
 * src/renderer.rs:857-868: no counterpart; for renderers that trace ahead of display or several views of a scene).
 * cameras[j] is frame j; its records are written frame_stride records after frame j-1's (0 = packed: one frame's worth).
 * Triangle-only scenes with device outputs go out as ONE launch in which a frame's last waves overlap the next frame's
 * bulk (and one block-mask launch for the group, params.cull_empty_blocks); every other case is traced frame by frame.
 * The same records as n_frames calls of rt_trace_primary. */
int rt_trace_primary_frames(rt_ctx* ctx, const CameraUniform* cameras, uint32_t n_frames, uint32_t width, uint32_t height,
                            const rt_shard* shard, rt_hit* out_primary, rt_hit* out_shadow, size_t frame_stride);

/* rt_trace_primary for callers that cannot allocate device memory while they run: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc. No counterpart in the reference. Same arguments, same
 * results; it calls no hipMalloc / hipFree and no allocating routine. The one buffer a trace needs beyond its inputs
 * and outputs - the per-frame box records (interpolated bounds and LOD-cull outcomes, 64 bytes per box) - lives in
 * a static device array of RT_NOALLOC_MAX_BOXES records (512 KiB) that every GPU the library runs on holds until
 * the process ends. Restrictions, each reported as an error, nothing is silently downgraded:
 *   - out_primary / out_shadow must be device memory (a host output would need a staging buffer)   RT_ERR_INVALID
 *   - scenes with more than RT_NOALLOC_MAX_BOXES boxes                                               RT_ERR_LIMIT
 *   - params.collect_stats must be 0                                                                 RT_ERR_STATE
 *   - a shard with an owner table only if an allocating call has used that table before             RT_ERR_STATE
 * Every call on a GPU shares that GPU's array, whichever context it comes from: calls on one context are ordered
 * as rt_set_stream describes; a call from ANOTHER context whose scene has boxes is ordered on the GPU behind the
 * launches of the context that used the array last (its stream waits for an event that context recorded; events are
 * made in rt_create, nothing is allocated here), so contexts can share a GPU, taking turns at the array - calls made
 * at the same time from different host threads are served one after the other. The array keeps the last call's
 * records; every call rewrites all records it reads, so nothing needs clearing between calls. */
#define RT_NOALLOC_MAX_BOXES 8192u
int rt_trace_primary_noalloc(rt_ctx* ctx, const CameraUniform* camera, uint32_t width, uint32_t height,
                             const rt_shard* shard, rt_hit* out_primary, rt_hit* out_shadow);

/* Ray-list mode (the reference's CPU benches trace explicit ray lists:
 * examples/massive_benchmark.rs:90-98). origins/dirs: n*3 floats each (host or device).
 * camera supplies position/time/lod for the box LOD cull and may be NULL when the
 * scene has no boxes. any_hit != 0 stops at the first accepted intersection. Lists of up to 32 768 rays through a
 * triangle-only scene are walked four lanes to a ray (same records; a short list's time is its longest ray's chain). */
int rt_trace_rays(rt_ctx* ctx, const CameraUniform* camera, const float* origins, const float* dirs,
                  uint32_t n, rt_hit* out, int any_hit);

/* Reassembles a frame from shard outputs: gathered holds, rank after rank, each rank's
 * packed output of rt_trace_primary with shard {tile_w,tile_h,first=rank,stride=n_ranks};
 * rank r's block starts at record r*rank_stride (rank_stride = 0 means blocks are packed
 * back to back, each padded to ceil(n_tiles/n_ranks) tiles). image: width*height row-major.
 * Both device pointers. No reference counterpart (single GPU there). */
int rt_assemble_tiles(rt_ctx* ctx, const rt_hit* gathered, uint32_t width, uint32_t height,
                      uint32_t tile_w, uint32_t tile_h, uint32_t n_ranks, size_t rank_stride, rt_hit* image);

/* Fills *info for one pixel the way the shader's debug pixel does
 * (raytracer_unified.wgsl:695-709), geometry fields only (hit_color is zero). */
int rt_debug_pixel(rt_ctx* ctx, const CameraUniform* camera, uint32_t width, uint32_t height,
                   uint32_t px, uint32_t py, RayDebugInfo* info);

/* ---- several GPUs: one process per GPU, RCCL over xGMI ----------------------------------- */

/* None of these has a counterpart in the reference, which renders on one GPU (src/renderer.rs:857-868 dispatches the
 * whole frame). A frame shards by image tile (rt_shard); the scene is replicated once per scene load and every frame's
 * tiles are gathered on one rank. RCCL is bound at run time (dlopen of the copy already in the process, else the
 * system's); without it these calls fail with RT_ERR_COMM and everything else works.
 *
 * rt_comm_unique_id: on ONE rank; the caller hands the RT_COMM_ID_BYTES bytes to the other ranks by any means it has
 *   (a file, a socket, MPI, an environment variable).
 * rt_comm_init: collective over all n_ranks contexts (one per process, each on its own GPU): ncclCommInitRank.
 * rt_broadcast_scene: collective. Rank `root` passes its host arrays as for rt_set_scene (the other ranks' array
 *   arguments are ignored); afterwards every rank holds that scene exactly as if it had called rt_set_scene itself:
 *   the bytes cross PCIe once, on the root, and go from GPU to GPU straight into the scene buffers. Each rank then
 *   bins its own grid (rt_build_grid is deterministic, so the grids are identical without being shipped).
 * rt_gather_frames: collective, queued on the context's stream like a launch (it does not block the host). Every rank
 *   passes `bytes` bytes of device memory (its packed tiles from rt_trace_primary / rt_render[_frames] with its
 *   shard); on `root`, gathered[r * bytes ...] receives rank r's block (gathered is ignored elsewhere). N-1 grouped
 *   point-to-point transfers, one per link into the root. rt_assemble_* then scatters the tiles into frames.
 *   Calls on one context must be issued in the same order on every rank. */
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(uint8_t* id /* RT_COMM_ID_BYTES */);
int rt_comm_init(rt_ctx* ctx, const uint8_t* id /* RT_COMM_ID_BYTES */, int rank, int n_ranks);
int rt_comm_info(rt_ctx* ctx, int* rank, int* n_ranks);
int rt_comm_destroy(rt_ctx* ctx); /* also done by rt_destroy */
int rt_broadcast_scene(rt_ctx* ctx, int root, const BoxData* boxes, uint32_t nb, const TriangleData* tris, uint32_t nt);
int rt_gather_frames(rt_ctx* ctx, int root, const void* local, size_t bytes, void* gathered);

/* ---- bounding volume hierarchy ----------------------------------------------------------- */

/* BVHStats (src/core/bvh.rs:34-40). */
typedef struct rt_bvh_stats {
    uint64_t node_count, leaf_count, max_depth, total_primitives;
} rt_bvh_stats;

/* BVHNode::build over the scene's triangles (src/core/bvh.rs:44-232: binned SAH, 12 buckets, leaves of up
 * to 4, the partition of :212-232 element for element), built on the GPU from the current vertices, level by level;
 * the tree is the reference's node for node (same boxes, same leaves, same order of a leaf's primitives). Rebuilt by
 * the caller after rt_set_scene, rt_update_vertices or rt_skin_vertices. */
int rt_build_bvh(rt_ctx* ctx);
/* The same tree built by the reference's recursion on the host (one thread) and uploaded: what rt_build_bvh is checked
 * against node for node, and its timing baseline. */
int rt_build_bvh_host(rt_ctx* ctx);
int rt_get_bvh_stats(rt_ctx* ctx, rt_bvh_stats* stats);
/* The tree as the traversal reads it (BVHNode flattened, src/core/bvh.rs:12-31): nodes = 8 floats per node in pre-order
 * - min.xyz, the bits of the "skip" index (the node after this one's subtree), max.xyz, the bits of the leaf's primitive
 * count (0: internal node; the left child is the next node, the right one follows the left subtree); leaf_first[node] =
 * first entry of a leaf in prims; prims = triangle indices in leaf order. Host memory, sized from rt_get_bvh_stats
 * (node_count, total_primitives); any pointer may be NULL. */
int rt_read_bvh(rt_ctx* ctx, float* nodes, uint32_t* leaf_first, uint32_t* prims);
/* traverse_bvh_triangles of the reference's benchmark (examples/massive_benchmark.rs:226-270,311-320) for a list
 * of rays: both children of every node whose box the ray meets, Moller-Trumbore in the leaves, smallest t.
 * out_t[i] = t, or -1 when nothing is hit (the example's None). origins/dirs/out_t all host or all device.
 * Two things are done differently without changing a bit of out_t: a triangle that is an exact copy of the one before
 * it in its leaf is not tested again (rt_params.skip_duplicate_triangles; the same t cannot lower the minimum), and up
 * to 262 144 rays are walked by 32 lanes each, a lane per subtree (the minimum over the lanes is the ray's). */
int rt_trace_rays_bvh(rt_ctx* ctx, const float* origins, const float* dirs, uint32_t n, float* out_t);

/* ---- skinning (animated models) --------------------------------------------------------- */

/* The reference loads animated models in their bind pose only (src/loaders/gltf_triangles.rs:179-283 walks
 * the node matrices and never reads the skin); these two calls are what its `models/animated` asset needs to
 * move: glTF linear-blend skinning evaluated on the GPU straight into the vertex array that rt_build_grid bins,
 * so an animated frame is rt_skin_vertices + rt_build_grid + rt_trace_primary with no vertex traffic over PCIe.
 * rt_set_skin: per-vertex bind positions (n*3), joint indices (n*4, u16), weights (n*4) and, per triangle of
 * the scene given to rt_set_scene, three vertex indices. Host pointers, copied.
 * rt_skin_vertices: one column-major 4x4 matrix per joint (global joint transform times inverse bind matrix;
 * host or device memory). vertex = ((S.x*p.x + S.y*p.y) + S.z*p.z) + S.w with S = sum of weight * matrix in
 * joint-slot order. Invalidates the grid, like rt_update_vertices. */
int rt_set_skin(rt_ctx* ctx, const float* positions, const uint16_t* joints, const float* weights, uint32_t n_vertices,
                const uint32_t* tri_indices, uint32_t n_triangles);
int rt_skin_vertices(rt_ctx* ctx, const float* joint_matrices, uint32_t n_joints);
/* Current triangle vertices, nt*9 floats (v0, v1, v2), to host or device memory. */
int rt_read_vertices(rt_ctx* ctx, float* xyz9);

/* ---- shaded picture ------------------------------------------------------------------- */

/* What main() of the shader does around trace_ray, beyond its constants. */
typedef struct rt_render_opts {
    uint32_t max_bounces;     /* MAX_BOUNCES (src/raytracer_unified.wgsl:656): 8 */
    uint32_t shadows;         /* 0 = the shader's lighting. 1 = a shaded point whose shadow ray (as defined at
                                 rt_trace_primary) is occluded receives no diffuse light - not in the shader */
    uint32_t debug_enabled;   /* DebugParams (src/types.rs:187-193): highlight of the debug pixel, wgsl:693-716 */
    uint32_t debug_pixel[2];
    uint32_t pixel_format;    /* RT_PIXEL_RGBA8 (0): 4 bytes per pixel, the shader's texture. RT_PIXEL_RGB8 (1): the same
                                 picture without its constant alpha byte, 3 bytes per pixel - a quarter less to move
                                 when the framebuffer is gathered from other GPUs (rt_assemble_frames_rgb8 puts the
                                 alpha byte back); no reference counterpart */
    uint32_t frame_stride;    /* rt_render_frames: pixels from one frame's output to the next; 0 = packed (the pixels
                                 one frame writes). Ranks whose shards differ in size use a common stride, so that the
                                 gathered blocks have one layout */
    uint32_t plain_bounces;   /* 0 (default): in pictures with reflection bounces of triangle-only scenes the scattered
                                 rays of a wave share their candidate lists from the first reflection on (k_render_deal;
                                 with texture-decided alpha only from six bounces up, where it pays) - the same picture
                                 byte for byte, the 8-bounce picture of the bundled static model 179 -> 167 us, without
                                 its alpha textures 166 -> 148. 1: every ray scans its own lists (k_render). No reference
                                 counterpart */
} rt_render_opts;
#define RT_PIXEL_RGBA8 0u
#define RT_PIXEL_RGB8 1u
void rt_default_render_opts(rt_render_opts* o);

/* Replaces the "Materials" storage buffer (src/renderer.rs:147-156): MaterialData records
 * indexed by TriangleData::material_id. Without it every triangle is the shader's grey
 * (wgsl:366-370). Evaluated as the shader's triangle test does (wgsl:279-366): base colour
 * (factor times the base-colour texture when texture_index >= 0), the alpha rules (MASK below
 * alpha_cutoff / BLEND below 0.01 discard the hit, so the ray goes on to what lies behind) -
 * for rt_trace_primary, rt_trace_rays and rt_render alike, whether the materials arrive before
 * or after the grid - and, in rt_render, the normal map, emissive factor and texture, and
 * metallic as reflectivity. Host pointer, copied. */
int rt_set_materials(rt_ctx* ctx, const MaterialData* materials, uint32_t n);

/* TextureData of the loader (src/loaders/gltf_triangles.rs:15-19): width * height RGBA8 texels, row-major. */
typedef struct rt_texture {
    const uint8_t* rgba8;
    uint32_t width, height;
} rt_texture;

/* Replaces the texture array and sampler of the renderer (src/renderer.rs:158-259): texture i becomes layer i
 * of one Rgba8Unorm array as wide and high as the largest texture, each written to the top-left corner of its
 * layer (the rest of a smaller texture's layer stays zero and uv still spans the whole layer, as in the
 * reference); sampled with repeat addressing and a linear filter at mip level 0, a layer index beyond the array
 * clamped to the last layer. n = 0 is the renderer's 1x1 white dummy. Filter arithmetic is f32:
 * ((c00*(1-fx) + c10*fx)*(1-fy) + (c01*(1-fx) + c11*fx)*fy with c = texel/255 and fx, fy the fractions of
 * uv*size - 0.5 (a GPU texture unit weights in fixed point, so a wgpu run can differ in a colour's last bits).
 * Host pointers, copied. */
int rt_set_textures(rt_ctx* ctx, const rt_texture* textures, uint32_t n);

/* Replaces dispatch_workgroups + the Rgba8Unorm output texture (src/renderer.rs:822-868,
 * src/core/ray_tracing_layer.rs:436-493 which returns these bytes): ray generation, up to
 * max_bounces trace_ray calls per pixel with mirror reflection, sky, ambient + directional
 * lighting, grid overlay (camera.show_grid), debug-pixel highlight; out_rgba8 gets 4 bytes per
 * pixel (r, g, b, 255), row-major, or tile-packed exactly like rt_trace_primary's records
 * when shard is given. Host or device pointer, as for rt_trace_primary. */
int rt_render(rt_ctx* ctx, const CameraUniform* camera, uint32_t width, uint32_t height, const rt_shard* shard,
              const rt_render_opts* opts, uint8_t* out_rgba8);

/* rt_render for up to 16 consecutive frames of a camera path in ONE launch: cameras[j] renders into
 * out_rgba8 + j * (bytes of one frame's output), everything else as in rt_render. The reference dispatches one
 * frame at a time (src/renderer.rs:857-868); a multi-GPU caller that owns only 1/N of every frame's tiles uses
 * this so that the few expensive pixel blocks of one frame (one long dependent chain each) overlap the bulk of
 * the other frames instead of leaving its GPU idle, and so that one framebuffer gather carries many frames. */
int rt_render_frames(rt_ctx* ctx, const CameraUniform* cameras, uint32_t n_frames, uint32_t width, uint32_t height,
                     const rt_shard* shard, const rt_render_opts* opts, uint8_t* out_rgba8);

/* rt_assemble_tiles for rt_render's 4-byte pixels. No reference counterpart. */
int rt_assemble_tiles_rgba8(rt_ctx* ctx, const uint8_t* gathered, uint32_t width, uint32_t height,
                            uint32_t tile_w, uint32_t tile_h, uint32_t n_ranks, size_t rank_stride, uint8_t* image);

/* The same for n_frames frames gathered together (rt_render_frames on every rank): frame j of rank r starts at
 * pixel r*rank_stride + j*frame_stride of gathered; images receives n_frames row-major pictures back to back.
 * No reference counterpart. */
int rt_assemble_frames_rgba8(rt_ctx* ctx, const uint8_t* gathered, uint32_t width, uint32_t height, uint32_t tile_w,
                             uint32_t tile_h, uint32_t n_ranks, uint32_t n_frames, size_t rank_stride,
                             size_t frame_stride, uint8_t* images);

/* The same for frames rendered with pixel_format = RT_PIXEL_RGB8: gathered holds 3-byte pixels (strides in
 * pixels), images receives 4-byte RGBA8 pictures with alpha 255. No reference counterpart. */
int rt_assemble_frames_rgb8(rt_ctx* ctx, const uint8_t* gathered, uint32_t width, uint32_t height, uint32_t tile_w,
                            uint32_t tile_h, uint32_t n_ranks, uint32_t n_frames, size_t rank_stride,
                            size_t frame_stride, uint8_t* images);

/* The general form of the two above, for ranks that own unequal shares of the frame: owner is the table every rank
 * rendered with (rt_shard.owner, `round` entries, values below n_ranks; rank r passed first = r, stride = round);
 * NULL means one tile per rank per round, as above (round is ignored). A gathering rank spends time on the collective
 * and on this call that the others do not, so it is given fewer tiles. pixel_format: how the gathered pixels are
 * stored (RT_PIXEL_RGBA8 or RT_PIXEL_RGB8). No reference counterpart. */
int rt_assemble_frames(rt_ctx* ctx, const uint8_t* gathered, uint32_t pixel_format, uint32_t width, uint32_t height,
                       uint32_t tile_w, uint32_t tile_h, uint32_t n_ranks, const uint32_t* owner, uint32_t round,
                       uint32_t n_frames, size_t rank_stride, size_t frame_stride, uint8_t* images);

/* Tiles of a frame that show nothing but sky: flags[t] = 0 when no 8x8 pixel block of tile t (row-major, tile_w x tile_h
 * pixels, both multiples of 8) can meet geometry from this camera - the block mask of params.cull_empty_blocks, made for
 * the whole frame whatever the parameter says - and 1 otherwise; all 1 when the frame can have no mask (boxes, a grid
 * or frame too large, a camera the bound on the float DDA's drift does not cover). A sky tile is a function of the camera
 * alone, so in a multi-GPU run nobody has to render or send it: the ranks deal the other tiles among themselves with
 * an owner table whose sky entries are RT_OWNER_SKY, and the gathering rank's rt_assemble_frames_sky fills those from the
 * camera. Synchronous (one small launch and a read-back): for cameras known ahead of the frames. flags: host memory, one
 * byte per tile. No reference counterpart. */
#define RT_OWNER_SKY 0xFFFFFFFFu
int rt_tile_flags(rt_ctx* ctx, const CameraUniform* camera, uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h,
                  uint8_t* flags);

/* rt_assemble_frames for owner tables with RT_OWNER_SKY entries: those tiles are not read from `gathered` but written as
 * the shader's sky (src/raytracer_unified.wgsl:591-596) seen through cameras[j] for frame j - byte for byte what
 * rt_render_frames writes for a pixel whose ray meets nothing (no debug-pixel highlight: callers that use one render
 * every tile). cameras: host memory, n_frames entries, n_frames <= 64. No reference counterpart. */
int rt_assemble_frames_sky(rt_ctx* ctx, const uint8_t* gathered, uint32_t pixel_format, uint32_t width, uint32_t height,
                           uint32_t tile_w, uint32_t tile_h, uint32_t n_ranks, const uint32_t* owner, uint32_t round,
                           const CameraUniform* cameras, uint32_t n_frames, size_t rank_stride, size_t frame_stride,
                           uint8_t* images);

/* Counters of the most recent call made with params.collect_stats = 1. rt_trace_primary and rt_trace_rays fill
 * every field (their counting kernels walk the reference's full lists). rt_render / rt_render_frames fill rays,
 * primary_hits, shadow_rays and shadow_hits only: cells_visited, tri_tests and box_tests read 0 after them (the
 * picture kernels have no counting instantiation). */
int rt_get_stats(rt_ctx* ctx, rt_stats* stats);
int rt_set_collect_stats(rt_ctx* ctx, int mode); /* 0, 1 or 2: params.collect_stats */

/* ---- primitive routines, all pairs ---------------------------------------------------- */

/* GPU counterparts of math::intersect_aabb (src/math/ray.rs:3-36) and
 * moller_trumbore_intersect (src/core/triangle_intersection.rs:31-82) evaluated for
 * every (ray, primitive) pair, the shape of the criterion loops
 * (benches/bounding_volumes.rs:162-175). Pointers are all host memory (copied in and out, the
 * call returns when out_t is complete) or all device memory (runs in place, asynchronously on
 * the context's stream).
 * out_t[r*n_prims + p]: slab result (-1 on miss) / triangle t (-1 on miss). */
int rt_intersect_aabb_pairs(rt_ctx* ctx, const float* origins, const float* dirs, uint32_t n_rays,
                            const float* box_min, const float* box_max, uint32_t n_prims, float* out_t);
int rt_intersect_tri_pairs(rt_ctx* ctx, const float* origins, const float* dirs, uint32_t n_rays,
                           const float* v012, uint32_t n_prims, float* out_t);
/* Same, with watertight_intersect (src/core/triangle_intersection.rs:101-181) as the triangle test. */
int rt_intersect_tri_pairs_watertight(rt_ctx* ctx, const float* origins, const float* dirs, uint32_t n_rays,
                                      const float* v012, uint32_t n_prims, float* out_t);

/* ---- timing --------------------------------------------------------------------------- */

/* hipEvent pair on the stream the kernels are launched on. */
int rt_timer_begin(rt_ctx* ctx);
int rt_timer_end(rt_ctx* ctx, float* elapsed_ms); /* synchronises on the end event */

/* Measurement aid without a counterpart in the reference: queues a plain copy of `bytes` bytes of device memory (16
 * bytes per lane, read once, written once) on the context's stream. Timed between rt_timer_begin / rt_timer_end over
 * buffers far larger than the caches it gives what this GPU's HBM delivers, the figure to hold a kernel's measured
 * traffic against beside the 8 TB/s of the data sheet. Size and both addresses must be multiples of 16. */
int rt_probe_copy(rt_ctx* ctx, const void* src, void* dst, size_t bytes);

#ifdef __cplusplus
}
#endif

/* ---- layout of the structs this header adds -------------------------------------------- */
/* A binding in another language (ffi/src/lib.rs, the ctypes views in ray-tracer_amd/_lib.py and oracle/binding.py)
 * re-declares these structs by hand. The sizes below are what such a binding asserts on its own side
 * (Rust: `const _: () = assert!(size_of::<RtParams>() == 68);`); they are checked here by the C or C++ compiler
 * that reads this header, so header and table cannot drift apart, and tests/test_abi_and_host.py compares the
 * ctypes views with them. All fields are 4- or 8-byte scalars in declaration order, no packing pragma. */
#define RT_SIZEOF_PARAMS 68
#define RT_SIZEOF_SHARD 24
#define RT_SIZEOF_HIT 8
#define RT_SIZEOF_STATS 88
#define RT_SIZEOF_RENDER_OPTS 32
#define RT_SIZEOF_TEXTURE 16
#define RT_SIZEOF_BVH_STATS 32
#if defined(__cplusplus)
static_assert(sizeof(rt_params) == RT_SIZEOF_PARAMS && sizeof(rt_shard) == RT_SIZEOF_SHARD && sizeof(rt_hit) == RT_SIZEOF_HIT &&
                  sizeof(rt_stats) == RT_SIZEOF_STATS && sizeof(rt_render_opts) == RT_SIZEOF_RENDER_OPTS &&
                  sizeof(rt_texture) == RT_SIZEOF_TEXTURE && sizeof(rt_bvh_stats) == RT_SIZEOF_BVH_STATS,
              "struct layout differs from the table other languages' bindings assert");
#elif defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L
_Static_assert(sizeof(rt_params) == RT_SIZEOF_PARAMS && sizeof(rt_shard) == RT_SIZEOF_SHARD && sizeof(rt_hit) == RT_SIZEOF_HIT &&
                   sizeof(rt_stats) == RT_SIZEOF_STATS && sizeof(rt_render_opts) == RT_SIZEOF_RENDER_OPTS &&
                   sizeof(rt_texture) == RT_SIZEOF_TEXTURE && sizeof(rt_bvh_stats) == RT_SIZEOF_BVH_STATS,
               "struct layout differs from the table other languages' bindings assert");
#endif

#endif /* RT_MI355_H */
