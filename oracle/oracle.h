/*
 * oracle.h - CPU restatement of the reference's trace path. TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product (librt_mi355.so, the ray-tracer_amd package) links, loads or
 * calls this code; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * do, and only as the checker / the timed CPU baseline.
 *
 * The reference is a Rust crate and there is no Rust toolchain in the build image, so
 * this is a restatement in C++ (scalar f32, no FMA contraction, no fast-math), function
 * by function, of:
 *   src/math/ray.rs:3-36, src/math/grid.rs:3-15, src/math/aabb.rs:9-29,
 *   src/core/triangle_intersection.rs:31-97,184-204, src/types.rs:110-115,235-244,
 *   src/grid.rs:9-20,92-100,110-176,218-358, src/grid_triangles.rs:29-261,
 *   src/camera.rs:51-122, src/core/bvh.rs:44-232,
 *   examples/massive_benchmark.rs:226-270,311-342, src/core/benchmark.rs:99-140,
 *   benches/bounding_volumes.rs:39-47,147-179
 * and, because the reference has no CPU grid traversal, of the geometry part of
 *   src/raytracer_unified.wgsl:147-235,238-277,381-409,430-588,625-653, and of the material part of the
 * triangle test (:279-366) against the texture array of src/renderer.rs:158-259.
 * Vector arithmetic follows glam 0.29.3's scalar Vec3 (Cargo.lock:1077-1078; source not
 * vendored in the reference): dot = (x*x' + y*y') + z*z', cross = (y*z' - y'*z, ...),
 * normalize = v * (1/sqrt(dot)), min/max component-wise f32::min/f32::max.
 *
 * Pinning: the reference's own unit tests for these routines are reproduced as
 * known-answer tests (tests/test_oracle_reference_kat.py). Grid binning, DDA traversal,
 * ray generation, LOD cull, moving boxes and tie-breaking have no reference test or
 * golden vector: for those parts parity is UNPINNED by the reference and rests on this
 * restatement plus its self-checks (DDA result == brute force over all triangles).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "../include/rt_mi355.h" /* types only: rt_params, rt_hit, rt_stats, wire structs */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- primitive routines ---------------------------------------------------------- */
float orc_intersect_aabb(const float o[3], const float d[3], const float bmin[3], const float bmax[3],
                         int wgsl_compat);
/* returns 1 on hit; out6 = t,u,v,nx,ny,nz */
int orc_moller_trumbore(const float o[3], const float d[3], const float v0[3], const float v1[3],
                        const float v2[3], int wgsl_compat, float out6[6]);
int orc_watertight(const float o[3], const float d[3], const float v0[3], const float v1[3],
                   const float v2[3], float out6[6]);
/* returns index into tris of the closest hit or -1; out6 as above */
int64_t orc_batch_intersect_triangles(const float o[3], const float d[3], const TriangleData* tris,
                                      const uint32_t* indices, size_t n_indices, float out6[6]);
void orc_world_to_cell(const float pos[3], const float bounds_min[3], float cell_size, int32_t out[3]);
void orc_aabb_union(const float amin[3], const float amax[3], const float bmin[3], const float bmax[3],
                    float omin[3], float omax[3]);
void orc_aabb_center(const float amin[3], const float amax[3], float out[3]);
float orc_aabb_surface_area(const float amin[3], const float amax[3]);
void orc_triangle_bounds(const TriangleData* t, float omin[3], float omax[3]);

/* all-pairs forms (shape of the criterion loops) */
void orc_intersect_aabb_pairs(const float* origins, const float* dirs, uint32_t n_rays, const float* box_min,
                              const float* box_max, uint32_t n_prims, int wgsl_compat, float* out_t);
/* watertight_intersect (src/core/triangle_intersection.rs:101-181) for every (ray, triangle) pair: t or -1. */
void orc_watertight_pairs(const float* origins, const float* dirs, uint32_t n_rays, const float* v012, uint32_t n_prims,
                          float* out_t);
void orc_intersect_tri_pairs(const float* origins, const float* dirs, uint32_t n_rays, const float* v012,
                             uint32_t n_prims, int wgsl_compat, float* out_t);

/* ---- camera ---------------------------------------------------------------------- */
void orc_camera_to_uniform(const float position[3], float yaw, float pitch, float time, float screen_height,
                           float fov, int show_grid, CameraUniform* out);

/* ---- grids ----------------------------------------------------------------------- */
typedef struct orc_grid orc_grid;
/* mode: RT_GRID_HIERARCHICAL or RT_GRID_TRIANGLES */
orc_grid* orc_grid_build(const BoxData* boxes, uint32_t nb, const TriangleData* tris, uint32_t nt, float cell_size,
                         uint32_t max_per_cell, int mode);
void orc_grid_free(orc_grid* g);
void orc_grid_info(const orc_grid* g, GridMetadata* meta, uint64_t* n_cells, uint64_t* n_refs,
                   uint64_t* dropped_refs);
void orc_grid_copy_csr(const orc_grid* g, uint32_t* cell_start, uint32_t* indices);
/* dense exports (to_gpu_buffers); out must hold n_cells records */
void orc_grid_copy_dense(const orc_grid* g, FineCellData* out);
void orc_grid_copy_dense_tri(const orc_grid* g, TriFineCellData* out);
/* coarse level counts, concatenated (u8, saturating); returns total length */
uint64_t orc_grid_coarse_counts(const orc_grid* g, uint8_t* out, uint64_t cap);

/* ---- tracing --------------------------------------------------------------------- */
typedef struct orc_scene {
    const BoxData* boxes;
    uint32_t nb;
    const TriangleData* tris;
    uint32_t nt;
    GridMetadata meta;
    const uint32_t* cell_start; /* n_cells + 1 */
    const uint32_t* indices;
    uint64_t n_cells;
    const MaterialData* mats; /* may be NULL: the "Materials" buffer, for the alpha rules of the triangle test */
    uint32_t n_mats;
    /* The texture array as the renderer builds it (src/renderer.rs:196-248): tex_layers layers of tex_w x tex_h RGBA8
     * texels, each source texture in the top-left corner of its layer, the rest zero. tex_layers == 0: the 1x1 white
     * dummy (:159-193). */
    const uint8_t* texels;
    uint32_t tex_w, tex_h, tex_layers;
} orc_scene;

/* textureSampleLevel(texture_array, sampler, uv, layer, 0.0) with the renderer's sampler (repeat, linear, src/renderer.rs:251-259),
 * filter weights in f32 (see rt_set_textures in rt_mi355.h). Parity unpinned by the reference (no texture test or fixture);
 * pinned by hand-computed known answers in tests/test_oracle_textures.py. */
void orc_sample_texture(const orc_scene* s, float u, float v, int32_t layer, float out_rgba[4]);

/* Primary (+ optional shadow) rays for the pixels selected by shard (NULL = whole frame);
 * outputs packed exactly as rt_trace_primary packs them. shadow_o/shadow_d (optional,
 * 3 floats per packed pixel) receive the generated shadow rays. */
void orc_trace_primary(const orc_scene* s, const rt_params* p, const CameraUniform* cam, uint32_t width,
                       uint32_t height, const rt_shard* shard, rt_hit* out_primary, rt_hit* out_shadow,
                       float* shadow_o, float* shadow_d, rt_stats* stats, int n_threads);
/* Linear-blend skinning of triangle corners (glTF 2.0 rules; the reference does not skin, so this restates the
 * build's own kernel, not a reference routine: parity unpinned by the reference). out_xyz9: n_triangles*9. */
void orc_skin_vertices(const float* positions, const uint16_t* joints, const float* weights, const uint32_t* tri_indices,
                       uint32_t n_triangles, const float* joint_matrices, float* out_xyz9);
/* The shaded picture of the shader's main() (raytracer_unified.wgsl:590-621, 655-720), materials as the triangle
 * test evaluates them (:279-366: base-colour texture, alpha rules, normal map, emissive texture, metallic);
 * 4 bytes per pixel, packed like orc_trace_primary's records. Parity unpinned by the reference (it has no image
 * fixture); pinned against itself by tests/test_oracle_golden.py. */
void orc_render(const orc_scene* s, const rt_params* p, const CameraUniform* cam, uint32_t width, uint32_t height,
                const rt_shard* shard, const MaterialData* mats, uint32_t n_mats, const rt_render_opts* opts,
                uint8_t* out_rgba8, rt_stats* stats, int n_threads);
void orc_primary_rays(const rt_params* p, const CameraUniform* cam, uint32_t width, uint32_t height,
                      float* origins, float* dirs);
void orc_trace_rays(const orc_scene* s, const rt_params* p, const CameraUniform* cam, const float* origins,
                    const float* dirs, uint32_t n, rt_hit* out, int any_hit, rt_stats* stats, int n_threads);
/* closest hit by testing every object in id order (no grid): the self-check for the DDA */
void orc_trace_rays_bruteforce(const orc_scene* s, const rt_params* p, const CameraUniform* cam,
                               const float* origins, const float* dirs, uint32_t n, rt_hit* out, int n_threads);
void orc_debug_pixel(const orc_scene* s, const rt_params* p, const CameraUniform* cam, uint32_t width,
                     uint32_t height, uint32_t px, uint32_t py, RayDebugInfo* info);
uint64_t orc_shard_pixels(uint32_t width, uint32_t height, const rt_shard* shard);

/* ---- reference CPU trace path: binned-SAH BVH + both-children traversal ------------ */
typedef struct orc_bvh orc_bvh;
orc_bvh* orc_bvh_build(const TriangleData* tris, uint32_t nt);
void orc_bvh_free(orc_bvh* b);
void orc_bvh_stats(const orc_bvh* b, uint64_t* nodes, uint64_t* leaves, uint64_t* max_depth, uint64_t* prims);
/* out_t[i] = closest t or -1; returns hit count */
uint64_t orc_bvh_trace(const orc_bvh* b, const TriangleData* tris, const float* origins, const float* dirs,
                       uint32_t n, float* out_t, int n_threads);
uint64_t orc_linear_scan(const TriangleData* tris, uint32_t nt, const float* origins, const float* dirs,
                         uint32_t n, float* out_t, int n_threads);

/* ---- workload generators ----------------------------------------------------------- */
void orc_generate_test_triangles(uint64_t count, TriangleData* out);
void orc_generate_test_rays(uint64_t count, float* origins, float* dirs);
void orc_random_unit_vector(uint32_t seed, float out[3]);
void orc_particle_aabbs(uint32_t count, float half, float* box_min, float* box_max);
void orc_pyramid_triangles(TriangleData out[6]);

#ifdef __cplusplus
}
#endif
#endif
