/*
 * rt_types.h - wire structs shared by the host and the trace core.
 *
 * Every struct here is a byte-for-byte mirror of a #[repr(C)] Pod struct of the
 * reference renderer, so a caller holding the reference's Vec<BoxData>,
 * Vec<TriangleData>, GridMetadata, Vec<FineCellData> or CameraUniform can hand
 * the raw bytes (bytemuck::cast_slice) straight to the C ABI in rt_mi355.h.
 *
 * Plain C, no dependencies; included by the HIP library, the CPU oracle and the
 * host bindings alike.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: src/grid.rs:5-7 */
#define RT_GRID_LEVELS 4u
#define RT_FINEST_CELL_SIZE 16.0f
#define RT_MAX_OBJECTS_PER_CELL 8192u
/* reference: src/grid_triangles.rs:5-7,127-128 */
#define RT_TRI_MAX_TRIANGLES_PER_CELL 256u
#define RT_TRI_MIN_CELL_SIZE 0.1f
#define RT_TRI_MAX_GRID_DIM 1024u

/* reference: src/types.rs:4-19 (CameraUniform), WGSL mirror raytracer_unified.wgsl:7-20 */
typedef struct CameraUniform {
    float position[3];
    float _pad1;
    float forward[3];
    float _pad2;
    float right[3];
    float _pad3;
    float up[3];
    float time;
    float lod_factor;
    float min_pixel_size;
    float show_grid;
    float _pad4;
} CameraUniform;

/* reference: src/types.rs:21-36 (BoxData), WGSL mirror raytracer_unified.wgsl:22-35.
 * min/max is the (possibly swept and padded) AABB used for binning;
 * center0/center1/half_size is the box that is actually intersected. */
typedef struct BoxData {
    float min[3];
    float is_moving;
    float max[3];
    float _pad2;
    float color[3];
    float reflectivity;
    float center0[3];
    float _pad4;
    float center1[3];
    float _pad5;
    float half_size[3];
    float _pad6;
} BoxData;

/* reference: src/types.rs:196-209 (TriangleData), WGSL mirror raytracer_unified.wgsl:37-48 */
typedef struct TriangleData {
    float v0[3];
    float material_id;
    float v1[3];
    float _pad1;
    float v2[3];
    float _pad2;
    float uv0[2];
    float uv1[2];
    float uv2[2];
    float _pad3[2];
} TriangleData;

/* reference: src/types.rs:248-261 (MaterialData) */
typedef struct MaterialData {
    float base_color[4];
    float emissive[3];
    int32_t texture_index;
    float metallic;
    float roughness;
    int32_t normal_texture_index;
    int32_t emissive_texture_index;
    uint32_t alpha_mode;
    float alpha_cutoff;
    float _pad[2];
} MaterialData;

/* reference: src/types.rs:296-302 (SceneConfig) */
typedef struct SceneConfig {
    uint32_t num_boxes;
    uint32_t num_triangles;
    uint32_t _pad[2];
} SceneConfig;

/* reference: src/types.rs:155-168 (RayDebugInfo), written by raytracer_unified.wgsl:700-709 */
typedef struct RayDebugInfo {
    float ray_origin[3];
    float hit;
    float ray_direction[3];
    float distance;
    float hit_position[3];
    float object_id;
    float hit_normal[3];
    float num_steps;
    float hit_color[3];
    float _pad;
} RayDebugInfo;

/* reference: src/types.rs:187-193 (DebugParams) */
typedef struct DebugParams {
    uint32_t debug_pixel[2];
    uint32_t enabled;
    uint32_t _pad;
} DebugParams;

/* reference: src/grid.rs:22-30 (GridMetadata) and src/grid_triangles.rs:264-272
 * (TriangleGridMetadata, same layout). grid_sizes[RT_GRID_LEVELS-1] is the fine level. */
typedef struct GridMetadata {
    float bounds_min[3];
    uint32_t num_levels;
    float bounds_max[3];
    float finest_cell_size;
    uint32_t grid_sizes[RT_GRID_LEVELS][4];
} GridMetadata;

/* reference: src/grid.rs:32-38 (FineCellData, 32784 bytes per cell) */
typedef struct FineCellData {
    uint32_t object_indices[RT_MAX_OBJECTS_PER_CELL];
    uint32_t count;
    uint32_t _pad[3];
} FineCellData;

/* reference: src/grid_triangles.rs:274-280 (its FineCellData, 1040 bytes per cell) */
typedef struct TriFineCellData {
    uint32_t object_indices[RT_TRI_MAX_TRIANGLES_PER_CELL];
    uint32_t count;
    uint32_t _pad[3];
} TriFineCellData;

#ifdef __cplusplus
} /* extern "C" */
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

RT_STATIC_ASSERT(sizeof(CameraUniform) == 80, "CameraUniform must be 80 bytes");
RT_STATIC_ASSERT(sizeof(BoxData) == 96, "BoxData must be 96 bytes");
RT_STATIC_ASSERT(sizeof(TriangleData) == 80, "TriangleData must be 80 bytes");
RT_STATIC_ASSERT(sizeof(MaterialData) == 64, "MaterialData must be 64 bytes");
RT_STATIC_ASSERT(sizeof(SceneConfig) == 16, "SceneConfig must be 16 bytes");
RT_STATIC_ASSERT(sizeof(RayDebugInfo) == 80, "RayDebugInfo must be 80 bytes");
RT_STATIC_ASSERT(sizeof(DebugParams) == 16, "DebugParams must be 16 bytes");
RT_STATIC_ASSERT(sizeof(GridMetadata) == 96, "GridMetadata must be 96 bytes");
RT_STATIC_ASSERT(sizeof(FineCellData) == 32784, "FineCellData must be 32784 bytes");
RT_STATIC_ASSERT(sizeof(TriFineCellData) == 1040, "TriFineCellData must be 1040 bytes");

#endif /* RT_TYPES_H */
