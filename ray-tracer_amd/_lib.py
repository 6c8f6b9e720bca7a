"""ctypes view of librt_mi355.so - the declarations of include/rt_mi355.h, one to one.

There is no fallback: if the shared library has not been built (ray-tracer_amd/csrc/Makefile,
or __graft_entry__.build()) importing this module raises, and every computing entry point of
the library itself fails with RT_ERR_NO_DEVICE when no gfx950 GPU is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_LIB_PATH") or os.path.join(_HERE, "librt_mi355.so")  # override: development builds only

f32 = np.float32
u32 = np.uint32

# numpy views of the wire structs (include/rt_types.h; reference src/types.rs, src/grid.rs)
TRIANGLE_DTYPE = np.dtype([("v0", f32, 3), ("material_id", f32), ("v1", f32, 3), ("_pad1", f32), ("v2", f32, 3),
                           ("_pad2", f32), ("uv0", f32, 2), ("uv1", f32, 2), ("uv2", f32, 2), ("_pad3", f32, 2)])
BOX_DTYPE = np.dtype([("min", f32, 3), ("is_moving", f32), ("max", f32, 3), ("_pad2", f32), ("color", f32, 3),
                      ("reflectivity", f32), ("center0", f32, 3), ("_pad4", f32), ("center1", f32, 3), ("_pad5", f32),
                      ("half_size", f32, 3), ("_pad6", f32)])
HIT_DTYPE = np.dtype([("id", u32), ("t", f32)])
MATERIAL_DTYPE = np.dtype([("base_color", f32, 4), ("emissive", f32, 3), ("texture_index", np.int32), ("metallic", f32),
                           ("roughness", f32), ("normal_texture_index", np.int32), ("emissive_texture_index", np.int32),
                           ("alpha_mode", u32), ("alpha_cutoff", f32), ("_pad", f32, 2)])
assert TRIANGLE_DTYPE.itemsize == 80 and BOX_DTYPE.itemsize == 96 and HIT_DTYPE.itemsize == 8
assert MATERIAL_DTYPE.itemsize == 64

RT_MISS = 0xFFFFFFFF
RT_OWNER_SKY = 0xFFFFFFFF  # rt_shard.owner entry of a tile nobody traces (rt_tile_flags, rt_assemble_frames_sky)
RT_OK, RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_STATE, RT_ERR_LIMIT, RT_ERR_COMM = 0, -1, -2, -3, -4, -5, -6
RT_COMM_ID_BYTES = 128
RT_GRID_HIERARCHICAL, RT_GRID_TRIANGLES = 0, 1
RT_NOALLOC_MAX_BOXES = 8192


class CameraUniform(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("_pad1", C.c_float), ("forward", C.c_float * 3), ("_pad2", C.c_float),
                ("right", C.c_float * 3), ("_pad3", C.c_float), ("up", C.c_float * 3), ("time", C.c_float),
                ("lod_factor", C.c_float), ("min_pixel_size", C.c_float), ("show_grid", C.c_float),
                ("_pad4", C.c_float)]


class GridMetadata(C.Structure):
    _fields_ = [("bounds_min", C.c_float * 3), ("num_levels", C.c_uint32), ("bounds_max", C.c_float * 3),
                ("finest_cell_size", C.c_float), ("grid_sizes", (C.c_uint32 * 4) * 4)]

    @property
    def fine_dims(self):
        return tuple(self.grid_sizes[3][k] for k in range(3))


class RtParams(C.Structure):
    _fields_ = [("finest_cell_size", C.c_float), ("max_objects_per_cell", C.c_uint32), ("max_dda_steps", C.c_uint32),
                ("wgsl_compat", C.c_uint32), ("lod_cull", C.c_uint32), ("moving_box_count", C.c_uint32),
                ("fov", C.c_float), ("light_direction", C.c_float * 3), ("shadow_offset", C.c_float),
                ("collect_stats", C.c_uint32), ("skip_duplicate_triangles", C.c_uint32), ("heavy_block_threshold", C.c_uint32), ("cull_empty_blocks", C.c_uint32),
                ("repack_candidates", C.c_uint32), ("persistent_waves", C.c_uint32)]


class RtShard(C.Structure):
    _fields_ = [("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("first", C.c_uint32), ("stride", C.c_uint32),
                ("owner", C.POINTER(C.c_uint32))]

    def __init__(self, tile_w=0, tile_h=0, first=0, stride=1, owner=None):
        """owner: which rank traces each tile of a round (sequence of ints; its length becomes `stride`)."""
        super().__init__(tile_w, tile_h, first, stride)
        self._owner = None
        if owner is not None:
            self._owner = (C.c_uint32 * len(owner))(*[int(k) for k in owner])  # kept alive with the struct
            self.owner = C.cast(self._owner, C.POINTER(C.c_uint32))
            self.stride = len(owner)

    def owner_list(self):
        return list(self._owner) if self._owner is not None else None


class RtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("primary_hits", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("shadow_hits", C.c_uint64), ("cells_visited", C.c_uint64), ("tri_tests", C.c_uint64),
                ("box_tests", C.c_uint64), ("cell_records_fetched", C.c_uint64), ("ref_records_fetched", C.c_uint64),
                ("ref_records_staged", C.c_uint64), ("cell_records_uniform", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class RtBvhStats(C.Structure):
    _fields_ = [("node_count", C.c_uint64), ("leaf_count", C.c_uint64), ("max_depth", C.c_uint64),
                ("total_primitives", C.c_uint64)]


class RtTexture(C.Structure):
    """rt_texture: TextureData of the reference's loader (src/loaders/gltf_triangles.rs:15-19)."""
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class RtRenderOpts(C.Structure):
    _fields_ = [("max_bounces", C.c_uint32), ("shadows", C.c_uint32), ("debug_enabled", C.c_uint32),
                ("debug_pixel", C.c_uint32 * 2), ("pixel_format", C.c_uint32), ("frame_stride", C.c_uint32), ("plain_bounces", C.c_uint32)]


class RayDebugInfo(C.Structure):
    _fields_ = [("ray_origin", C.c_float * 3), ("hit", C.c_float), ("ray_direction", C.c_float * 3),
                ("distance", C.c_float), ("hit_position", C.c_float * 3), ("object_id", C.c_float),
                ("hit_normal", C.c_float * 3), ("num_steps", C.c_float), ("hit_color", C.c_float * 3),
                ("_pad", C.c_float)]


# name -> (restype, argtypes): exactly the functions include/rt_mi355.h declares
_VP, _FP, _U32, _I = C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_int
PROTOTYPES = {
    "rt_default_params": (None, [C.POINTER(RtParams)]),
    "rt_create": (_I, [C.POINTER(_VP), _I, C.POINTER(RtParams)]),
    "rt_destroy": (None, [_VP]),
    "rt_last_error": (C.c_char_p, [_VP]),
    "rt_create_error": (C.c_char_p, []),
    "rt_set_stream": (_I, [_VP, _VP]),
    "rt_sync": (_I, [_VP]),
    "rt_camera_to_uniform": (None, [_FP, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _I,
                                    C.POINTER(CameraUniform)]),
    "rt_set_scene": (_I, [_VP, _VP, _U32, _VP, _U32]),
    "rt_get_scene": (_I, [_VP, C.POINTER(_U32), C.POINTER(_U32)]),
    "rt_set_grid_dense": (_I, [_VP, C.POINTER(GridMetadata), _VP, C.c_size_t]),
    "rt_set_grid_dense_tri": (_I, [_VP, C.POINTER(GridMetadata), _VP, C.c_size_t]),
    "rt_set_grid_csr": (_I, [_VP, C.POINTER(GridMetadata), _VP, _VP]),
    "rt_build_grid": (_I, [_VP, C.c_float, _U32, _I]),
    "rt_get_grid": (_I, [_VP, C.POINTER(GridMetadata), C.POINTER(_U32), C.POINTER(_U32)]),
    "rt_read_grid_csr": (_I, [_VP, _VP, _VP]),
    "rt_update_vertices": (_I, [_VP, _VP, _U32, _U32]),
    "rt_trace_primary": (_I, [_VP, C.POINTER(CameraUniform), _U32, _U32, C.POINTER(RtShard), _VP, _VP]),
    "rt_trace_primary_noalloc": (_I, [_VP, C.POINTER(CameraUniform), _U32, _U32, C.POINTER(RtShard), _VP, _VP]),
    "rt_trace_rays": (_I, [_VP, C.POINTER(CameraUniform), _VP, _VP, _U32, _VP, _I]),
    "rt_assemble_tiles": (_I, [_VP, _VP, _U32, _U32, _U32, _U32, _U32, C.c_size_t, _VP]),
    "rt_comm_unique_id": (_I, [_VP]),
    "rt_comm_init": (_I, [_VP, _VP, _I, _I]),
    "rt_comm_info": (_I, [_VP, C.POINTER(_I), C.POINTER(_I)]),
    "rt_comm_destroy": (_I, [_VP]),
    "rt_broadcast_scene": (_I, [_VP, _I, _VP, _U32, _VP, _U32]),
    "rt_gather_frames": (_I, [_VP, _I, _VP, C.c_size_t, _VP]),
    "rt_build_bvh": (_I, [_VP]),
    "rt_build_bvh_host": (_I, [_VP]),
    "rt_read_bvh": (_I, [_VP, _VP, _VP, _VP]),
    "rt_get_bvh_stats": (_I, [_VP, C.POINTER(RtBvhStats)]),
    "rt_trace_rays_bvh": (_I, [_VP, _VP, _VP, _U32, _VP]),
    "rt_set_skin": (_I, [_VP, _VP, _VP, _VP, _U32, _VP, _U32]),
    "rt_skin_vertices": (_I, [_VP, _VP, _U32]),
    "rt_read_vertices": (_I, [_VP, _VP]),
    "rt_assemble_tiles_rgba8": (_I, [_VP, _VP, _U32, _U32, _U32, _U32, _U32, C.c_size_t, _VP]),
    "rt_assemble_frames_rgba8": (_I, [_VP, _VP, _U32, _U32, _U32, _U32, _U32, _U32, C.c_size_t, C.c_size_t, _VP]),
    "rt_assemble_frames_rgb8": (_I, [_VP, _VP, _U32, _U32, _U32, _U32, _U32, _U32, C.c_size_t, C.c_size_t, _VP]),
    "rt_assemble_frames": (_I, [_VP, _VP, _U32, _U32, _U32, _U32, _U32, _U32, _VP, _U32, _U32, C.c_size_t, C.c_size_t, _VP]),
    "rt_assemble_frames_sky": (_I, [_VP, _VP, _U32, _U32, _U32, _U32, _U32, _U32, _VP, _U32, _VP, _U32, C.c_size_t, C.c_size_t, _VP]),
    "rt_tile_flags": (_I, [_VP, C.POINTER(CameraUniform), _U32, _U32, _U32, _U32, _VP]),
    "rt_render_frames": (_I, [_VP, _VP, _U32, _U32, _U32, C.POINTER(RtShard), C.POINTER(RtRenderOpts), _VP]),
    "rt_default_render_opts": (None, [C.POINTER(RtRenderOpts)]),
    "rt_set_materials": (_I, [_VP, _VP, _U32]),
    "rt_get_heavy_blocks": (_I, [_VP, C.POINTER(_U32), C.POINTER(_U32), C.POINTER(_U32)]),
    "rt_get_cull_stats": (_I, [_VP, C.POINTER(_U32), C.POINTER(_U32)]),
    "rt_checked_violations": (_I, [_VP, C.POINTER(_U32), C.POINTER(_U32 * 3)]),
    "rt_trace_primary_frames": (_I, [_VP, _VP, _U32, _U32, _U32, C.POINTER(RtShard), _VP, _VP, C.c_size_t]),
    "rt_get_build_path": (_I, [_VP, C.POINTER(_I)]),
    "rt_get_trace_refs": (_I, [_VP, C.POINTER(C.c_uint32)]),
    "rt_set_textures": (_I, [_VP, _VP, _U32]),
    "rt_render": (_I, [_VP, C.POINTER(CameraUniform), _U32, _U32, C.POINTER(RtShard), C.POINTER(RtRenderOpts), _VP]),
    "rt_debug_pixel": (_I, [_VP, C.POINTER(CameraUniform), _U32, _U32, _U32, _U32, C.POINTER(RayDebugInfo)]),
    "rt_get_stats": (_I, [_VP, C.POINTER(RtStats)]),
    "rt_set_collect_stats": (_I, [_VP, _I]),
    "rt_intersect_aabb_pairs": (_I, [_VP, _VP, _VP, _U32, _VP, _VP, _U32, _VP]),
    "rt_intersect_tri_pairs": (_I, [_VP, _VP, _VP, _U32, _VP, _U32, _VP]),
    "rt_intersect_tri_pairs_watertight": (_I, [_VP, _VP, _VP, _U32, _VP, _U32, _VP]),
    "rt_timer_begin": (_I, [_VP]),
    "rt_timer_end": (_I, [_VP, C.POINTER(C.c_float)]),
    "rt_probe_copy": (_I, [_VP, _VP, _VP, C.c_size_t]),
}

_lib = None


def load():
    """dlopen librt_mi355.so and bind every prototype. Raises if the library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C ray-tracer_amd/csrc` "
                "(or __graft_entry__.build()). There is no CPU fallback for the trace core.")
        # One HIP runtime per process: PyTorch ships its own libamdhip64 (same soname as /opt/rocm's). If it is
        # loaded first, this library binds to that copy; loaded the other way round the process would hold two
        # runtimes and the second one finds no device. Plain C/C++/Rust callers simply use /opt/rocm's.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError here = header and library disagree
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib
