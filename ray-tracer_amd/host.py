"""Host side of the trace core, shaped after the reference's own call sites.

  Camera        src/camera.rs:43-122         yaw/pitch camera, per-scene start poses, to_uniform
  TraceCore     include/rt_mi355.h           thin object wrapper over the C ABI (one rt_ctx)
  RayTracer     src/renderer.rs:40-351,800+  "new(scene) builds the grid, render(camera, time) traces"
                src/core/ray_tracing_layer.rs:155,436  (ComputeState::new / ::render)

Nothing here computes an intersection: arrays go to librt_mi355.so, hit records come back.
Outputs may be numpy arrays (host, synchronous) or torch CUDA tensors / raw device pointers
(asynchronous on the context's stream).
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import (BOX_DTYPE, HIT_DTYPE, MATERIAL_DTYPE, RtBvhStats, RtRenderOpts, RT_GRID_HIERARCHICAL, RT_GRID_TRIANGLES, RT_MISS, TRIANGLE_DTYPE,
                   CameraUniform, GridMetadata, RayDebugInfo, RtParams, RtShard, RtStats)

DEFAULT_FOV = 0.785398  # raytracer_unified.wgsl:4, src/renderer.rs DEFAULT_FOV


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


def default_params(**overrides):
    p = RtParams()
    _lib.load().rt_default_params(C.byref(p))
    for k, v in overrides.items():
        if k == "light_direction":
            p.light_direction = (C.c_float * 3)(*v)
        else:
            setattr(p, k, v)
    return p


class Camera:
    """reference: src/camera.rs:43-122 (movement/keyboard handling is not part of the trace path)."""

    # src/camera.rs:59-66
    SCENE_POSES = {
        "composed": ((0.0, 40.0, 40.0), math.pi, -0.7),
        "walls": ((0.0, 5.0, 0.0), 0.0, 0.0),
        "tunnel": ((0.0, 0.0, 20.0), math.pi, 0.0),
        "gltf": ((200.0, 200.0, 300.0), 3.35, -0.28),
        "pyramid": ((0.0, 8.0, 20.0), math.pi, -0.5),
    }
    DEFAULT_POSE = ((0.0, 8.0, 15.0), math.pi, -0.6)

    def __init__(self, position, yaw, pitch):
        self.position = tuple(float(x) for x in position)
        self.yaw, self.pitch = float(yaw), float(pitch)

    @classmethod
    def for_scene(cls, scene_name):
        return cls(*cls.SCENE_POSES.get(scene_name, cls.DEFAULT_POSE))

    def to_uniform(self, time=0.0, screen_height=1080.0, fov=DEFAULT_FOV, show_grid=False):
        out = CameraUniform()
        _lib.load().rt_camera_to_uniform((C.c_float * 3)(*self.position), self.yaw, self.pitch, time, screen_height, fov,
                                         int(show_grid), C.byref(out))
        return out

    def forward(self):
        return tuple(self.to_uniform().forward)

    def right(self):
        return tuple(self.to_uniform().right)

    def up(self):
        return (0.0, 1.0, 0.0)


def _ref(x, ctype):
    """byref(x) for a ctypes struct of type `ctype`, or of any other struct type with the same layout size
    (callers may hold their own declaration of the wire structs)."""
    if x is None:
        return None
    if isinstance(x, ctype):
        return C.byref(x)
    if isinstance(x, C.Structure) and C.sizeof(x) == C.sizeof(ctype):
        return C.cast(C.byref(x), C.POINTER(ctype))
    raise TypeError(f"expected {ctype.__name__}, got {type(x).__name__}")


def _ptr(a):
    """host numpy array, torch tensor, int device pointer or None -> c_void_p."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        if not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


def shard_pixels(width, height, shard):
    """Number of rt_hit records a trace call with this shard writes."""
    if shard is None:
        return width * height
    tiles_x = -(-width // shard.tile_w)
    tiles_y = -(-height // shard.tile_h)
    n_tiles = tiles_x * tiles_y
    owner = shard.owner_list()
    if owner is not None:
        rem = n_tiles % shard.stride
        n_local = (n_tiles // shard.stride * sum(1 for k in owner if k == shard.first)
                   + sum(1 for k in owner[:rem] if k == shard.first))
    else:
        n_local = -(-(n_tiles - shard.first) // shard.stride) if shard.first < n_tiles else 0
    return n_local * shard.tile_w * shard.tile_h


class TraceCore:
    """One rt_ctx. Methods are the C ABI's entry points with numpy/torch conveniences."""

    def __init__(self, device_id=0, params=None):
        self._L = _lib.load()
        self._h = C.c_void_p()
        self.params = params if params is not None else default_params()
        rc = self._L.rt_create(C.byref(self._h), device_id, C.byref(self.params))
        if rc != 0:
            raise RtError(rc, self._L.rt_create_error().decode())
        self.device_id = device_id
        self.comm_rank, self.comm_size = 0, 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RtError(rc, self._L.rt_last_error(self._h).decode())

    # ---- streams / timing
    def set_stream(self, stream_ptr):
        self._ck(self._L.rt_set_stream(self._h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def sync(self):
        self._ck(self._L.rt_sync(self._h))

    def timer_begin(self):
        self._ck(self._L.rt_timer_begin(self._h))

    def timer_end(self):
        ms = C.c_float()
        self._ck(self._L.rt_timer_end(self._h, C.byref(ms)))
        return ms.value

    def probe_copy(self, src_ptr, dst_ptr, nbytes):
        """Queue a plain device-to-device copy on the context's stream (a bandwidth probe; time it with the timer)."""
        self._ck(self._L.rt_probe_copy(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), nbytes))

    # ---- scene / grid
    def set_scene(self, boxes=None, tris=None):
        boxes = np.ascontiguousarray(boxes if boxes is not None else np.zeros(0, BOX_DTYPE), BOX_DTYPE)
        tris = np.ascontiguousarray(tris if tris is not None else np.zeros(0, TRIANGLE_DTYPE), TRIANGLE_DTYPE)
        self._ck(self._L.rt_set_scene(self._h, _ptr(boxes), len(boxes), _ptr(tris), len(tris)))
        self.nb, self.nt = len(boxes), len(tris)

    def set_grid_csr(self, meta, cell_start, indices):
        cell_start = np.ascontiguousarray(cell_start, np.uint32)
        indices = np.ascontiguousarray(indices, np.uint32)
        self._ck(self._L.rt_set_grid_csr(self._h, _ref(meta, GridMetadata), _ptr(cell_start), _ptr(indices)))

    def set_grid_dense(self, meta, dense_bytes, tri_cells=False):
        dense_bytes = np.ascontiguousarray(dense_bytes, np.uint8)
        rec = 1040 if tri_cells else 32784
        if dense_bytes.size % rec:
            raise ValueError("dense grid bytes are not a whole number of cell records")
        fn = self._L.rt_set_grid_dense_tri if tri_cells else self._L.rt_set_grid_dense
        self._ck(fn(self._h, _ref(meta, GridMetadata), _ptr(dense_bytes), dense_bytes.size // rec))

    def build_grid(self, cell_size=0.0, max_per_cell=0, mode=RT_GRID_HIERARCHICAL):
        self._ck(self._L.rt_build_grid(self._h, cell_size, max_per_cell, mode))
        return self.grid_info()

    def grid_info(self):
        meta, nc, nr = GridMetadata(), C.c_uint32(), C.c_uint32()
        self._ck(self._L.rt_get_grid(self._h, C.byref(meta), C.byref(nc), C.byref(nr)))
        return meta, nc.value, nr.value

    def heavy_blocks(self):
        """(heavy, early): 8x8 pixel blocks the last planned trace_primary launch marked for the next one - to be traced
        four lanes per ray / to be started ahead of picture order."""
        return self.heavy_plan()[:2]

    def heavy_plan(self):
        """(heavy, early, threshold_us): heavy_blocks() and the run time the next launch will call heavy."""
        n, e, t = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self._L.rt_get_heavy_blocks(self._h, C.byref(n), C.byref(e), C.byref(t)))
        return n.value, e.value, t.value

    def cull_stats(self):
        """(blocks, kept): 8x8 pixel blocks of the last trace_primary frame and how many its block mask kept for tracing
        (rt_params.cull_empty_blocks); (0, 0) when that launch had no mask."""
        n, k = C.c_uint32(), C.c_uint32()
        self._ck(self._L.rt_get_cull_stats(self._h, C.byref(n), C.byref(k)))
        return n.value, k.value

    def checked_violations(self):
        """(count, (kind, a, b)) from a checked build of the library (RT_LIB_PATH=.../librt_mi355_checked.so): index
        violations its trace kernels have counted in this process; None from the shipped library, which has no checks."""
        n, first = C.c_uint32(), (C.c_uint32 * 3)()
        rc = self._L.rt_checked_violations(self._h, C.byref(n), C.byref(first))
        if rc == -4:  # RT_ERR_STATE: not a checked build
            return None
        self._ck(rc)
        return n.value, tuple(first)

    def build_path(self):
        """0: general build, 1: rebuild with every size kept on the device, -1: grid supplied by the caller."""
        v = C.c_int()
        self._ck(self._L.rt_get_build_path(self._h, C.byref(v)))
        return v.value

    def read_grid_csr(self):
        _, nc, nr = self.grid_info()
        cell_start, indices = np.empty(nc + 1, np.uint32), np.empty(nr, np.uint32)
        self._ck(self._L.rt_read_grid_csr(self._h, _ptr(cell_start), _ptr(indices)))
        return cell_start, indices

    def update_vertices(self, xyz9, first=0, count=None):
        if isinstance(xyz9, np.ndarray):
            xyz9 = np.ascontiguousarray(xyz9, np.float32)
            n = xyz9.size // 9
        else:
            n = xyz9.numel() // 9
        self._ck(self._L.rt_update_vertices(self._h, _ptr(xyz9), first, n if count is None else count))

    # ---- several GPUs (RCCL through the C ABI)
    @staticmethod
    def comm_unique_id():
        """RT_COMM_ID_BYTES bytes made on one rank; hand them to the others, then every rank calls comm_init."""
        buf = (C.c_uint8 * _lib.RT_COMM_ID_BYTES)()
        rc = _lib.load().rt_comm_unique_id(buf)
        if rc != 0:
            raise RtError(rc, "rt_comm_unique_id failed (RCCL missing?)")
        return bytes(buf)

    def comm_init(self, comm_id, rank, n_ranks):
        buf = (C.c_uint8 * _lib.RT_COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        self._ck(self._L.rt_comm_init(self._h, buf, rank, n_ranks))
        self.comm_rank, self.comm_size = rank, n_ranks

    def comm_destroy(self):
        self._ck(self._L.rt_comm_destroy(self._h))
        self.comm_size = 0

    def broadcast_scene(self, root, boxes=None, tris=None):
        """Collective: rank `root`'s arrays become every rank's scene (as set_scene would). Returns (nb, nt)."""
        boxes = np.ascontiguousarray(boxes if boxes is not None else np.zeros(0, BOX_DTYPE), BOX_DTYPE)
        tris = np.ascontiguousarray(tris if tris is not None else np.zeros(0, TRIANGLE_DTYPE), TRIANGLE_DTYPE)
        self._ck(self._L.rt_broadcast_scene(self._h, root, _ptr(boxes), len(boxes), _ptr(tris), len(tris)))
        nb, nt = C.c_uint32(), C.c_uint32()
        self._ck(self._L.rt_get_scene(self._h, C.byref(nb), C.byref(nt)))
        self.nb, self.nt = nb.value, nt.value
        return self.nb, self.nt

    def gather_frames(self, root, local, n_bytes, gathered=None):
        """Collective, asynchronous on the context's stream: every rank's `n_bytes` device bytes land in
        gathered[r * n_bytes ...] on rank `root`."""
        self._ck(self._L.rt_gather_frames(self._h, root, _ptr(local), n_bytes, _ptr(gathered)))

    # ---- tracing
    # ---- bounding volume hierarchy
    def build_bvh(self, on_host=False):
        """rt_build_bvh (on the GPU), or rt_build_bvh_host: the same tree by the reference's recursion on one host
        thread. Returns BVHStats as a dict."""
        self._ck((self._L.rt_build_bvh_host if on_host else self._L.rt_build_bvh)(self._h))
        st = RtBvhStats()
        self._ck(self._L.rt_get_bvh_stats(self._h, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in st._fields_}

    def read_bvh(self):
        """rt_read_bvh: (nodes float32 (n_nodes, 8), leaf_first uint32 (n_nodes,), prims uint32 (n_prims,))."""
        st = RtBvhStats()
        self._ck(self._L.rt_get_bvh_stats(self._h, C.byref(st)))
        nodes = np.empty((int(st.node_count), 8), np.float32)
        leaf_first = np.empty(int(st.node_count), np.uint32)
        prims = np.empty(int(st.total_primitives), np.uint32)
        self._ck(self._L.rt_read_bvh(self._h, _ptr(nodes), _ptr(leaf_first), _ptr(prims)))
        return nodes, leaf_first, prims

    def trace_rays_bvh(self, origins, dirs, out=None, n=None):
        """Closest t per ray (-1 = none). numpy in -> numpy out; torch CUDA tensors throughout with `out`."""
        if out is None:
            origins = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
            dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
            out = np.empty(len(origins), np.float32)
            n = len(origins)
        elif n is None:
            n = origins.shape[0]
        self._ck(self._L.rt_trace_rays_bvh(self._h, _ptr(origins), _ptr(dirs), n, _ptr(out)))
        return out

    # ---- skinning
    def set_skin(self, positions, joints, weights, tri_indices):
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        joints = np.ascontiguousarray(joints, np.uint16).reshape(-1, 4)
        weights = np.ascontiguousarray(weights, np.float32).reshape(-1, 4)
        tri_indices = np.ascontiguousarray(tri_indices, np.uint32).reshape(-1, 3)
        if not (len(positions) == len(joints) == len(weights)):
            raise ValueError("positions, joints and weights must describe the same vertices")
        self._ck(self._L.rt_set_skin(self._h, _ptr(positions), _ptr(joints), _ptr(weights), len(positions),
                                     _ptr(tri_indices), len(tri_indices)))

    def skin_vertices(self, joint_matrices):
        """(n_joints, 16) float32 column-major matrices: numpy array or a torch CUDA tensor."""
        if isinstance(joint_matrices, np.ndarray):
            joint_matrices = np.ascontiguousarray(joint_matrices, np.float32).reshape(-1, 16)
        self._ck(self._L.rt_skin_vertices(self._h, _ptr(joint_matrices), joint_matrices.shape[0]))

    def read_vertices(self):
        out = np.empty((self.nt, 9), np.float32)
        self._ck(self._L.rt_read_vertices(self._h, _ptr(out)))
        return out

    def trace_primary(self, cam, width, height, shard=None, shadow=True, out_primary=None, out_shadow=None):
        """Returns (primary, shadow|None). Pass torch CUDA tensors (n,2 int32/8 bytes per record) or
        device pointers as out_* to keep results on the GPU; default is new numpy HIT_DTYPE arrays."""
        n = shard_pixels(width, height, shard)
        if out_primary is None:
            out_primary = np.empty(n, HIT_DTYPE)
        if shadow and out_shadow is None:
            out_shadow = np.empty(n, HIT_DTYPE)
        self._ck(self._L.rt_trace_primary(self._h, _ref(cam, CameraUniform), width, height, _ref(shard, RtShard),
                                          _ptr(out_primary),
                                          _ptr(out_shadow) if shadow else None))
        return out_primary, (out_shadow if shadow else None)

    def trace_primary_frames(self, cams, width, height, shard=None, shadow=True, out_primary=None, out_shadow=None, frame_stride=0):
        """rt_trace_primary_frames: up to 16 consecutive frames in one call. Returns (primary, shadow|None), numpy
        HIT_DTYPE arrays of shape (n_frames, n_records) unless torch CUDA tensors / device pointers are passed."""
        n = frame_stride or shard_pixels(width, height, shard)
        arr = (CameraUniform * len(cams))()
        for j, cam in enumerate(cams):
            C.memmove(C.byref(arr, j * C.sizeof(CameraUniform)), C.byref(cam), C.sizeof(CameraUniform))
        if out_primary is None:
            out_primary = np.empty((len(cams), n), HIT_DTYPE)
        if shadow and out_shadow is None:
            out_shadow = np.empty((len(cams), n), HIT_DTYPE)
        self._ck(self._L.rt_trace_primary_frames(self._h, arr, len(cams), width, height, _ref(shard, RtShard), _ptr(out_primary),
                                                 _ptr(out_shadow) if shadow else None, frame_stride))
        return out_primary, (out_shadow if shadow else None)

    def trace_primary_noalloc(self, cam, width, height, out_primary, out_shadow=None, shard=None):
        """rt_trace_primary_noalloc: the same trace without any device allocation; outputs must be device memory
        (torch CUDA tensors or raw device pointers)."""
        self._ck(self._L.rt_trace_primary_noalloc(self._h, _ref(cam, CameraUniform), width, height, _ref(shard, RtShard),
                                                  _ptr(out_primary), _ptr(out_shadow)))
        return out_primary, out_shadow

    def trace_rays(self, origins, dirs, cam=None, any_hit=False, out=None, n=None):
        if isinstance(origins, np.ndarray):
            origins = np.ascontiguousarray(origins, np.float32)
            dirs = np.ascontiguousarray(dirs, np.float32)
            n = origins.size // 3
        elif n is None:
            n = origins.numel() // 3
        if out is None:
            out = np.empty(n, HIT_DTYPE)
        self._ck(self._L.rt_trace_rays(self._h, _ref(cam, CameraUniform), _ptr(origins), _ptr(dirs), n,
                                       _ptr(out), int(any_hit)))
        return out

    def assemble_tiles(self, gathered, width, height, tile_w, tile_h, n_ranks, image, rank_stride=0):
        self._ck(self._L.rt_assemble_tiles(self._h, _ptr(gathered), width, height, tile_w, tile_h, n_ranks, rank_stride,
                                           _ptr(image)))

    def assemble_tiles_rgba8(self, gathered, width, height, tile_w, tile_h, n_ranks, image, rank_stride=0):
        self._ck(self._L.rt_assemble_tiles_rgba8(self._h, _ptr(gathered), width, height, tile_w, tile_h, n_ranks,
                                                 rank_stride, _ptr(image)))

    # ---- shaded picture
    def trace_refs(self):
        """Reference records the trace kernels walk (cell references minus skipped duplicate triangles)."""
        n = C.c_uint32()
        self._ck(self._L.rt_get_trace_refs(self._h, C.byref(n)))
        return n.value

    def set_materials(self, materials):
        m = np.ascontiguousarray(materials if materials is not None else np.zeros(0, MATERIAL_DTYPE), MATERIAL_DTYPE)
        self._ck(self._L.rt_set_materials(self._h, _ptr(m), len(m)))

    def set_textures(self, textures):
        """textures: sequence of (height, width, 4) uint8 arrays (TextureData of the loader, RGBA8 rows)."""
        keep = [np.ascontiguousarray(t, np.uint8) for t in (textures or [])]
        for t in keep:
            if t.ndim != 3 or t.shape[2] != 4:
                raise ValueError("a texture is a (height, width, 4) uint8 array")
        arr = (_lib.RtTexture * max(1, len(keep)))()
        for i, t in enumerate(keep):
            arr[i].rgba8, arr[i].width, arr[i].height = t.ctypes.data, t.shape[1], t.shape[0]
        self._ck(self._L.rt_set_textures(self._h, arr if keep else None, len(keep)))

    def render(self, cam, width, height, shard=None, max_bounces=8, shadows=False, debug_pixel=None, out=None, plain_bounces=False):
        """The shader's picture: (n_pixels, 4) uint8 RGBA, row-major (tile-packed with a shard). `out` may be a
        torch CUDA uint8 tensor / device pointer to keep it on the GPU."""
        n = shard_pixels(width, height, shard)
        if out is None:
            out = np.empty((n, 4), np.uint8)
        o = RtRenderOpts()
        self._L.rt_default_render_opts(C.byref(o))
        o.max_bounces, o.shadows, o.plain_bounces = max_bounces, 1 if shadows else 0, 1 if plain_bounces else 0
        if debug_pixel is not None:
            o.debug_enabled, o.debug_pixel[0], o.debug_pixel[1] = 1, debug_pixel[0], debug_pixel[1]
        self._ck(self._L.rt_render(self._h, _ref(cam, CameraUniform), width, height, _ref(shard, RtShard), C.byref(o),
                                   _ptr(out)))
        return out

    def render_frames(self, cams, width, height, shard=None, max_bounces=8, shadows=False, out=None, rgb8=False,
                      frame_stride=0):
        """Up to 16 consecutive frames in one launch: (n_frames, n_pixels, 4) uint8, or (n_frames, n_pixels, 3)
        with rgb8 (no alpha byte). `cams`: sequence of CameraUniform. frame_stride: pixels between frames in `out`
        when that is more than one frame writes (0 = packed)."""
        n = frame_stride or shard_pixels(width, height, shard)
        arr = (CameraUniform * len(cams))()
        for j, cam in enumerate(cams):
            C.memmove(C.byref(arr, j * C.sizeof(CameraUniform)), C.byref(cam), C.sizeof(CameraUniform))
        if out is None:
            out = np.empty((len(cams), n, 3 if rgb8 else 4), np.uint8)
        o = RtRenderOpts()
        self._L.rt_default_render_opts(C.byref(o))
        o.max_bounces, o.shadows, o.pixel_format = max_bounces, 1 if shadows else 0, 1 if rgb8 else 0
        o.frame_stride = frame_stride
        self._ck(self._L.rt_render_frames(self._h, arr, len(cams), width, height, _ref(shard, RtShard), C.byref(o), _ptr(out)))
        return out

    def assemble_frames_rgba8(self, gathered, width, height, tile_w, tile_h, n_ranks, n_frames, rank_stride, frame_stride,
                              images):
        self._ck(self._L.rt_assemble_frames_rgba8(self._h, _ptr(gathered), width, height, tile_w, tile_h, n_ranks, n_frames,
                                                  rank_stride, frame_stride, _ptr(images)))

    def assemble_frames_rgb8(self, gathered, width, height, tile_w, tile_h, n_ranks, n_frames, rank_stride, frame_stride,
                             images):
        self._ck(self._L.rt_assemble_frames_rgb8(self._h, _ptr(gathered), width, height, tile_w, tile_h, n_ranks, n_frames,
                                                 rank_stride, frame_stride, _ptr(images)))

    def assemble_frames(self, gathered, rgb8, width, height, tile_w, tile_h, n_ranks, owner, n_frames, rank_stride,
                        frame_stride, images):
        """owner: the table the ranks rendered with (RtShard(owner=...)), None = one tile each per round."""
        tab = (C.c_uint32 * len(owner))(*[int(k) for k in owner]) if owner is not None else None
        self._ck(self._L.rt_assemble_frames(self._h, _ptr(gathered), 1 if rgb8 else 0, width, height, tile_w, tile_h, n_ranks,
                                            tab, len(owner) if owner is not None else 0, n_frames, rank_stride, frame_stride,
                                            _ptr(images)))

    def tile_flags(self, cam, width, height, tile_w, tile_h):
        """rt_tile_flags: uint8 per tile (row-major), 0 = the tile shows nothing but sky from this camera. Synchronous."""
        n = (-(-width // tile_w)) * (-(-height // tile_h))
        flags = np.empty(n, np.uint8)
        self._ck(self._L.rt_tile_flags(self._h, _ref(cam, CameraUniform), width, height, tile_w, tile_h, _ptr(flags)))
        return flags

    def assemble_frames_sky(self, gathered, rgb8, width, height, tile_w, tile_h, n_ranks, owner, cams, rank_stride, frame_stride,
                            images):
        """rt_assemble_frames_sky: as assemble_frames, tiles whose owner entry is RT_OWNER_SKY filled from cams[j]."""
        # owner / cams may be ctypes arrays made once by the caller (a table over every tile of a frame has thousands of entries)
        tab = owner if isinstance(owner, C.Array) else (C.c_uint32 * len(owner))(*[int(k) for k in owner])
        arr = cams
        if not isinstance(cams, C.Array):
            arr = (CameraUniform * len(cams))()
            for j, cam in enumerate(cams):
                C.memmove(C.byref(arr, j * C.sizeof(CameraUniform)), C.byref(cam), C.sizeof(CameraUniform))
        self._ck(self._L.rt_assemble_frames_sky(self._h, _ptr(gathered), 1 if rgb8 else 0, width, height, tile_w, tile_h, n_ranks,
                                                tab, len(tab), arr, len(arr), rank_stride, frame_stride, _ptr(images)))

    def debug_pixel(self, cam, width, height, px, py):
        info = RayDebugInfo()
        self._ck(self._L.rt_debug_pixel(self._h, _ref(cam, CameraUniform), width, height, px, py, C.byref(info)))
        return info

    def set_collect_stats(self, enabled):
        self._ck(self._L.rt_set_collect_stats(self._h, int(enabled)))

    def stats(self):
        s = RtStats()
        self._ck(self._L.rt_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    # ---- primitive routines
    def intersect_aabb_pairs(self, origins, dirs, box_min, box_max, out=None):
        """numpy in -> numpy out; with `out` a device tensor, all arguments are float32 device tensors (n,3)."""
        if out is not None:
            self._ck(self._L.rt_intersect_aabb_pairs(self._h, _ptr(origins), _ptr(dirs), origins.shape[0], _ptr(box_min),
                                                     _ptr(box_max), box_min.shape[0], _ptr(out)))
            return out
        origins, dirs, box_min, box_max = (np.ascontiguousarray(a, np.float32).reshape(-1, 3)
                                           for a in (origins, dirs, box_min, box_max))
        out = np.empty((len(origins), len(box_min)), np.float32)
        self._ck(self._L.rt_intersect_aabb_pairs(self._h, _ptr(origins), _ptr(dirs), len(origins), _ptr(box_min),
                                                 _ptr(box_max), len(box_min), _ptr(out)))
        return out

    def intersect_tri_pairs(self, origins, dirs, v012, out=None, watertight=False):
        fn = self._L.rt_intersect_tri_pairs_watertight if watertight else self._L.rt_intersect_tri_pairs
        if out is not None:  # device tensors throughout
            self._ck(fn(self._h, _ptr(origins), _ptr(dirs), origins.shape[0], _ptr(v012),
                                                    v012.shape[0], _ptr(out)))
            return out
        origins, dirs = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (origins, dirs))
        v012 = np.ascontiguousarray(v012, np.float32).reshape(-1, 9)
        out = np.empty((len(origins), len(v012)), np.float32)
        self._ck(fn(self._h, _ptr(origins), _ptr(dirs), len(origins), _ptr(v012), len(v012), _ptr(out)))
        return out


_SHARED_STREAMS = {}


def shared_stream(device, k):
    """The k-th of a small set of torch streams kept per device for the life of the process. HIP deals a process's
    streams out to a few hardware queues (four by default) as they are first used, and a stream keeps its queue; two
    busy streams on one queue run one after the other. A program that makes fresh streams for every phase - torch
    hands out another of its 32 pooled streams each time - ends up with contexts whose streams share a queue. Frames in
    flight, groups in flight and benchmark legs all take their streams from here: slot k is the same stream every time."""
    import torch
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    pool = _SHARED_STREAMS.setdefault(key, [])
    while len(pool) <= k:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[k]


def triangles_from_xyz9(xyz9, material_id=0):
    """(n,9) float32 vertex rows -> TriangleData records (src/types.rs:211-233 with zero uvs)."""
    xyz9 = np.ascontiguousarray(xyz9, np.float32).reshape(-1, 9)
    t = np.zeros(len(xyz9), TRIANGLE_DTYPE)
    t["v0"], t["v1"], t["v2"] = xyz9[:, 0:3], xyz9[:, 3:6], xyz9[:, 6:9]
    t["material_id"] = material_id
    return t


class RayTracer:
    """The renderer-shaped wrapper: construct with a scene (grid is built on the GPU, as
    RayTracer::new calls HierarchicalGrid::build, src/renderer.rs:97-99), then render() per frame
    (src/renderer.rs:800-868). Returns hit records instead of an RGBA texture: shading is outside
    the trace core."""

    def __init__(self, boxes=None, tris=None, width=1920, height=1080, device_id=0, params=None):
        self.width, self.height = width, height
        self.core = TraceCore(device_id, params)
        self.core.set_scene(boxes, tris)
        self.grid_meta, self.n_cells, self.n_refs = self.core.build_grid()

    def render(self, camera, time=0.0, shadow=True, **kw):
        cam = camera.to_uniform(time, float(self.height), self.core.params.fov, False)
        return self.core.trace_primary(cam, self.width, self.height, shadow=shadow, **kw)
