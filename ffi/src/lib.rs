//! `extern "C"` declarations of include/rt_mi355.h plus a small safe wrapper.
//! The renderer's own `#[repr(C)]` Pod structs (src/types.rs, src/grid.rs) are passed as raw bytes;
//! nothing is re-declared here except the handful of structs the ABI adds.
use std::ffi::{c_char, c_int, c_void, CStr};

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, bytemuck::Pod, bytemuck::Zeroable)]
pub struct RtHit { pub id: u32, pub t: f32 }
pub const RT_MISS: u32 = 0xFFFF_FFFF;
/// `RtShard::owner` entry of a tile nobody traces (rt_tile_flags, rt_assemble_frames_sky).
pub const RT_OWNER_SKY: u32 = 0xFFFF_FFFF;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct RtParams {
    pub finest_cell_size: f32, pub max_objects_per_cell: u32, pub max_dda_steps: u32, pub wgsl_compat: u32,
    pub lod_cull: u32, pub moving_box_count: u32, pub fov: f32, pub light_direction: [f32; 3],
    pub shadow_offset: f32, pub collect_stats: u32, pub skip_duplicate_triangles: u32, pub heavy_block_threshold: u32, pub cull_empty_blocks: u32, pub repack_candidates: u32,
    pub persistent_waves: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct RtShard { pub tile_w: u32, pub tile_h: u32, pub first: u32, pub stride: u32, pub owner: *const u32 }

#[repr(C)]
pub struct RtCtx { _private: [u8; 0] }

// The table at the end of include/rt_mi355.h (RT_SIZEOF_*), which the C / C++ compiler checks against the header's own
// structs: a struct that grows there without growing here stops this crate from compiling instead of corrupting memory.
const _: () = assert!(std::mem::size_of::<RtParams>() == 68);
const _: () = assert!(std::mem::size_of::<RtShard>() == 24);
const _: () = assert!(std::mem::size_of::<RtHit>() == 8);

extern "C" {
    pub fn rt_default_params(p: *mut RtParams);
    pub fn rt_create(out: *mut *mut RtCtx, device_id: c_int, params: *const RtParams) -> c_int;
    pub fn rt_destroy(ctx: *mut RtCtx);
    pub fn rt_last_error(ctx: *const RtCtx) -> *const c_char;
    pub fn rt_create_error() -> *const c_char;
    pub fn rt_sync(ctx: *mut RtCtx) -> c_int;
    pub fn rt_set_scene(ctx: *mut RtCtx, boxes: *const c_void, nb: u32, tris: *const c_void, nt: u32) -> c_int;
    pub fn rt_set_grid_dense(ctx: *mut RtCtx, meta: *const c_void, fine_cells: *const c_void, n_cells: usize) -> c_int;
    pub fn rt_set_grid_csr(ctx: *mut RtCtx, meta: *const c_void, cell_start: *const u32, indices: *const u32) -> c_int;
    pub fn rt_build_grid(ctx: *mut RtCtx, cell_size: f32, max_per_cell: u32, mode: c_int) -> c_int;
    pub fn rt_update_vertices(ctx: *mut RtCtx, xyz9: *const f32, first: u32, count: u32) -> c_int;
    pub fn rt_trace_primary(ctx: *mut RtCtx, camera: *const c_void, width: u32, height: u32, shard: *const RtShard,
                            out_primary: *mut RtHit, out_shadow: *mut RtHit) -> c_int;
    /// `rt_trace_primary` without any device allocation (device outputs only; see include/rt_mi355.h).
    pub fn rt_trace_primary_noalloc(ctx: *mut RtCtx, camera: *const c_void, width: u32, height: u32, shard: *const RtShard,
                                    out_primary: *mut RtHit, out_shadow: *mut RtHit) -> c_int;
    pub fn rt_trace_rays(ctx: *mut RtCtx, camera: *const c_void, origins: *const f32, dirs: *const f32, n: u32,
                         out: *mut RtHit, any_hit: c_int) -> c_int;
    /// Several GPUs, one process each (include/rt_mi355.h, "several GPUs"): RCCL scene broadcast and framebuffer gather.
    pub fn rt_comm_unique_id(id: *mut u8) -> c_int;
    pub fn rt_comm_init(ctx: *mut RtCtx, id: *const u8, rank: c_int, n_ranks: c_int) -> c_int;
    pub fn rt_comm_info(ctx: *mut RtCtx, rank: *mut c_int, n_ranks: *mut c_int) -> c_int;
    pub fn rt_comm_destroy(ctx: *mut RtCtx) -> c_int;
    pub fn rt_broadcast_scene(ctx: *mut RtCtx, root: c_int, boxes: *const c_void, nb: u32, tris: *const c_void, nt: u32) -> c_int;
    pub fn rt_gather_frames(ctx: *mut RtCtx, root: c_int, local: *const c_void, bytes: usize, gathered: *mut c_void) -> c_int;
    pub fn rt_set_stream(ctx: *mut RtCtx, hip_stream: *mut c_void) -> c_int;
    pub fn rt_set_materials(ctx: *mut RtCtx, materials: *const c_void, n: u32) -> c_int;
    pub fn rt_get_trace_refs(ctx: *mut RtCtx, n_trace_refs: *mut u32) -> c_int;
    /// What the last planned `rt_trace_primary` launch decided for the next one (params.heavy_block_threshold).
    pub fn rt_get_heavy_blocks(ctx: *mut RtCtx, n_heavy: *mut u32, n_early: *mut u32, threshold_us: *mut u32) -> c_int;
    /// `rt_trace_primary` for up to 16 consecutive frames of a camera path in one launch.
    pub fn rt_trace_primary_frames(ctx: *mut RtCtx, cameras: *const c_void, n_frames: u32, width: u32, height: u32,
                                   shard: *const c_void, out_primary: *mut c_void, out_shadow: *mut c_void, frame_stride: usize) -> c_int;
    /// Frame blocks of the last `rt_trace_primary` launch and how many its block mask kept (params.cull_empty_blocks).
    pub fn rt_get_cull_stats(ctx: *mut RtCtx, blocks_total: *mut u32, blocks_kept: *mut u32) -> c_int;
    pub fn rt_set_textures(ctx: *mut RtCtx, textures: *const RtTexture, n: u32) -> c_int;
    pub fn rt_render_frames(ctx: *mut RtCtx, cameras: *const c_void, n_frames: u32, width: u32, height: u32,
                            shard: *const c_void, opts: *const c_void, out_rgba8: *mut u8) -> c_int;
    pub fn rt_assemble_frames_rgb8(ctx: *mut RtCtx, gathered: *const u8, width: u32, height: u32, tile_w: u32, tile_h: u32,
                                   n_ranks: u32, n_frames: u32, rank_stride: usize, frame_stride: usize, images: *mut u8) -> c_int;
    pub fn rt_assemble_frames(ctx: *mut RtCtx, gathered: *const u8, pixel_format: u32, width: u32, height: u32, tile_w: u32,
                              tile_h: u32, n_ranks: u32, owner: *const u32, round: u32, n_frames: u32, rank_stride: usize,
                              frame_stride: usize, images: *mut u8) -> c_int;
    /// flags: one byte per tile, 0 = the tile shows nothing but sky from this camera (nobody renders or sends it).
    pub fn rt_tile_flags(ctx: *mut RtCtx, camera: *const c_void, width: u32, height: u32, tile_w: u32, tile_h: u32,
                         flags: *mut u8) -> c_int;
    /// rt_assemble_frames for owner tables with RT_OWNER_SKY entries: those tiles are written from cameras[frame].
    pub fn rt_assemble_frames_sky(ctx: *mut RtCtx, gathered: *const u8, pixel_format: u32, width: u32, height: u32, tile_w: u32,
                                  tile_h: u32, n_ranks: u32, owner: *const u32, round: u32, cameras: *const c_void, n_frames: u32,
                                  rank_stride: usize, frame_stride: usize, images: *mut u8) -> c_int;
    pub fn rt_assemble_frames_rgba8(ctx: *mut RtCtx, gathered: *const u8, width: u32, height: u32, tile_w: u32, tile_h: u32,
                                    n_ranks: u32, n_frames: u32, rank_stride: usize, frame_stride: usize, images: *mut u8) -> c_int;
    pub fn rt_set_skin(ctx: *mut RtCtx, positions: *const f32, joints: *const u16, weights: *const f32, n_vertices: u32,
                       tri_indices: *const u32, n_triangles: u32) -> c_int;
    pub fn rt_skin_vertices(ctx: *mut RtCtx, joint_matrices: *const f32, n_joints: u32) -> c_int;
    pub fn rt_read_vertices(ctx: *mut RtCtx, xyz9: *mut f32) -> c_int;
    pub fn rt_build_bvh(ctx: *mut RtCtx) -> c_int;
    /// The same tree by the reference's recursion on one host thread (the GPU builder's checker).
    pub fn rt_build_bvh_host(ctx: *mut RtCtx) -> c_int;
    /// nodes: 8 f32 per node in pre-order; sizes from rt_get_bvh_stats. Any pointer may be null.
    pub fn rt_read_bvh(ctx: *mut RtCtx, nodes: *mut f32, leaf_first: *mut u32, prims: *mut u32) -> c_int;
    pub fn rt_trace_rays_bvh(ctx: *mut RtCtx, origins: *const f32, dirs: *const f32, n: u32, out_t: *mut f32) -> c_int;
    pub fn rt_intersect_tri_pairs_watertight(ctx: *mut RtCtx, origins: *const f32, dirs: *const f32, n_rays: u32,
                                             v012: *const f32, n_prims: u32, out_t: *mut f32) -> c_int;
    pub fn rt_default_render_opts(o: *mut RtRenderOpts);
    pub fn rt_render(ctx: *mut RtCtx, camera: *const c_void, width: u32, height: u32, shard: *const RtShard,
                     opts: *const RtRenderOpts, out_rgba8: *mut u8) -> c_int;
}

/// `rt_render_opts` of include/rt_mi355.h.
#[repr(C)]
#[derive(Copy, Clone, Default)]
pub struct RtRenderOpts { pub max_bounces: u32, pub shadows: u32, pub debug_enabled: u32, pub debug_pixel: [u32; 2], pub pixel_format: u32, pub frame_stride: u32, pub plain_bounces: u32 }

/// Owns one context. `Send` but not `Sync`: the C side is not re-entrant (one in-order stream).
pub struct TraceCore { ctx: *mut RtCtx }
unsafe impl Send for TraceCore {}

impl TraceCore {
    pub fn new(device_id: i32) -> Result<Self, String> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { rt_create(&mut ctx, device_id, std::ptr::null()) };
        if rc != 0 { return Err(unsafe { CStr::from_ptr(rt_create_error()) }.to_string_lossy().into_owned()); }
        Ok(Self { ctx })
    }
    fn check(&self, rc: c_int) -> Result<(), String> {
        if rc == 0 { Ok(()) } else { Err(unsafe { CStr::from_ptr(rt_last_error(self.ctx)) }.to_string_lossy().into_owned()) }
    }
    /// `boxes` / `triangles`: `bytemuck::cast_slice(&Vec<BoxData>)` / `(&Vec<TriangleData>)`.
    pub fn set_scene(&mut self, boxes: &[u8], triangles: &[u8]) -> Result<(), String> {
        assert!(boxes.len() % 96 == 0 && triangles.len() % 80 == 0);
        self.check(unsafe { rt_set_scene(self.ctx, boxes.as_ptr().cast(), (boxes.len() / 96) as u32,
                                         triangles.as_ptr().cast(), (triangles.len() / 80) as u32) })
    }
    /// On-GPU replacement of `HierarchicalGrid::build` (mode 0) / `TriangleGrid::build` (mode 1).
    pub fn build_grid(&mut self) -> Result<(), String> { self.check(unsafe { rt_build_grid(self.ctx, 0.0, 0, 0) }) }
    /// `camera`: `bytemuck::bytes_of(&CameraUniform)`. Returns (primary, shadow) hit records, row-major.
    pub fn trace_primary(&self, camera: &[u8], width: u32, height: u32) -> Result<(Vec<RtHit>, Vec<RtHit>), String> {
        assert_eq!(camera.len(), 80);
        let n = (width * height) as usize;
        let (mut p, mut s) = (vec![RtHit::default(); n], vec![RtHit::default(); n]);
        self.check(unsafe { rt_trace_primary(self.ctx, camera.as_ptr().cast(), width, height, std::ptr::null(),
                                             p.as_mut_ptr(), s.as_mut_ptr()) })?;
        Ok((p, s))
    }
}

/// `rt_texture`: one `TextureData` of the loader (src/loaders/gltf_triangles.rs:15-19), borrowed for the call.
#[repr(C)]
pub struct RtTexture {
    pub rgba8: *const u8,
    pub width: u32,
    pub height: u32,
}

impl TraceCore {
    /// `Vec<TextureData>` as `(width, height, RGBA8 bytes)`: what the renderer turns into its texture array
    /// (src/renderer.rs:158-259).
    pub fn set_textures(&mut self, textures: &[(u32, u32, &[u8])]) -> Result<(), String> {
        let recs: Vec<RtTexture> = textures.iter().map(|(w, h, data)| {
            assert!(data.len() >= (*w as usize) * (*h as usize) * 4);
            RtTexture { rgba8: data.as_ptr(), width: *w, height: *h }
        }).collect();
        self.check(unsafe { rt_set_textures(self.ctx, if recs.is_empty() { std::ptr::null() } else { recs.as_ptr() }, recs.len() as u32) })
    }

    /// `Vec<MaterialData>` bytes (src/types.rs:248-261), the renderer's "Materials" buffer.
    pub fn set_materials(&mut self, materials: &[u8]) -> Result<(), String> {
        self.check(unsafe { rt_set_materials(self.ctx, materials.as_ptr() as *const c_void, (materials.len() / 64) as u32) })
    }

    /// The Rgba8Unorm picture `ComputeState::render` returns (src/core/ray_tracing_layer.rs:436-493).
    pub fn render(&self, camera: &[u8], width: u32, height: u32) -> Result<Vec<u8>, String> {
        let mut rgba = vec![0u8; width as usize * height as usize * 4];
        self.check(unsafe { rt_render(self.ctx, camera.as_ptr() as *const c_void, width, height, std::ptr::null(),
                                      std::ptr::null(), rgba.as_mut_ptr()) })?;
        Ok(rgba)
    }
}

impl Drop for TraceCore { fn drop(&mut self) { unsafe { rt_destroy(self.ctx) } } }
