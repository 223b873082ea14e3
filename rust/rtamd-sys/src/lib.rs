//! Raw bindings to `include/rtamd.h`, field for field.  UNTESTED here (no Rust toolchain in the build
//! environment).  The struct layouts follow `abi_layout.json` beside this crate: size of every struct, offset and
//! size of every field, as `tests/abi_check.c` pins them with `_Static_assert`s in a C11 build of the header and as
//! `tests/test_abi.py` compares them with the ctypes mirror (`raytracer-rust_amd/lib.py`).  `#[repr(C)]` with the
//! field types below reproduces exactly those offsets on LP64.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct rt_context {
    _private: [u8; 0],
}

pub const RT_MAT_LAMBERTIAN: i32 = 0;
pub const RT_MAT_METAL: i32 = 1;
pub const RT_MAT_DIELECTRIC: i32 = 2;
pub const RT_MAT_TEXTURE: i32 = 3;
pub const RT_MAT_EMISSIVE: i32 = 4;
pub const RT_SSAA_UNIFORM: i32 = 0;
pub const RT_SSAA_JITTER: i32 = 1;
pub const RT_MAX_ITERATIONS: usize = 320;
pub const RT_FLAG_COUNTERS: u32 = 1;
pub const RT_FLAG_TIMING: u32 = 2;
pub const RT_NOALLOC_SCRATCH_BYTES: usize = 64 << 20;

#[repr(C)]
pub struct rt_scene_desc {
    pub width: u32,
    pub height: u32,
    pub background: [f64; 3],
    pub cam_eye: [f64; 3],
    pub cam_lower_left: [f64; 3],
    pub cam_horizontal: [f64; 3],
    pub cam_vertical: [f64; 3],
    pub n_materials: u32,
    pub mat_type: *const i32,
    pub mat_rgb: *const f64,
    pub mat_param: *const f64,
    pub mat_texture: *const i32,
    pub n_textures: u32,
    pub tex_width: *const u32,
    pub tex_height: *const u32,
    pub tex_rgb8: *const *const u8,
    pub n_spheres: u32,
    pub sphere_center: *const f64,
    pub sphere_radius: *const f64,
    pub sphere_material: *const i32,
    pub sphere_object: *const i32,
    pub n_planes: u32,
    pub plane_center: *const f64,
    pub plane_normal: *const f64,
    pub plane_material: *const i32,
    pub plane_object: *const i32,
    pub n_triangles: u64,
    pub tri_vertices: *const f64,
    pub tri_uv: *const f64,
    pub tri_has_uv: *const u8,
    pub tri_material: *const i32,
    pub tri_object: *const i32,
    pub tri_index: *const i32,
    pub n_lights: u32,
    pub light_point: *const f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_render_params {
    pub ssaa_mode: i32,
    pub ssaa_n: u32,
    pub depth: u32,
    pub seed: u64,
    pub sample_begin: u32,
    pub sample_end: u32,
    pub tile_index: u32,
    pub tile_count: u32,
    pub max_paths_per_pass: u32,
    pub max_shadow_generations: u32,
    pub flags: u32,
    pub sample_offset: [f64; 2],
}

#[repr(C)]
pub struct rt_stats {
    pub rays_total: u64,
    pub rays_per_iteration: [u64; RT_MAX_ITERATIONS],
    pub paths: u64,
    pub nodes_visited: u64,
    pub prims_tested: u64,
    pub queue_overflow: u64,
    pub hit_launches: u32,
    pub hit_ms: f64,
    pub bvh_nodes: u32,
    pub bvh_depth: u32,
    pub bvh_prims: u32,
    pub bvh_build_ms: f64,
    pub wave_node_steps: u64,
    pub wave_leaf_steps: u64,
    pub wave_refills: u64,
    pub self_hit_rays: u64,
    pub step_lanes: u64,
}

extern "C" {
    pub fn rt_context_create(device_id: c_int, out: *mut *mut rt_context) -> c_int;
    pub fn rt_context_destroy(ctx: *mut rt_context);
    pub fn rt_last_error(ctx: *const rt_context) -> *const c_char;
    pub fn rt_scene_upload(ctx: *mut rt_context, scene: *const rt_scene_desc) -> c_int;
    pub fn rt_render_accumulate(ctx: *mut rt_context, params: *const rt_render_params, accum_dev: *mut f32, stream: *mut c_void) -> c_int;
    pub fn rt_resolve(ctx: *mut rt_context, accum_dev: *const f32, spp_total: u32, rgb8_dev: *mut u8, color_dev: *mut f32, stream: *mut c_void) -> c_int;
    pub fn rt_render_image(ctx: *mut rt_context, params: *const rt_render_params, accum_host: *mut f32, rgb8_host: *mut u8) -> c_int;
    pub fn rt_primary_hits(ctx: *mut rt_context, dx: f64, dy: f64, object_id: *mut i32, triangle_id: *mut i32, t: *mut f32, normal_xyz: *mut f32) -> c_int;
    pub fn rt_primary_rays(ctx: *mut rt_context, dx: f64, dy: f64, origin_xyz: *mut f32, dir_xyz: *mut f32) -> c_int;
    pub fn rt_render_accumulate_noalloc(ctx: *mut rt_context, params: *const rt_render_params, accum_dev: *mut f32, stream: *mut c_void) -> c_int;
    pub fn rt_scene_upload_devices(ctxs: *const *mut rt_context, n: u32, scene: *const rt_scene_desc) -> c_int;
    pub fn rt_accum_reduce(dst: *mut rt_context, dst_accum_dev: *mut f32, src: *const *mut rt_context, src_accum_dev: *const *const f32, n_src: u32, stream: *mut c_void) -> c_int;
    pub fn rt_accum_reduce_rccl(ctxs: *const *mut rt_context, accum_dev: *const *mut f32, n: u32) -> c_int;
    pub fn rt_render_image_devices(ctxs: *const *mut rt_context, n: u32, params: *const rt_render_params, split_tiles: c_int, accum_host: *mut f32, rgb8_host: *mut u8) -> c_int;
    pub fn rt_get_stats(ctx: *mut rt_context, out: *mut rt_stats) -> c_int;
    pub fn rt_stats_reset(ctx: *mut rt_context) -> c_int;
    pub fn rt_debug_gather(ctx: *mut rt_context, table_bytes: u64, iters: u32, bytes_loaded: *mut f64) -> c_int;
    pub fn rt_context_set_bvh_builder(ctx: *mut rt_context, builder: c_int) -> c_int;
    pub fn rt_bvh_build_validate(ctx: *mut rt_context, n: u64, boxes: *const f32, n_nodes: *mut u32, depth: *mut u32, max_stack: *mut u32, build_ms: *mut f64) -> c_int;
}
