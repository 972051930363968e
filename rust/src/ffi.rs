//! `extern "C"` view of include/rt_abi.h (librt_amd.so): every struct, constant and entry point, in
//! the header's order.  tests/test_rust_binding.py compares this file with the header field by
//! field, so a change on either side that is not mirrored fails the build's tests.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_HIP: c_int = -2;
pub const RT_ERR_LIMIT: c_int = -3;
pub const RT_ERR_NO_DEVICE: c_int = -4;

pub const RT_MAX_LIGHTS: i32 = 16;
pub const RT_MAX_DEPTH: i32 = 5;
pub const RT_MAX_XFORM_NEST: i32 = 8;
pub const RT_MAX_PATTERN_NEST: i32 = 8;
pub const RT_TILE: i32 = 8;
pub const RT_MAX_CSG_NEST: i32 = 8;
pub const RT_STATIC_MAX_TILES: i32 = 1 << 19;
pub const RT_MAX_CSG_HITS: i32 = 128;

// node kinds
pub const RT_GROUP: i32 = 0;
pub const RT_SPHERE: i32 = 1;
pub const RT_PLANE: i32 = 2;
pub const RT_CUBE: i32 = 3;
pub const RT_CYLINDER: i32 = 4;
pub const RT_CONE: i32 = 5;
pub const RT_TRIANGLE: i32 = 6;
pub const RT_SMOOTH_TRIANGLE: i32 = 7;
pub const RT_CSG: i32 = 8;
// node flags
pub const RT_NODE_CASTS_SHADOW: i32 = 1;
pub const RT_NODE_CLOSED: i32 = 2;
pub const RT_NODE_CSG_DIFFERENCE: i32 = 0 << 4;
pub const RT_NODE_CSG_INTERSECTION: i32 = 1 << 4;
pub const RT_NODE_CSG_UNION: i32 = 2 << 4;
pub const RT_NODE_CSG_OP_MASK: i32 = 3 << 4;
// pattern kinds
pub const RT_PAT_SOLID: i32 = 0;
pub const RT_PAT_STRIPED: i32 = 1;
pub const RT_PAT_GRADIENT: i32 = 2;
pub const RT_PAT_RING: i32 = 3;
pub const RT_PAT_CHECKER: i32 = 4;
pub const RT_PAT_BLENDED: i32 = 5;
pub const RT_PAT_PERTURBED: i32 = 6;
pub const RT_PAT_TEST: i32 = 7;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_node {
    pub r#type: i32,
    pub flags: i32,
    pub parent: i32,
    pub skip: i32,
    pub material: i32,
    pub xform: i32,
    pub data: i32,
    pub ctx: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_xform {
    pub inv: [f64; 12],
    pub inv_t: [f64; 9],
    pub reserved: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_aabb {
    pub min: [f64; 3],
    pub max: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_range {
    pub minimum: f64,
    pub maximum: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_tri {
    pub p1: [f64; 3],
    pub e1: [f64; 3],
    pub e2: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_tri_normals {
    pub n1: [f64; 3],
    pub n2: [f64; 3],
    pub n3: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_material {
    pub ambient: f64,
    pub diffuse: f64,
    pub specular: f64,
    pub shininess: f64,
    pub reflective: f64,
    pub transparency: f64,
    pub refractive_index: f64,
    pub pattern: i32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_pattern {
    pub r#type: i32,
    pub a: i32,
    pub b: i32,
    pub xform: i32,
    pub color: [f64; 3],
    pub scale: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_light {
    pub position: [f64; 3],
    pub intensity: [f64; 3],
}

#[repr(C)]
pub struct rt_scene_desc {
    pub n_nodes: i32,
    pub nodes: *const rt_node,
    pub n_xforms: i32,
    pub xforms: *const rt_xform,
    pub n_aabbs: i32,
    pub aabbs: *const rt_aabb,
    pub n_ranges: i32,
    pub ranges: *const rt_range,
    pub n_tris: i32,
    pub tris: *const rt_tri,
    pub tri_normals: *const rt_tri_normals,
    pub n_materials: i32,
    pub materials: *const rt_material,
    pub n_patterns: i32,
    pub patterns: *const rt_pattern,
    pub n_pattern_xforms: i32,
    pub pattern_xforms: *const f64,
    pub n_lights: i32,
    pub lights: *const rt_light,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_camera {
    pub width: i32,
    pub height: i32,
    pub half_width: f64,
    pub half_height: f64,
    pub pixel_size: f64,
    pub inv: [f64; 12],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rt_params {
    pub max_depth: i32,
    pub rank: i32,
    pub world_size: i32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_stats {
    pub rays_primary: u64,
    pub rays_shadow: u64,
    pub rays_reflect: u64,
    pub rays_refract: u64,
    pub kernel_ms: f32,
    pub reserved: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_build_info {
    pub validate_ms: f32,
    pub records_ms: f32,
    pub upload_ms: f32,
    pub n_nodes: i32,
    pub n_records: i32,
    pub n_cull_records: i32,
    pub reserved: i32,
    pub walk_bytes: u64,
    pub table_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rt_multi_stats {
    pub rays_primary: u64,
    pub rays_shadow: u64,
    pub rays_reflect: u64,
    pub rays_refract: u64,
    pub kernel_ms: f32,
    pub render_wall_ms: f32,
    pub exchange_ms: f32,
    pub total_ms: f32,
}

/// Opaque handles.
pub enum rt_scene {}
pub enum rt_multi {}

extern "C" {
    pub fn rt_device_count() -> c_int;
    pub fn rt_scene_create(desc: *const rt_scene_desc, device: c_int, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_scene_build_info(scene: *const rt_scene, out: *mut rt_build_info) -> c_int;
    pub fn rt_local_rows(height: i32, params: *const rt_params) -> i64;
    pub fn rt_global_row(local_row: i64, params: *const rt_params) -> i64;
    pub fn rt_render_device(scene: *mut rt_scene, camera: *const rt_camera, params: *const rt_params, d_rgb: *mut f64,
                            hip_stream: *mut c_void) -> c_int;
    pub fn rt_render_device_static(scene: *mut rt_scene, camera: *const rt_camera, params: *const rt_params,
                                   d_rgb: *mut f64, hip_stream: *mut c_void) -> c_int;
    pub fn rt_render(scene: *mut rt_scene, camera: *const rt_camera, params: *const rt_params, rgb_out: *mut f64,
                     stats: *mut rt_stats) -> c_int;
    pub fn rt_render_device_u8(scene: *mut rt_scene, camera: *const rt_camera, params: *const rt_params,
                               d_rgb8: *mut u8, hip_stream: *mut c_void) -> c_int;
    pub fn rt_render_u8(scene: *mut rt_scene, camera: *const rt_camera, params: *const rt_params, rgb8_out: *mut u8,
                        stats: *mut rt_stats) -> c_int;
    pub fn rt_render_ppm(scene: *mut rt_scene, camera: *const rt_camera, params: *const rt_params, text_out: *mut c_char,
                         capacity: u64, length: *mut u64, stats: *mut rt_stats) -> c_int;
    pub fn rt_ppm_encode(device: c_int, rgb8: *const u8, width: i32, height: i32, text_out: *mut c_char, capacity: u64,
                         length: *mut u64) -> c_int;
    pub fn rt_color_at(scene: *mut rt_scene, n: i64, origins: *const f64, directions: *const f64, remaining: i32,
                       rgb_out: *mut f64, stats: *mut rt_stats) -> c_int;
    pub fn rt_host_alloc(bytes: u64) -> *mut c_void;
    pub fn rt_host_free(p: *mut c_void);
    pub fn rt_scene_launch_times(scene: *mut rt_scene, ms: *mut f32, cap: c_int) -> c_int;
    pub fn rt_scene_ray_counts(scene: *mut rt_scene, stats: *mut rt_stats) -> c_int;
    pub fn rt_multi_create(desc: *const rt_scene_desc, n_ranks: i32, devices: *const i32, out: *mut *mut rt_multi) -> c_int;
    pub fn rt_multi_destroy(multi: *mut rt_multi);
    pub fn rt_multi_ranks(multi: *const rt_multi) -> i32;
    pub fn rt_render_multi(multi: *mut rt_multi, camera: *const rt_camera, max_depth: i32, rgb_out: *mut f64,
                           stats: *mut rt_multi_stats) -> c_int;
    pub fn rt_render_multi_u8(multi: *mut rt_multi, camera: *const rt_camera, max_depth: i32, rgb8_out: *mut u8,
                              stats: *mut rt_multi_stats) -> c_int;
    pub fn rt_render_multi_device(multi: *mut rt_multi, camera: *const rt_camera, max_depth: i32, d_frame: *mut c_void,
                                  u8: i32, stats: *mut rt_multi_stats) -> c_int;
    pub fn rt_selftest_division(device: c_int, n: i64, a: *const f64, d: *const f64, out_fast: *mut f64,
                                out_ieee: *mut f64) -> c_int;
    pub fn rt_last_error() -> *const c_char;
}
