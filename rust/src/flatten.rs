//! World -> the plain tables of include/rt_abi.h, and the replacement of `render`.
//!
//! The crate's scene graph is a pointer graph of `Arc<dyn Shape>` with private fields.  Three small
//! trait methods expose what the tables need and nothing else (their bodies are in
//! `src/additions/*.rs`, one file per module they are appended to):
//!
//!   trait Shape    { fn flat(&self) -> FlatShape; }        // kind, private parameters, children
//!   trait Material { fn flat(&self) -> FlatMaterial<'_>; } // Phong coefficients and the pattern
//!   trait Pattern  { fn flat(&self) -> FlatPattern<'_>; }  // kind, sub-patterns, colour / scale
//!
//! Everything else (transform, material, casts_shadow, bounds) comes through the getters the traits
//! already have.  The tables are laid out exactly as the C++ host of this library lays them out
//! (raytracer_amd/host/rt_host.cpp, `Flattener`): nodes depth first in pre-order, children in
//! `Group::children` order, one rt_xform per non-identity transform, materials de-duplicated by
//! `Arc` identity, patterns in pre-order with children after their parent.
//!
//! Every inverse is the crate's own `Matrix::inverse` (cofactor expansion): the device multiplies by
//! exactly the bits the CPU path computes per ray (src/geometry/shape.rs:24, 77, 86).

use std::ffi::CStr;
use std::sync::Arc;

use crate::ffi::*;
use crate::geometry::shape::Shape;
use crate::materials::material::Material;
use crate::matrices::matrix::Matrix;
use crate::patterns::pattern::Pattern;
use crate::scene::camera::Camera;
use crate::scene::world::World;
use crate::tuples::bounding_box::BoundingBox;
use crate::tuples::color::Color;
use crate::tuples::point_light::PointLight;
use crate::tuples::tuple::Tuple;
use crate::window::canvas::Canvas;
use crate::MAX_RAY_RECURSION_DEPTH;

/// What a shape is, beyond what `Shape`'s getters already tell.
pub enum FlatShape {
    Group { children: Vec<Arc<dyn Shape>> },
    /// operation: 0 difference, 1 intersection, 2 union (src/geometry/csg.rs:27-31)
    Csg { operation: i32, left: Arc<dyn Shape>, right: Arc<dyn Shape> },
    Sphere,
    Plane,
    Cube,
    Cylinder { minimum: f64, maximum: f64, closed: bool },
    Cone { minimum: f64, maximum: f64, closed: bool },
    Triangle { p1: Tuple, e1: Tuple, e2: Tuple, normal: Tuple },
    SmoothTriangle { p1: Tuple, e1: Tuple, e2: Tuple, n1: Tuple, n2: Tuple, n3: Tuple },
}

/// Phong's fields (src/materials/phong.rs:10-19).
pub struct FlatMaterial<'a> {
    pub pattern: &'a dyn Pattern,
    pub ambient: f64,
    pub diffuse: f64,
    pub specular: f64,
    pub shininess: f64,
    pub reflective: f64,
    pub transparency: f64,
    pub refractive_index: f64,
}

/// A pattern node (src/patterns/*.rs).  `transform` is `Pattern::get_transform`.
pub enum FlatPattern<'a> {
    Solid { color: Color },
    /// kind: RT_PAT_STRIPED, RT_PAT_GRADIENT, RT_PAT_RING, RT_PAT_CHECKER or RT_PAT_BLENDED
    Binary { kind: i32, former: &'a dyn Pattern, latter: &'a dyn Pattern },
    Perturbed { delegate: &'a dyn Pattern, scale: f64 },
    Test,
}

#[derive(Default)]
pub struct SceneBuilder {
    pub nodes: Vec<rt_node>,
    pub xforms: Vec<rt_xform>,
    pub aabbs: Vec<rt_aabb>,
    pub ranges: Vec<rt_range>,
    pub tris: Vec<rt_tri>,
    pub tri_normals: Vec<rt_tri_normals>,
    pub materials: Vec<rt_material>,
    pub patterns: Vec<rt_pattern>,
    pub pattern_xforms: Vec<f64>,
    pub lights: Vec<rt_light>,
    /// materials already added, by identity (`Arc::ptr_eq`), so that shapes sharing a material share its entry
    seen_materials: Vec<(Arc<dyn Material>, i32)>,
}

fn xyz(t: Tuple) -> [f64; 3] {
    [t.x, t.y, t.z]
}

/// Exact test (the crate's `==` on matrices is tolerant): only a bit-exact identity may be skipped.
fn is_identity(m: &Matrix) -> bool {
    for r in 0..4 {
        for c in 0..4 {
            if *m.get(r, c).unwrap() != if r == c { 1.0 } else { 0.0 } {
                return false;
            }
        }
    }
    true
}

/// Rows 0..2 of the cofactor inverse and the 3x3 of its transpose.  Fails where the reference would
/// panic (|det| < EPSILON, src/matrices/matrix.rs:204) and on anything whose inverse is not affine.
fn make_xform(transform: &Matrix) -> Result<rt_xform, String> {
    let inv = transform.inverse().map_err(|e| e.to_string())?;
    let at = |r: usize, c: usize| *inv.get(r, c).unwrap();
    if at(3, 0) != 0.0 || at(3, 1) != 0.0 || at(3, 2) != 0.0 || at(3, 3) != 1.0 {
        return Err("transform is not affine (bottom row of the inverse is not 0 0 0 1)".to_string());
    }
    let mut x = rt_xform { inv: [0.0; 12], inv_t: [0.0; 9], reserved: [0.0; 3] };
    for r in 0..3 {
        for c in 0..4 {
            x.inv[r * 4 + c] = at(r, c);
        }
    }
    for i in 0..3 {
        for k in 0..3 {
            x.inv_t[i * 3 + k] = at(k, i);
        }
    }
    Ok(x)
}

fn aabb_of(b: &BoundingBox) -> rt_aabb {
    rt_aabb { min: xyz(b.min()), max: xyz(b.max()) }
}

impl SceneBuilder {
    pub fn new() -> SceneBuilder {
        SceneBuilder::default()
    }

    /// Appends the pattern tree under `p` in pre-order; returns the index of its root.
    pub fn add_pattern(&mut self, p: &dyn Pattern) -> Result<i32, String> {
        let idx = self.patterns.len();
        self.patterns.push(rt_pattern { r#type: RT_PAT_SOLID, a: -1, b: -1, xform: -1, color: [1.0, 1.0, 1.0], scale: 0.0 });
        let mut rp = self.patterns[idx];
        let flat = p.flat();
        // Solid's transform is the identity by construction (src/patterns/solid.rs:26-28)
        if !matches!(flat, FlatPattern::Solid { .. }) {
            let t = p.get_transform();
            if !is_identity(&t) {
                let x = make_xform(&t)?;
                rp.xform = (self.pattern_xforms.len() / 12) as i32;
                self.pattern_xforms.extend_from_slice(&x.inv);
            }
        }
        match flat {
            FlatPattern::Solid { color } => {
                rp.r#type = RT_PAT_SOLID;
                rp.color = [color.red, color.green, color.blue];
            }
            FlatPattern::Binary { kind, former, latter } => {
                rp.r#type = kind;
                rp.a = self.add_pattern(former)?;
                rp.b = self.add_pattern(latter)?;
            }
            FlatPattern::Perturbed { delegate, scale } => {
                rp.r#type = RT_PAT_PERTURBED;
                rp.scale = scale;
                rp.a = self.add_pattern(delegate)?;
            }
            FlatPattern::Test => rp.r#type = RT_PAT_TEST,
        }
        self.patterns[idx] = rp;
        Ok(idx as i32)
    }

    pub fn add_material(&mut self, m: &Arc<dyn Material>) -> Result<i32, String> {
        if let Some((_, i)) = self.seen_materials.iter().find(|(seen, _)| Arc::ptr_eq(seen, m)) {
            return Ok(*i);
        }
        let f = m.flat();
        let pattern = self.add_pattern(f.pattern)?;
        self.materials.push(rt_material {
            ambient: f.ambient,
            diffuse: f.diffuse,
            specular: f.specular,
            shininess: f.shininess,
            reflective: f.reflective,
            transparency: f.transparency,
            refractive_index: f.refractive_index,
            pattern,
            reserved: 0,
        });
        let idx = (self.materials.len() - 1) as i32;
        self.seen_materials.push((m.clone(), idx));
        Ok(idx)
    }

    /// Appends `shape` and, for groups and CSG shapes, everything below it, depth first.
    /// parent: index of the enclosing group or CSG shape (-1 for a world object); ctx: nearest
    /// enclosing node whose transform is not the identity (-1 if none).
    pub fn add_shape(&mut self, shape: &Arc<dyn Shape>, parent: i32, ctx: i32) -> Result<(), String> {
        let idx = self.nodes.len();
        let shadow = if shape.casts_shadow() { RT_NODE_CASTS_SHADOW } else { 0 };
        let mut n = rt_node { r#type: RT_GROUP, flags: shadow, parent, skip: 0, material: -1, xform: -1, data: -1, ctx };
        self.nodes.push(n);
        let transform = shape.get_transform();
        if !is_identity(&transform) {
            n.xform = self.xforms.len() as i32;
            self.xforms.push(make_xform(&transform)?);
        }
        let flat = shape.flat();
        let children: Option<Vec<Arc<dyn Shape>>> = match &flat {
            FlatShape::Group { children } => Some(children.clone()),
            FlatShape::Csg { operation, left, right } => {
                n.r#type = RT_CSG;
                n.flags = shadow | (*operation << 4);
                Some(vec![left.clone(), right.clone()])
            }
            _ => None,
        };
        if let Some(children) = children {
            // Group::bounds / CSG::bounds (src/geometry/group.rs:196-208, src/geometry/csg.rs:74-78)
            n.data = self.aabbs.len() as i32;
            self.aabbs.push(aabb_of(&shape.bounds()));
            self.nodes[idx] = n;
            let child_ctx = if n.xform >= 0 { idx as i32 } else { ctx };
            for c in &children {
                self.add_shape(c, idx as i32, child_ctx)?;
            }
            self.nodes[idx].skip = self.nodes.len() as i32;
            return Ok(());
        }
        n.material = self.add_material(&shape.get_material())?;
        match flat {
            FlatShape::Sphere => n.r#type = RT_SPHERE,
            FlatShape::Plane => n.r#type = RT_PLANE,
            FlatShape::Cube => n.r#type = RT_CUBE,
            FlatShape::Cylinder { minimum, maximum, closed } | FlatShape::Cone { minimum, maximum, closed } => {
                n.r#type = if matches!(flat, FlatShape::Cylinder { .. }) { RT_CYLINDER } else { RT_CONE };
                if closed {
                    n.flags |= RT_NODE_CLOSED;
                }
                n.data = self.ranges.len() as i32;
                self.ranges.push(rt_range { minimum, maximum });
            }
            FlatShape::Triangle { p1, e1, e2, normal } => {
                n.r#type = RT_TRIANGLE;
                n.data = self.tris.len() as i32;
                self.tris.push(rt_tri { p1: xyz(p1), e1: xyz(e1), e2: xyz(e2) });
                // the precomputed face normal (src/geometry/triangle.rs:48); n2, n3 unused
                self.tri_normals.push(rt_tri_normals { n1: xyz(normal), n2: [0.0; 3], n3: [0.0; 3] });
            }
            FlatShape::SmoothTriangle { p1, e1, e2, n1, n2, n3 } => {
                n.r#type = RT_SMOOTH_TRIANGLE;
                n.data = self.tris.len() as i32;
                self.tris.push(rt_tri { p1: xyz(p1), e1: xyz(e1), e2: xyz(e2) });
                self.tri_normals.push(rt_tri_normals { n1: xyz(n1), n2: xyz(n2), n3: xyz(n3) });
            }
            FlatShape::Group { .. } | FlatShape::Csg { .. } => unreachable!(),
        }
        n.skip = idx as i32 + 1;
        self.nodes[idx] = n;
        Ok(())
    }

    pub fn add_light(&mut self, l: &PointLight) {
        self.lights.push(rt_light {
            position: xyz(l.position),
            intensity: [l.intensity.red, l.intensity.green, l.intensity.blue],
        });
    }

    /// The description rt_scene_create takes; it borrows the builder's vectors.
    pub fn desc(&self) -> rt_scene_desc {
        rt_scene_desc {
            n_nodes: self.nodes.len() as i32,
            nodes: self.nodes.as_ptr(),
            n_xforms: self.xforms.len() as i32,
            xforms: self.xforms.as_ptr(),
            n_aabbs: self.aabbs.len() as i32,
            aabbs: self.aabbs.as_ptr(),
            n_ranges: self.ranges.len() as i32,
            ranges: self.ranges.as_ptr(),
            n_tris: self.tris.len() as i32,
            tris: self.tris.as_ptr(),
            tri_normals: self.tri_normals.as_ptr(),
            n_materials: self.materials.len() as i32,
            materials: self.materials.as_ptr(),
            n_patterns: self.patterns.len() as i32,
            patterns: self.patterns.as_ptr(),
            n_pattern_xforms: (self.pattern_xforms.len() / 12) as i32,
            pattern_xforms: self.pattern_xforms.as_ptr(),
            n_lights: self.lights.len() as i32,
            lights: self.lights.as_ptr(),
        }
    }
}

/// `World::flatten` (the body of the method added to src/scene/world.rs, which owns the private
/// `objects` and `lights`): objects in `World::objects` order, then the lights in order.
pub fn flatten_world(objects: &[Arc<dyn Shape>], lights: &[Arc<PointLight>]) -> Result<SceneBuilder, String> {
    let mut b = SceneBuilder::new();
    for o in objects {
        b.add_shape(o, -1, -1)?;
    }
    for l in lights {
        b.add_light(l);
    }
    Ok(b)
}

/// `Camera::to_rt_camera` (body of the method added to src/scene/camera.rs): what Camera::new derived
/// (src/scene/camera.rs:17-43) and rows 0..2 of the inverse that ray_for_pixel computes per pixel
/// (camera.rs:60).  width/height are Camera::width()/height(), i.e. vsize/hsize (camera.rs:68-74).
pub fn camera_block(width: usize, height: usize, half_width: f64, half_height: f64, pixel_size: f64,
                    transform: &Matrix) -> Result<rt_camera, String> {
    let inv = transform.inverse().map_err(|e| e.to_string())?;
    let mut rows = [0.0f64; 12];
    for r in 0..3 {
        for c in 0..4 {
            rows[r * 4 + c] = *inv.get(r, c).unwrap();
        }
    }
    Ok(rt_camera { width: width as i32, height: height as i32, half_width, half_height, pixel_size, inv: rows })
}

fn check(status: i32) {
    if status != RT_OK {
        let msg = unsafe { CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned();
        // the crate's CPU path unwraps its errors (e.g. a singular transform panics inside
        // Shape::intersect); the GPU path reports the same conditions here
        panic!("rt_amd error {status}: {msg}");
    }
}

/// Replacement of `pub fn render(world: Arc<World>, camera: Arc<Camera>) -> Canvas` (src/lib.rs:77-133),
/// same signature: the twelve worker threads and the channel become one call into the library, on
/// every GPU of this process (one GPU: `rt_multi` with one rank is `rt_render`).
pub fn render(world: Arc<World>, camera: Arc<Camera>) -> Canvas {
    let flat = world.flatten().unwrap_or_else(|e| panic!("{e}"));
    let cam = camera.to_rt_camera().unwrap_or_else(|e| panic!("{e}"));
    let (w, h) = (camera.width(), camera.height());
    let n_gpus = unsafe { rt_device_count() }.max(1);
    let mut rgb = vec![0f64; w * h * 3];
    unsafe {
        let mut multi: *mut rt_multi = std::ptr::null_mut();
        check(rt_multi_create(&flat.desc(), n_gpus, std::ptr::null(), &mut multi));
        let rc = rt_render_multi(multi, &cam, MAX_RAY_RECURSION_DEPTH as i32, rgb.as_mut_ptr(), std::ptr::null_mut());
        rt_multi_destroy(multi);
        check(rc);
    }
    let mut canvas = Canvas::new(w, h);
    for y in 0..h {
        for x in 0..w {
            let i = (y * w + x) * 3;
            canvas.write_pixel(x, y, Color::new(rgb[i], rgb[i + 1], rgb[i + 2])).unwrap();
        }
    }
    canvas
}

/// What `run` (src/lib.rs:57-75) needs in the end is the PPM text: render, quantise
/// (Canvas::clamp, src/window/canvas.rs:50-60) and format (Canvas::to_ppm, canvas.rs:62-126) on the
/// device, one copy of the text back.  Single GPU.
pub fn render_to_ppm(world: Arc<World>, camera: Arc<Camera>) -> String {
    let flat = world.flatten().unwrap_or_else(|e| panic!("{e}"));
    let cam = camera.to_rt_camera().unwrap_or_else(|e| panic!("{e}"));
    let prm = rt_params { max_depth: MAX_RAY_RECURSION_DEPTH as i32, rank: 0, world_size: 1, reserved: 0 };
    let capacity = 32 + camera.height() * (camera.width() * 12 + 1);
    let mut text = vec![0u8; capacity];
    let mut length = 0u64;
    unsafe {
        let mut scene: *mut rt_scene = std::ptr::null_mut();
        check(rt_scene_create(&flat.desc(), 0, &mut scene));
        let rc = rt_render_ppm(scene, &cam, &prm, text.as_mut_ptr() as *mut _, capacity as u64, &mut length,
                               std::ptr::null_mut());
        rt_scene_destroy(scene);
        check(rc);
    }
    text.truncate(length as usize);
    String::from_utf8(text).expect("PPM text is ASCII")
}
