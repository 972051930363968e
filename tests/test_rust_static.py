"""Static evidence for the Rust side (rust/src): the crate cannot be compiled where this library is
built (no cargo), so what CAN be checked is checked against the reference's sources as text.  Every
private field and every method that rust/src/additions/*.rs and rust/src/flatten.rs use on one of the
reference's types must exist under that name, with the type or signature the use needs, in the reference
file the addition is appended to; and the three traits that gain a `flat` method must have the getters
flatten.rs calls.  Runs only where the reference checkout is present (it never travels to the GPU box)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference checkout is not here")

# type -> the reference file its `impl` block is appended to
HOME = {
    "Sphere": "src/geometry/sphere.rs", "Plane": "src/geometry/plane.rs", "Cube": "src/geometry/cube.rs",
    "Cylinder": "src/geometry/cylinder.rs", "Cone": "src/geometry/cone.rs", "Triangle": "src/geometry/triangle.rs",
    "SmoothTriangle": "src/geometry/smooth_triangle.rs", "Group": "src/geometry/group.rs", "CSG": "src/geometry/csg.rs",
    "TestShape": "src/geometry/test_shape.rs", "Phong": "src/materials/phong.rs", "Solid": "src/patterns/solid.rs",
    "Striped": "src/patterns/striped.rs", "Gradient": "src/patterns/gradient.rs", "Ring": "src/patterns/ring.rs",
    "Checker": "src/patterns/checker.rs", "Blended": "src/patterns/blended.rs", "Perturbed": "src/patterns/perturbed.rs",
    "TestPattern": "src/patterns/test_pattern.rs", "World": "src/scene/world.rs", "Camera": "src/scene/camera.rs",
}


def ref(path):
    return open(os.path.join(REF, path)).read()


def rust(path):
    return open(os.path.join(ROOT, "rust", "src", path)).read()


def struct_fields(text, name):
    m = re.search(r"pub struct %s\s*\{(.*?)\n\}" % name, text, flags=re.S)
    assert m, "no struct %s" % name
    body = re.sub(r"//[^\n]*", "", m.group(1))
    return {f.group(1): " ".join(f.group(2).split()) for f in re.finditer(r"(?:pub )?(\w+)\s*:\s*([^,\n]+(?:<[^\n]*>)?)\s*,", body)}


def methods(text):
    return set(re.findall(r"\bfn (\w+)\s*[<(]", text))


def impl_blocks(text):
    """(type, body) of every inherent `impl X { ... }` block."""
    out = []
    for m in re.finditer(r"^impl (\w+) \{", text, flags=re.M):
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        out.append((m.group(1), text[m.end():i - 1]))
    return out


def addition_blocks():
    for f in ("geometry.rs", "materials_patterns.rs", "scene.rs"):
        for typ, body in impl_blocks(rust("additions/" + f)):
            yield f, typ, body


def test_every_type_that_gets_an_addition_exists_where_it_is_appended():
    seen = set()
    for f, typ, _ in addition_blocks():
        assert typ in HOME, (f, typ)
        assert re.search(r"pub struct %s\b" % typ, ref(HOME[typ])), (typ, HOME[typ])
        seen.add(typ)
    # every shape, the material and every pattern of the reference has one
    assert seen == set(HOME)
    for d, trait in (("src/geometry", "Shape"), ("src/patterns", "Pattern"), ("src/materials", "Material")):
        for name in os.listdir(os.path.join(REF, d)):
            for typ in re.findall(r"impl %s for (\w+)" % trait, ref(os.path.join(d, name))):
                assert typ in seen, "the reference's %s (%s/%s) has no flat_node" % (typ, d, name)


def test_private_fields_and_methods_used_by_the_additions_exist():
    checked = 0
    for f, typ, body in addition_blocks():
        text = ref(HOME[typ])
        fields, fns = struct_fields(text, typ), methods(text)
        for name, call in re.findall(r"self\.(\w+)(\()?", body):
            if call:
                assert name in fns, "%s: %s::%s() is not in %s" % (f, typ, name, HOME[typ])
            else:
                assert name in fields, "%s: %s.%s is not a field in %s (it has %s)" % (f, typ, name, HOME[typ], sorted(fields))
            checked += 1
    assert checked >= 40


def test_field_types_are_what_the_flat_values_take():
    """The additions copy fields into FlatShape / FlatMaterial / FlatPattern by value or by reference:
    the reference's field types must allow exactly that."""
    want = {
        ("Cylinder", "minimum"): "f64", ("Cylinder", "maximum"): "f64", ("Cylinder", "closed"): "bool",
        ("Cone", "minimum"): "f64", ("Cone", "maximum"): "f64", ("Cone", "closed"): "bool",
        ("Triangle", "p1"): "Tuple", ("Triangle", "e1"): "Tuple", ("Triangle", "e2"): "Tuple", ("Triangle", "normal"): "Tuple",
        ("SmoothTriangle", "p1"): "Tuple", ("SmoothTriangle", "e1"): "Tuple", ("SmoothTriangle", "e2"): "Tuple",
        ("SmoothTriangle", "n1"): "Tuple", ("SmoothTriangle", "n2"): "Tuple", ("SmoothTriangle", "n3"): "Tuple",
        ("Group", "children"): "RwLock<Vec<Arc<dyn Shape>>>",          # .read().unwrap().clone()
        ("CSG", "operation"): "Operation", ("CSG", "left"): "Arc<dyn Shape>", ("CSG", "right"): "Arc<dyn Shape>",
        ("Phong", "pattern"): "Box<dyn Pattern>",                       # .as_ref() -> &dyn Pattern
        ("Phong", "ambient"): "f64", ("Phong", "diffuse"): "f64", ("Phong", "specular"): "f64", ("Phong", "shininess"): "f64",
        ("Phong", "reflective"): "f64", ("Phong", "transparency"): "f64", ("Phong", "refractive_index"): "f64",
        ("Solid", "color"): "Color",
        ("Striped", "former"): "Box<dyn Pattern>", ("Striped", "latter"): "Box<dyn Pattern>",
        ("Gradient", "former"): "Box<dyn Pattern>", ("Gradient", "latter"): "Box<dyn Pattern>",
        ("Ring", "former"): "Box<dyn Pattern>", ("Ring", "latter"): "Box<dyn Pattern>",
        ("Checker", "former"): "Box<dyn Pattern>", ("Checker", "latter"): "Box<dyn Pattern>",
        ("Blended", "former"): "Box<dyn Pattern>", ("Blended", "latter"): "Box<dyn Pattern>",
        ("Perturbed", "delegate"): "Box<dyn Pattern>", ("Perturbed", "scale"): "f64",
        ("World", "objects"): "Vec<Arc<dyn Shape>>", ("World", "lights"): "Vec<Arc<PointLight>>",
        ("Camera", "half_width"): "f64", ("Camera", "half_height"): "f64", ("Camera", "pixel_size"): "f64",
        ("Camera", "transform"): "Matrix",
    }
    for (typ, field), ty in want.items():
        got = struct_fields(ref(HOME[typ]), typ)
        assert got.get(field) == ty, (typ, field, got.get(field))
    # fields that are moved out of `&self` must be Copy
    for path, name in (("src/tuples/tuple.rs", "Tuple"), ("src/tuples/color.rs", "Color")):
        m = re.search(r"#\[derive\(([^)]*)\)\]\s*pub struct %s\b" % name, ref(path))
        assert m and "Copy" in m.group(1) and "Clone" in m.group(1), name
    # the three variants the CSG addition matches on (src/geometry/csg.rs:27-31)
    m = re.search(r"pub enum Operation\s*\{(.*?)\}", ref("src/geometry/csg.rs"), flags=re.S)
    assert [v.strip() for v in m.group(1).split(",") if v.strip()] == ["Difference", "Intersection", "Union"]
    for v in ("Difference", "Intersection", "Union"):
        assert "Operation::%s" % v in rust("additions/geometry.rs")


def test_traits_have_the_getters_flatten_calls():
    flatten = rust("flatten.rs")
    shape = ref("src/geometry/shape.rs")
    body = shape[shape.index("pub trait Shape"):]
    for sig, use in ((r"fn get_transform\(&self\) -> Arc<Matrix>;", "shape.get_transform()"),
                     (r"fn get_material\(&self\) -> Arc<dyn Material>;", "shape.get_material()"),
                     (r"fn casts_shadow\(&self\) -> bool;", "shape.casts_shadow()"),
                     (r"fn bounds\(&self\) -> BoundingBox;", "shape.bounds()")):
        assert re.search(sig, body), sig
        assert use in flatten, use
    pattern = ref("src/patterns/pattern.rs")
    assert re.search(r"pub trait Pattern: Sync \+ Send", pattern)
    assert re.search(r"fn get_transform\(&self\) -> Arc<Matrix>;", pattern) and "p.get_transform()" in flatten
    assert re.search(r"pub trait Material: Send \+ Sync", ref("src/materials/material.rs"))
    assert re.search(r"pub trait Shape: Sync \+ Send", shape)
    # the new trait method is named `flat` on all three and is not taken yet
    traits = rust("additions/traits.rs")
    for path, ret in (("src/geometry/shape.rs", "FlatShape"), ("src/materials/material.rs", "FlatMaterial<'_>"),
                      ("src/patterns/pattern.rs", "FlatPattern<'_>")):
        assert "fn flat(&self) -> crate::flatten::%s;" % ret in traits
        assert "flat" not in methods(ref(path)) and "flat_node" not in methods(ref(path))
    for call in ("shape.flat()", "m.flat()", "p.flat()"):
        assert call in flatten


def test_other_reference_items_flatten_uses():
    flatten = rust("flatten.rs")
    uses = {
        "transform.inverse()": ("src/matrices/matrix.rs", r"pub fn inverse\(&self\) -> Result<Matrix, &'static str>"),
        "inv.get(r, c)": ("src/matrices/matrix.rs", r"pub fn get\(&self, row: usize, column: usize\) -> Option<&f64>"),
        "b.min()": ("src/tuples/bounding_box.rs", r"pub fn min\(&self\) -> Tuple"),
        "b.max()": ("src/tuples/bounding_box.rs", r"pub fn max\(&self\) -> Tuple"),
        "Canvas::new(w, h)": ("src/window/canvas.rs", r"pub fn new\(width: usize, height: usize\) -> Canvas"),
        "canvas.write_pixel(x, y, Color::new(": ("src/window/canvas.rs", r"pub fn write_pixel\(&mut self, x: usize, y: usize, color: Color\) -> Result"),
        "Color::new(rgb[i], rgb[i + 1], rgb[i + 2])": ("src/tuples/color.rs", r"pub fn new\(red: f64, green: f64, blue: f64\) -> Color"),
        "camera.width()": ("src/scene/camera.rs", r"pub fn width\(&self\) -> usize"),
        "camera.height()": ("src/scene/camera.rs", r"pub fn height\(&self\) -> usize"),
        "l.position": ("src/tuples/point_light.rs", r"pub position: Tuple"),
        "l.intensity.red": ("src/tuples/color.rs", r"pub red: f64"),
        "l.intensity.green": ("src/tuples/color.rs", r"pub green: f64"),
        "l.intensity.blue": ("src/tuples/color.rs", r"pub blue: f64"),
        "[t.x, t.y, t.z]": ("src/tuples/tuple.rs", r"pub x: f64,\s*pub y: f64,\s*pub z: f64"),
        "crate::MAX_RAY_RECURSION_DEPTH": ("src/lib.rs", r"static MAX_RAY_RECURSION_DEPTH: usize = 5;"),
    }
    for use, (path, sig) in uses.items():
        assert use in flatten, use
        assert re.search(sig, ref(path)), (path, sig)
    # the replacement keeps the signature of the function it replaces (src/lib.rs:77)
    sig = "pub fn render(world: Arc<World>, camera: Arc<Camera>) -> Canvas"
    assert sig in ref("src/lib.rs") and sig in flatten
    # every module path flatten.rs imports exists in the crate
    for mod in re.findall(r"^use crate::([\w:]+)::\w+;", flatten, flags=re.M):
        if mod == "ffi":
            continue
        assert os.path.exists(os.path.join(REF, "src", mod.replace("::", "/") + ".rs")), mod
