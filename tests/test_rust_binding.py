"""rust/src/ffi.rs must mirror include/rt_abi.h: the Rust toolchain is not available where this
library is built, so the binding is checked as text -- every struct with the same fields in the same
order with matching types, every function with the same name, arity and argument types, every
constant with the same value -- and rust/src/flatten.rs must fill every field of every table."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_TO_RUST = {
    "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "uint8_t": "u8", "int": "c_int", "float": "f32",
    "double": "f64", "char": "c_char", "void": "c_void",
}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def c_type_to_rust(ctype):
    ctype = ctype.strip()
    const = ctype.startswith("const ")
    core = ctype.replace("const ", "").strip()
    stars = core.count("*")
    core = core.replace("*", "").strip()
    core = re.sub(r"^struct ", "", core)
    base = C_TO_RUST.get(core, core)  # rt_* struct names are the same on both sides
    for _ in range(stars):
        base = ("*const " if const else "*mut ") + base
        const = False  # only the innermost pointee is const in this header
    return base


def parse_header():
    text = strip_comments(open(os.path.join(ROOT, "include", "rt_abi.h")).read())
    structs = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # "double min[3]" / "const rt_node* nodes" / "int32_t a, b" / "double x, y"
            mm = re.match(r"(.*?[\w\*])\s+((?:\**\w+(?:\[\d+\])?\s*,?\s*)+)$", decl)
            ctype, names = mm.group(1), mm.group(2)
            # a star written next to the type belongs to the type
            for name in [n.strip() for n in names.split(",")]:
                arr = re.match(r"(\w+)\[(\d+)\]", name)
                if arr:
                    fields.append((arr.group(1), "[%s; %s]" % (c_type_to_rust(ctype), arr.group(2))))
                else:
                    fields.append((name, c_type_to_rust(ctype)))
        structs[m.group(1)] = fields
    funcs = {}
    body = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", text, flags=re.S)
    body = re.sub(r"typedef enum \w+ \{.*?\} \w+;", "", body, flags=re.S)
    for m in re.finditer(r"([\w\* ]+?)\b(rt_\w+)\s*\(([^)]*)\)\s*;", body):
        ret = m.group(1).strip()
        args = []
        if m.group(3).strip() != "void":
            for a in m.group(3).split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?[\w\*])\s+(\w+)$", a)
                args.append(c_type_to_rust(mm.group(1)))
        funcs[m.group(2)] = (None if ret == "void" else c_type_to_rust(ret), args)
    consts = {m.group(1): eval(m.group(2)) for m in re.finditer(r"#define (RT_[A-Z_]+) (\(?[0-9 <]+\)?)", text)}
    for m in re.finditer(r"enum \{(.*?)\};", text, flags=re.S):
        value = -1
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, expr = [s.strip() for s in item.split("=")]
                value = eval(expr)
            else:
                name, value = item, value + 1
            consts[name] = value
    status = re.search(r"typedef enum rt_status \{(.*?)\}", text, flags=re.S).group(1)
    value = -1
    for item in status.split(","):
        item = item.strip()
        if item:
            if "=" in item:
                name, expr = [s.strip() for s in item.split("=")]
                value = int(expr)
            else:
                name, value = item, value + 1
            consts[name] = value
    return structs, funcs, consts


def parse_rust():
    text = strip_comments(open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read())
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct (\w+) \{(.*?)\}", text, flags=re.S):
        fields = []
        for f in m.group(2).split(","):
            f = " ".join(f.split())
            if f:
                name, typ = f.replace("pub ", "", 1).split(":", 1)
                fields.append((name.strip().replace("r#", ""), typ.strip()))
        structs[m.group(1)] = fields
    not_repr_c = set(re.findall(r"pub struct (\w+)", text)) - set(structs)
    funcs = {}
    ext = re.search(r'extern "C" \{(.*)\}', text, flags=re.S).group(1)
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", ext, flags=re.S):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            if a:
                args.append(a.split(":", 1)[1].strip())
        funcs[m.group(1)] = (m.group(3).strip() if m.group(3) else None, args)
    consts = {m.group(1): eval(m.group(2)) for m in re.finditer(r"pub const (RT_\w+): \w+ = ([^;]+);", text)}
    return structs, funcs, consts, not_repr_c


def test_structs_match_field_by_field():
    c_structs, _, _ = parse_header()
    r_structs, _, _, not_repr_c = parse_rust()
    assert not not_repr_c, "structs without #[repr(C)]: %s" % not_repr_c
    assert len(c_structs) >= 15
    assert set(c_structs) == set(r_structs)
    for name, fields in c_structs.items():
        assert r_structs[name] == fields, name


def test_every_function_is_declared_with_the_same_signature():
    _, c_funcs, _ = parse_header()
    _, r_funcs, _, _ = parse_rust()
    assert len(c_funcs) >= 26
    assert set(c_funcs) == set(r_funcs)
    for name, sig in c_funcs.items():
        assert r_funcs[name] == sig, name


def test_constants_match():
    _, _, c_consts = parse_header()
    _, _, r_consts, _ = parse_rust()
    assert len(c_consts) >= 35
    assert c_consts == r_consts


def test_struct_sizes_follow_from_the_fields():
    """Sizes implied by the Rust declarations (natural alignment, repr(C)) equal the ctypes sizes."""
    import ctypes as C
    from raytracer_amd import abi
    size = {"i32": 4, "u32": 4, "f32": 4, "i64": 8, "u64": 8, "f64": 8, "u8": 1, "c_int": 4}
    r_structs, _, _, _ = parse_rust()
    for name, fields in r_structs.items():
        off, align = 0, 1
        for _, typ in fields:
            arr = re.match(r"\[(\w+); (\d+)\]", typ)
            if arr:
                s, n = size[arr.group(1)], int(arr.group(2))
            elif typ.startswith("*"):
                s, n = 8, 1
            else:
                s, n = size[typ], 1
            off = (off + s - 1) // s * s + s * n
            align = max(align, s)
        total = (off + align - 1) // align * align
        assert total == C.sizeof(getattr(abi, name)), name


def test_flatten_fills_every_table_and_field():
    """rust/src/flatten.rs builds every table of rt_scene_desc and names every field of the structs it
    constructs; the per-module additions cover all nine shape types, Phong and all eight patterns."""
    text = strip_comments(open(os.path.join(ROOT, "rust", "src", "flatten.rs")).read())
    c_structs, _, _ = parse_header()
    for name in ("rt_node", "rt_xform", "rt_aabb", "rt_range", "rt_tri", "rt_tri_normals", "rt_material", "rt_pattern",
                 "rt_light", "rt_scene_desc", "rt_camera", "rt_params"):
        lits = re.findall(name + r" \{(.*?)\}", text, flags=re.S)
        assert lits, "flatten.rs never builds an " + name
        for field, _ in c_structs[name]:
            assert any(re.search(r"\b(?:r#)?%s\b" % field, lit) for lit in lits), (name, field)
    geo = open(os.path.join(ROOT, "rust", "src", "additions", "geometry.rs")).read()
    for t in ("Sphere", "Plane", "Cube", "Cylinder", "Cone", "Triangle", "SmoothTriangle", "Group", "CSG"):
        assert re.search(r"impl %s \{\s*(//[^\n]*\s*)*pub\(crate\) fn flat_node" % t, geo), t
    mp = open(os.path.join(ROOT, "rust", "src", "additions", "materials_patterns.rs")).read()
    for t in ("Phong", "Solid", "Striped", "Gradient", "Ring", "Checker", "Blended", "Perturbed", "TestPattern"):
        assert re.search(r"impl %s \{\s*pub\(crate\) fn flat_node" % t, mp), t
    for fn in ("rt_multi_create", "rt_render_multi", "rt_scene_create", "rt_render_ppm", "rt_last_error"):
        assert fn in text
    build = open(os.path.join(ROOT, "rust", "build.rs")).read()
    assert "-ffp-contract=off" in build and "--offload-arch=" in build and "rt_multi.hip" in build
