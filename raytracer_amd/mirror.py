"""Host-side mirror of the reference's scene-construction interface.

The reference builds a scene from `Matrix`, pattern, `Phong`, shape, `Group`, `PointLight`, `World`
and `Camera` constructors (src/lib.rs:135-323 shows the idiom).  This module offers the same
constructors, with the same argument order and meaning, on top of a native library that exports a
handle-based C builder interface.  The native library does all arithmetic (matrix products, cofactor
inverses, bounding boxes, `divide`), so that every f64 result is produced by restated reference
arithmetic and not by Python.

`Api(lib, prefix)` binds one such library.  The shipped package binds the host library
(`librt_host.so`, prefix `rth_`); the test-only oracle binds its own library with prefix `orc_`.
Both export the same builder entry points, which is what lets one scene description be built twice.
"""
import ctypes as C
import math

c_d = C.c_double
c_dp = C.POINTER(C.c_double)
c_i64 = C.c_int64


def _arr(seq):
    seq = [float(v) for v in seq]
    return (c_d * len(seq))(*seq)


def point(x, y, z):
    """Tuple::point (src/tuples/tuple.rs:33-35)."""
    return (float(x), float(y), float(z), 1.0)


def vector(x, y, z):
    """Tuple::vector (src/tuples/tuple.rs:25-27)."""
    return (float(x), float(y), float(z), 0.0)


class Color(tuple):
    """Color (src/tuples/color.rs:6-36)."""

    def __new__(cls, r, g, b):
        return super().__new__(cls, (float(r), float(g), float(b)))

    @staticmethod
    def white():
        return Color(1.0, 1.0, 1.0)

    @staticmethod
    def black():
        return Color(0.0, 0.0, 0.0)

    @staticmethod
    def red():
        return Color(1.0, 0.0, 0.0)

    @staticmethod
    def green():
        return Color(0.0, 1.0, 0.0)

    @staticmethod
    def blue():
        return Color(0.0, 0.0, 1.0)


class Ray:
    """Ray (src/tuples/ray.rs:6-22)."""

    def __init__(self, origin, direction):
        self.origin = tuple(origin)
        self.direction = tuple(direction)


class PointLight:
    """PointLight (src/tuples/point_light.rs:9-19)."""

    def __init__(self, position, intensity):
        self.position = tuple(position)
        self.intensity = tuple(intensity)


# Shape / pattern kind codes shared by every native library behind this module.
GROUP, SPHERE, PLANE, CUBE, CYLINDER, CONE, TRIANGLE, SMOOTH_TRIANGLE, CSG_SHAPE = range(9)
DIFFERENCE, INTERSECTION, UNION = range(3)  # Operation (src/geometry/csg.rs:27-31)
P_SOLID, P_STRIPED, P_GRADIENT, P_RING, P_CHECKER, P_BLENDED, P_PERTURBED, P_TEST = range(8)


class Api:
    """Scene-construction interface bound to one native library."""

    def __init__(self, lib, prefix):
        self.lib = lib
        self.prefix = prefix
        self._bind()
        api = self

        class Matrix:
            """Matrix (src/matrices/matrix.rs); 4x4, row-major."""

            def __init__(self, values):
                self.v = tuple(float(x) for x in values)
                assert len(self.v) == 16

            @staticmethod
            def from_rows(rows):
                return Matrix([x for r in rows for x in r])

            @staticmethod
            def identity(size=4):
                return api._mat("mat_identity")

            @staticmethod
            def translation(x, y, z):
                return api._mat("mat_translation", x, y, z)

            @staticmethod
            def scaling(x, y, z):
                return api._mat("mat_scaling", x, y, z)

            @staticmethod
            def reflect_x():
                return Matrix.scaling(-1.0, 1.0, 1.0)

            @staticmethod
            def reflect_y():
                return Matrix.scaling(1.0, -1.0, 1.0)

            @staticmethod
            def reflect_z():
                return Matrix.scaling(1.0, 1.0, -1.0)

            @staticmethod
            def rotation_x(r):
                return api._mat("mat_rotation_x", r)

            @staticmethod
            def rotation_y(r):
                return api._mat("mat_rotation_y", r)

            @staticmethod
            def rotation_z(r):
                return api._mat("mat_rotation_z", r)

            @staticmethod
            def shearing(a, b, c, d, e, f):
                return api._mat("mat_shearing", a, b, c, d, e, f)

            @staticmethod
            def view_transform(frm, to, up):
                out = (c_d * 16)()
                api._f("mat_view_transform")(_arr(frm[:3]), _arr(to[:3]), _arr(up[:3]), out)
                return Matrix(out)

            def get(self, r, c):
                return self.v[r * 4 + c]

            def __mul__(self, other):
                if isinstance(other, Matrix):
                    out = (c_d * 16)()
                    api._f("mat_mul")(_arr(self.v), _arr(other.v), out)
                    return Matrix(out)
                out = (c_d * 4)()
                api._f("mat_mul_tuple")(_arr(self.v), _arr(other), out)
                return tuple(out)

            def inverse(self):
                out = (c_d * 16)()
                if api._f("mat_inverse")(_arr(self.v), out) != 0:
                    raise ValueError("Error: Matrix is not invertible due to a zero determinant")
                return Matrix(out)

            def transpose(self):
                out = (c_d * 16)()
                api._f("mat_transpose")(_arr(self.v), out)
                return Matrix(out)

            def rows(self):
                return [list(self.v[i * 4:i * 4 + 4]) for i in range(4)]

        class Pattern:
            def __init__(self, h, kind, transform=None):
                self.h = h
                self.kind = kind
                self.transform = transform

        def Solid(color=Color.white()):
            """Solid::new (src/patterns/solid.rs:12-14)."""
            return Pattern(api._f("pattern_solid")(*[float(c) for c in color]), P_SOLID)

        def _binary(kind):
            def make(former=None, latter=None, transform=None):
                former = former or Solid(Color.white())
                latter = latter or Solid(Color.black())
                transform = transform or Matrix.identity()
                h = api._f("pattern_binary")(kind, former.h, latter.h, _arr(transform.v))
                return Pattern(h, kind, transform)
            return make

        def Perturbed(delegate=None, scale=0.2, transform=None):
            """Perturbed::new (src/patterns/perturbed.rs:43-49)."""
            delegate = delegate or Solid(Color.white())
            transform = transform or Matrix.identity()
            h = api._f("pattern_perturbed")(delegate.h, float(scale), _arr(transform.v))
            return Pattern(h, P_PERTURBED, transform)

        def TestPattern(transform=None):
            """TestPattern::new (src/patterns/test_pattern.rs:12-14)."""
            transform = transform or Matrix.identity()
            return Pattern(api._f("pattern_test")(_arr(transform.v)), P_TEST, transform)

        class Phong:
            """Phong::new (src/materials/phong.rs:22-42)."""

            def __init__(self, pattern, ambient, diffuse, specular, shininess, reflective,
                         transparency, refractive_index):
                self.pattern = pattern
                self.ambient, self.diffuse, self.specular = ambient, diffuse, specular
                self.shininess, self.reflective = shininess, reflective
                self.transparency, self.refractive_index = transparency, refractive_index
                self.h = api._f("phong")(pattern.h, float(ambient), float(diffuse), float(specular),
                                         float(shininess), float(reflective), float(transparency),
                                         float(refractive_index))

            @staticmethod
            def default():
                """Phong::default (src/materials/phong.rs:44-55)."""
                return Phong(Solid(Color.white()), 0.1, 0.9, 0.9, 200.0, 0.0, 0.0, 1.0)

        class Shape:
            def __init__(self, h, kind, transform, material, casts_shadow):
                self.h = h
                self.kind = kind
                self.transform = transform
                self.material = material
                self.casts_shadow = casts_shadow

            def bounds(self):
                out = (c_d * 6)()
                api._f("bounds")(self.h, out)
                return tuple(out[:3]), tuple(out[3:])

            def parent_space_bounds_of(self):
                out = (c_d * 6)()
                api._f("parent_space_bounds")(self.h, out)
                return tuple(out[:3]), tuple(out[3:])

            def divide(self, threshold):
                api._f("divide")(self.h, int(threshold))

            def num_of_children(self):
                return api._f("group_num_children")(self.h) if self.kind in (GROUP, CSG_SHAPE) else 0

            def left(self):
                """CSG::left (src/geometry/csg.rs:66-68)."""
                return self.get_child(0)

            def right(self):
                """CSG::right (src/geometry/csg.rs:70-72)."""
                return self.get_child(1)

            def filter_intersections(self, intersections):
                """CSG::filter_intersections (src/geometry/csg.rs:80-107) on [(t, shape), ...]."""
                n = len(intersections)
                ts = (c_d * max(n, 1))(*[float(t) for t, _ in intersections])
                objs = (c_i64 * max(n, 1))(*[o.h for _, o in intersections])
                k = api._f("csg_filter")(self.h, n, ts, objs)
                return [(ts[i], objs[i]) for i in range(k)]

            def get_child(self, i):
                h = api._f("group_child")(self.h, int(i))
                return Shape(h, api._f("shape_kind")(h), None, None, True)

            def add_child(self, child):
                if api._f("group_add_child")(self.h, child.h) != 0:
                    raise ValueError("add_child on a non-group shape")

            def add_children(self, children):
                for c in children:
                    self.add_child(c)

            def triangle_data(self):
                out = (c_d * 27)()
                api._f("triangle_data")(self.h, out)
                v = [tuple(out[i * 3:i * 3 + 3]) for i in range(9)]
                keys = ["p1", "p2", "p3", "e1", "e2", "normal", "n1", "n2", "n3"]
                return dict(zip(keys, v))

        def _simple(kind):
            def make(transform=None, material=None, casts_shadow=True):
                transform = transform or Matrix.identity()
                material = material or Phong.default()
                h = api._f("shape")(kind, _arr(transform.v), material.h, int(casts_shadow),
                                    -math.inf, math.inf, 0)
                return Shape(h, kind, transform, material, casts_shadow)
            return make

        def _quadric(kind):
            def make(transform=None, material=None, casts_shadow=True, minimum=-math.inf,
                     maximum=math.inf, closed=False):
                transform = transform or Matrix.identity()
                material = material or Phong.default()
                h = api._f("shape")(kind, _arr(transform.v), material.h, int(casts_shadow),
                                    float(minimum), float(maximum), int(closed))
                s = Shape(h, kind, transform, material, casts_shadow)
                s.minimum, s.maximum, s.closed = minimum, maximum, closed
                return s
            return make

        def Triangle(p1, p2, p3, transform=None, material=None, casts_shadow=True):
            """Triangle::new (src/geometry/triangle.rs:34-63)."""
            transform = transform or Matrix.identity()
            material = material or Phong.default()
            pts = list(p1[:3]) + list(p2[:3]) + list(p3[:3])
            h = api._f("triangle")(_arr(pts), _arr(transform.v), material.h, int(casts_shadow))
            return Shape(h, TRIANGLE, transform, material, casts_shadow)

        def SmoothTriangle(p1, p2, p3, n1, n2, n3, transform=None, material=None, casts_shadow=True):
            """SmoothTriangle::new (src/geometry/smooth_triangle.rs:34-70)."""
            transform = transform or Matrix.identity()
            material = material or Phong.default()
            pts = list(p1[:3]) + list(p2[:3]) + list(p3[:3])
            nrm = list(n1[:3]) + list(n2[:3]) + list(n3[:3])
            h = api._f("smooth_triangle")(_arr(pts), _arr(nrm), _arr(transform.v), material.h,
                                          int(casts_shadow))
            return Shape(h, SMOOTH_TRIANGLE, transform, material, casts_shadow)

        def Group(transform=None):
            """Group::new / Group::default (src/geometry/group.rs:26-48)."""
            transform = transform or Matrix.identity()
            return Shape(api._f("group")(_arr(transform.v)), GROUP, transform, None, True)

        def CSG(operation, left, right, transform=None, material=None):
            """CSG::new / CSG::default (src/geometry/csg.rs:34-64); casts_shadow is false there."""
            transform = transform or Matrix.identity()
            material = material or Phong.default()
            h = api._f("csg")(int(operation), left.h, right.h, _arr(transform.v), material.h)
            return Shape(h, CSG_SHAPE, transform, material, False)

        def intersection_allowed(operation, lhit, inl, inr):
            """CSG::intersection_allowed (src/geometry/csg.rs:113-140)."""
            return bool(api._f("csg_allowed")(int(operation), int(lhit), int(inl), int(inr)))

        class ObjFileParser:
            """ObjFileParser (src/scene/obj_file_parser.rs:23-235)."""

            def __init__(self, h):
                self.h = h

            @staticmethod
            def parse_obj_file(file_path, transform=None, material=None, casts_shadow=True):
                transform = transform or Matrix.identity()
                material = material or Phong.default()
                h = api._f("obj_parse")(str(file_path).encode(), _arr(transform.v), material.h,
                                        int(casts_shadow))
                if h < 0:
                    raise IOError("Error: Could not read lines from: %s" % file_path)
                return ObjFileParser(h)

            @property
            def vertices(self):
                out = (c_d * 4)()
                res = []
                for i in range(api._f("obj_num_vertices")(self.h)):
                    api._f("obj_vertex")(self.h, i, out)
                    res.append(tuple(out))
                return res

            @property
            def vertex_normals(self):
                out = (c_d * 4)()
                res = []
                for i in range(api._f("obj_num_normals")(self.h)):
                    api._f("obj_normal")(self.h, i, out)
                    res.append(tuple(out))
                return res

            def num_groups(self):
                return api._f("obj_num_groups")(self.h)

            def group(self, name=None):
                h = api._f("obj_group")(self.h, None if name is None else name.encode())
                if h < 0:
                    return None
                return Shape(h, GROUP, None, None, True)

            def obj_to_group(self, transform=None):
                transform = transform or Matrix.identity()
                return Shape(api._f("obj_to_group")(self.h, _arr(transform.v)), GROUP, transform,
                             None, True)

        class World:
            """World::new (src/scene/world.rs:21-23)."""

            def __init__(self, objects, lights):
                self.objects = list(objects)
                self.lights = list(lights)
                self.h = api._f("world")()
                for o in self.objects:
                    api._f("world_add_object")(self.h, o.h)
                for l in self.lights:
                    api._f("world_add_light")(self.h, _arr(l.position[:3]), _arr(l.intensity))

        class Camera:
            """Camera::new (src/scene/camera.rs:17-43)."""

            def __init__(self, hsize, vsize, field_of_view, transform):
                self.hsize, self.vsize = int(hsize), int(vsize)
                self.field_of_view = field_of_view
                self.transform = transform
                self.h = api._f("camera")(self.hsize, self.vsize, float(field_of_view),
                                          _arr(transform.v))
                info = (c_d * 5)()
                api._f("camera_info")(self.h, info)
                self.half_width, self.half_height, self.pixel_size = info[0], info[1], info[2]

            def height(self):
                return self.hsize  # src/scene/camera.rs:68-70

            def width(self):
                return self.vsize  # src/scene/camera.rs:72-74

        self.Matrix = Matrix
        self.Pattern = Pattern
        self.Solid = Solid
        self.Striped = _binary(P_STRIPED)
        self.Gradient = _binary(P_GRADIENT)
        self.Ring = _binary(P_RING)
        self.Checker = _binary(P_CHECKER)
        self.Blended = _binary(P_BLENDED)
        self.Perturbed = Perturbed
        self.TestPattern = TestPattern
        self.Phong = Phong
        self.Shape = Shape
        self.Sphere = _simple(SPHERE)
        self.Plane = _simple(PLANE)
        self.Cube = _simple(CUBE)
        self.Cylinder = _quadric(CYLINDER)
        self.Cone = _quadric(CONE)
        self.Triangle = Triangle
        self.SmoothTriangle = SmoothTriangle
        self.Group = Group
        self.CSG = CSG
        self.intersection_allowed = intersection_allowed
        self.DIFFERENCE, self.INTERSECTION, self.UNION = DIFFERENCE, INTERSECTION, UNION
        self.ObjFileParser = ObjFileParser
        self.World = World
        self.Camera = Camera
        self.PointLight = PointLight
        self.Color = Color
        self.Ray = Ray
        self.point = point
        self.vector = vector

    # -- plumbing -------------------------------------------------------------------------------
    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def _mat(self, name, *args):
        out = (c_d * 16)()
        self._f(name)(*[float(a) for a in args], out)
        return self.Matrix(out)

    def reset(self):
        """Drop every handle created through this library."""
        self._f("reset")()

    def _bind(self):
        sig = {
            "reset": (None, []),
            "mat_identity": (None, [c_dp]),
            "mat_translation": (None, [c_d, c_d, c_d, c_dp]),
            "mat_scaling": (None, [c_d, c_d, c_d, c_dp]),
            "mat_rotation_x": (None, [c_d, c_dp]),
            "mat_rotation_y": (None, [c_d, c_dp]),
            "mat_rotation_z": (None, [c_d, c_dp]),
            "mat_shearing": (None, [c_d] * 6 + [c_dp]),
            "mat_view_transform": (None, [c_dp, c_dp, c_dp, c_dp]),
            "mat_mul": (None, [c_dp, c_dp, c_dp]),
            "mat_inverse": (C.c_int, [c_dp, c_dp]),
            "mat_transpose": (None, [c_dp, c_dp]),
            "mat_mul_tuple": (None, [c_dp, c_dp, c_dp]),
            "pattern_solid": (c_i64, [c_d, c_d, c_d]),
            "pattern_binary": (c_i64, [C.c_int, c_i64, c_i64, c_dp]),
            "pattern_perturbed": (c_i64, [c_i64, c_d, c_dp]),
            "pattern_test": (c_i64, [c_dp]),
            "phong": (c_i64, [c_i64] + [c_d] * 7),
            "shape": (c_i64, [C.c_int, c_dp, c_i64, C.c_int, c_d, c_d, C.c_int]),
            "triangle": (c_i64, [c_dp, c_dp, c_i64, C.c_int]),
            "smooth_triangle": (c_i64, [c_dp, c_dp, c_dp, c_i64, C.c_int]),
            "group": (c_i64, [c_dp]),
            "group_add_child": (C.c_int, [c_i64, c_i64]),
            "csg": (c_i64, [C.c_int, c_i64, c_i64, c_dp, c_i64]),
            "csg_allowed": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
            "csg_filter": (C.c_int, [c_i64, C.c_int, c_dp, C.POINTER(c_i64)]),
            "shape_kind": (C.c_int, [c_i64]),
            "group_num_children": (C.c_int, [c_i64]),
            "group_child": (c_i64, [c_i64, C.c_int]),
            "divide": (None, [c_i64, C.c_int]),
            "bounds": (None, [c_i64, c_dp]),
            "parent_space_bounds": (None, [c_i64, c_dp]),
            "triangle_data": (None, [c_i64, c_dp]),
            "obj_parse": (c_i64, [C.c_char_p, c_dp, c_i64, C.c_int]),
            "obj_num_vertices": (C.c_int, [c_i64]),
            "obj_num_normals": (C.c_int, [c_i64]),
            "obj_vertex": (None, [c_i64, C.c_int, c_dp]),
            "obj_normal": (None, [c_i64, C.c_int, c_dp]),
            "obj_num_groups": (C.c_int, [c_i64]),
            "obj_group": (c_i64, [c_i64, C.c_char_p]),
            "obj_to_group": (c_i64, [c_i64, c_dp]),
            "world": (c_i64, []),
            "world_add_object": (None, [c_i64, c_i64]),
            "world_add_light": (None, [c_i64, c_dp, c_dp]),
            "camera": (c_i64, [c_i64, c_i64, c_d, c_dp]),
            "camera_info": (None, [c_i64, c_dp]),
        }
        for name, (res, args) in sig.items():
            fn = self._f(name)
            fn.restype = res
            fn.argtypes = args
