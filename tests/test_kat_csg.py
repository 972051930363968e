"""Known answers of src/geometry/csg.rs, once per library where both restate the function (the
truth tables, filter_intersections, bounds, children) and on the oracle alone for local_intersect
(the shipped host library has no CPU intersection code: that is the GPU's job, see
tests/test_gpu_csg.py)."""
import pytest


@pytest.fixture(params=["host", "oracle"])
def lib(request, host, orc):
    return host if request.param == "host" else orc


def unit_sphere(o, transform=None):
    return o.Sphere(transform or o.Matrix.identity(), o.Phong.default(), True)


def unit_cube(o):
    return o.Cube(o.Matrix.identity(), o.Phong.default(), True)


def test_csg_children_and_shadow_flag(lib):  # csg.rs:247-266, 34-59
    s1, s2 = unit_sphere(lib), unit_cube(lib)
    c = lib.CSG(lib.UNION, s1, s2)
    assert c.num_of_children() == 2
    assert c.left().h == s1.h and c.right().h == s2.h
    assert c.casts_shadow is False


def test_csg_bounds_contain_children(lib):  # csg.rs:268-288
    left = unit_sphere(lib)
    right = unit_sphere(lib, lib.Matrix.translation(2.0, 3.0, 4.0))
    c = lib.CSG(lib.DIFFERENCE, left, right)
    assert c.bounds() == ((-1.0, -1.0, -1.0), (3.0, 4.0, 5.0))


UNION_TABLE = [(True, True, True, False), (True, True, False, True), (True, False, True, False),
               (True, False, False, True), (False, True, True, False), (False, True, False, False),
               (False, False, True, True), (False, False, False, True)]               # csg.rs:290-310
INTERSECTION_TABLE = [(True, True, True, True), (True, True, False, False), (True, False, True, True),
                      (True, False, False, False), (False, True, True, True), (False, True, False, True),
                      (False, False, True, False), (False, False, False, False)]      # csg.rs:312-335
DIFFERENCE_TABLE = [(True, True, True, False), (True, True, False, True), (True, False, True, False),
                    (True, False, False, True), (False, True, True, True), (False, True, False, True),
                    (False, False, True, False), (False, False, False, False)]        # csg.rs:337-360


def test_intersection_allowed_truth_tables(lib):
    for op, table in ((lib.UNION, UNION_TABLE), (lib.INTERSECTION, INTERSECTION_TABLE),
                      (lib.DIFFERENCE, DIFFERENCE_TABLE)):
        for lhit, inl, inr, expected in table:
            assert lib.intersection_allowed(op, lhit, inl, inr) is expected, (op, lhit, inl, inr)


def test_filter_intersections(lib):  # csg.rs:362-396
    for op, x0, x1 in ((lib.UNION, 0, 3), (lib.INTERSECTION, 1, 2), (lib.DIFFERENCE, 0, 1)):
        s1, s2 = unit_sphere(lib), unit_cube(lib)
        xs = [(1.0, s1), (2.0, s2), (3.0, s1), (4.0, s2)]
        c = lib.CSG(op, s1, s2)
        got = c.filter_intersections(xs)
        assert got == [(xs[x0][0], xs[x0][1].h), (xs[x1][0], xs[x1][1].h)]


def test_filter_treats_nested_operands_as_left(lib):
    """`includes` descends through groups and CSG shapes (group.rs:172-182, csg.rs:177-179)."""
    a, b, c3 = unit_sphere(lib), unit_cube(lib), unit_sphere(lib)
    g = lib.Group()
    g.add_child(a)
    inner = lib.CSG(lib.UNION, g, b)
    outer = lib.CSG(lib.DIFFERENCE, inner, c3)
    # a and b both count as "left" of outer: 1.0 (left, outside right) kept, 2.0 (right, inside left) kept,
    # 3.0 (left, inside right) dropped, 4.0 (right, inside left: inl toggled twice -> outside) dropped
    got = outer.filter_intersections([(1.0, a), (2.0, c3), (3.0, b), (4.0, c3)])
    assert got == [(1.0, a.h), (2.0, c3.h)]


def test_divide_reaches_the_operands(lib):  # csg.rs:216-219
    g = lib.Group()
    for x in (-2.0, 2.0):
        g.add_child(unit_sphere(lib, lib.Matrix.translation(x, 0.0, 0.0)))
    c = lib.CSG(lib.UNION, g, unit_cube(lib))
    c.divide(1)
    left = c.left()
    assert left.num_of_children() == 2
    assert all(left.get_child(i).kind == 0 for i in range(2))  # each sphere went into a sub-group


# ---- local_intersect, oracle only ----------------------------------------------------------------------
def test_ray_misses_csg(orc):  # csg.rs:398-414
    c = orc.CSG(orc.UNION, unit_sphere(orc), unit_cube(orc))
    assert orc.intersect(c, orc.Ray(orc.point(0, 2, -5), orc.vector(0, 0, 1))) == []


def test_ray_hits_csg(orc):  # csg.rs:416-442
    s1 = unit_sphere(orc)
    s2 = unit_sphere(orc, orc.Matrix.translation(0.0, 0.0, 0.5))
    c = orc.CSG(orc.UNION, s1, s2)
    xs = orc.intersect(c, orc.Ray(orc.point(0, 0, -5), orc.vector(0, 0, 1)))
    assert [(t, h) for t, h, _, _ in xs] == [(4.0, s1.h), (6.5, s2.h)]


def test_csg_bbox_gates_the_children(orc):  # csg.rs:444-479, with spheres for the TestShape operands
    c = orc.CSG(orc.DIFFERENCE, unit_sphere(orc), unit_sphere(orc))
    assert orc.intersect(c, orc.Ray(orc.point(0, 0, -5), orc.vector(0, 1, 0))) == []
    assert len(orc.intersect(c, orc.Ray(orc.point(0, 0, -5), orc.vector(0, 0, 1)))) >= 0


def test_csg_in_world_shades_the_operand(orc):
    """An intersection of a CSG shape carries the operand that was hit (csg.rs:144-159), so
    shading uses that operand's material and a normal that passes through the CSG's transform."""
    red = orc.Phong(orc.Solid(orc.Color(1, 0, 0)), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0)
    blue = orc.Phong(orc.Solid(orc.Color(0, 0, 1)), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0)
    left = orc.Sphere(orc.Matrix.identity(), red, True)
    right = orc.Sphere(orc.Matrix.translation(0.0, 0.0, -0.75) * orc.Matrix.scaling(0.5, 0.5, 0.5), blue, True)
    c = orc.CSG(orc.DIFFERENCE, left, right, orc.Matrix.translation(0.0, 0.0, 1.0))
    w = orc.World([c], [orc.PointLight(orc.point(-10, 10, -10), orc.Color.white())])
    # the ray enters the carved hollow: first kept intersection is the far side of `right` (blue)
    col = orc.color_at(w, orc.Ray(orc.point(0, 0, -5), orc.vector(0, 0, 1)), 5)
    assert tuple(col) == (0.0, 0.0, 1.0)
    # off-axis past the hollow's rim it meets `left` itself (red)
    col = orc.color_at(w, orc.Ray(orc.point(0.8, 0, -5), orc.vector(0, 0, 1)), 5)
    assert tuple(col) == (1.0, 0.0, 0.0)
