"""The reference's per-shape intersection known answers (src/geometry/*.rs test modules), evaluated on
the device through the C ABI.

The ABI reports colours, not intersections, so each shape gets a material that turns the hit into a
colour: ambient 1, no diffuse or specular term, and the reference's TestPattern
(src/patterns/test_pattern.rs), whose colour is the pattern-space point, shifted by (10, 10, 10) so
that no hit is black.  One white light makes World::color_at return exactly
`over_point + 10` for the nearest intersection with t >= 0 (src/tuples/intersection.rs hit) and
(0, 0, 0) for a miss.  Expected: origin + t * direction for the book's t, to 2e-5 (over_point is
1e-5 off the surface); and the CPU oracle's colour to 1e-11.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INF = math.inf


def probe_material(api):
    return api.Phong(api.TestPattern(api.Matrix.translation(-10, -10, -10)), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0)


def first_hit(host, orc, make_shape, origin, direction, normalise=False):
    """Colour of the ray on device and oracle; returns the device's hit point or None for a miss."""
    d = np.array(direction, dtype=np.float64)
    if normalise:
        d = d / np.linalg.norm(d)
    o = np.array([origin], dtype=np.float64)

    def world(api):
        return api.World([make_shape(api, probe_material(api))], [api.PointLight(api.point(-10, 10, -10), api.Color.white())])
    got = host.color_at(world(host), o, d[None, :], remaining=0)[0]
    ref = orc.color_at_batch(world(orc), o, d[None, :], remaining=0)[0]
    assert np.array_equal(np.isnan(got), np.isnan(ref))   # the cone's apex has the normal 0/0
    ok = ~np.isnan(ref)
    assert np.abs(got[ok] - ref[ok]).max(initial=0.0) <= 1e-11
    if (got == 0.0).all():
        return None, d
    return got - 10.0, d


def expect(host, orc, make_shape, origin, direction, times, normalise=False):
    """times: the intersections the reference's test lists; the hit is the smallest t >= 0."""
    p, d = first_hit(host, orc, make_shape, origin, direction, normalise)
    ahead = [t for t in times if t >= 0]
    if not ahead:
        assert p is None
        return
    assert p is not None
    want = np.array(origin, dtype=np.float64) + min(ahead) * d
    assert np.abs(p - want).max() <= 2e-5, (p, want)


def sphere(t=None):
    return lambda api, m: api.Sphere(t(api) if t else api.Matrix.identity(), m, True)


@pytest.mark.parametrize("origin,times", [  # sphere.rs:153-246
    ((0, 0, -5), [4.0, 6.0]), ((0, 1, -5), [5.0, 5.0]), ((0, 2, -5), []), ((0, 0, 0), [-1.0, 1.0]), ((0, 0, 5), [-6.0, -4.0]),
])
def test_sphere(host, orc, origin, times):
    expect(host, orc, sphere(), origin, (0, 0, 1), times)


def test_sphere_transformed(host, orc):  # sphere.rs:288-322
    # the pattern sees the object-space point: scale the expectation back
    p, _ = first_hit(host, orc, sphere(lambda api: api.Matrix.scaling(2, 2, 2)), (0, 0, -5), (0, 0, 1))
    assert np.abs(p - np.array([0, 0, -2.0]) / 2.0).max() <= 2e-5   # t = 3: world (0, 0, -2)
    p, _ = first_hit(host, orc, sphere(lambda api: api.Matrix.translation(5, 0, 0)), (0, 0, -5), (0, 0, 1))
    assert p is None


def test_plane(host, orc):  # plane.rs:152-223
    plane = lambda api, m: api.Plane(api.Matrix.identity(), m, True)
    expect(host, orc, plane, (0, 10, 0), (0, 0, 1), [])      # parallel
    expect(host, orc, plane, (0, 0, 0), (0, 0, 1), [])       # coplanar
    expect(host, orc, plane, (0, 1, 0), (0, -1, 0), [1.0])   # from above
    expect(host, orc, plane, (0, -1, 0), (0, 1, 0), [1.0])   # from below


cube = lambda api, m: api.Cube(api.Matrix.identity(), m, True)


@pytest.mark.parametrize("origin,direction,times", [  # cube.rs:185-281
    ((5, 0.5, 0), (-1, 0, 0), (4.0, 6.0)), ((-5, 0.5, 0), (1, 0, 0), (4.0, 6.0)), ((0.5, 5, 0), (0, -1, 0), (4.0, 6.0)),
    ((0.5, -5, 0), (0, 1, 0), (4.0, 6.0)), ((0.5, 0, 5), (0, 0, -1), (4.0, 6.0)), ((0.5, 0, -5), (0, 0, 1), (4.0, 6.0)),
    ((0, 0.5, 0), (0, 0, 1), (-1.0, 1.0)),
    ((-2, 0, 0), (0.2673, 0.5345, 0.8018), ()), ((0, -2, 0), (0.8018, 0.2673, 0.5345), ()),
    ((0, 0, -2), (0.5345, 0.8018, 0.2673), ()), ((2, 0, 2), (0, 0, -1), ()), ((0, 2, 2), (0, -1, 0), ()),
    ((2, 2, 0), (-1, 0, 0), ()),
])
def test_cube(host, orc, origin, direction, times):
    expect(host, orc, cube, origin, direction, times)


def cylinder(lo=-INF, hi=INF, closed=False):
    return lambda api, m: api.Cylinder(api.Matrix.identity(), m, True, lo, hi, closed)


@pytest.mark.parametrize("origin,direction,times", [  # cylinder.rs:246-268
    ((1, 0, -5), (0, 0, 1), (5.0, 5.0)), ((0, 0, -5), (0, 0, 1), (4.0, 6.0)), ((0.5, 0, -5), (0.1, 1, 1), (6.80798, 7.08872)),
    ((1, 0, 0), (0, 1, 0), ()), ((0, 0, 0), (0, 1, 0), ()), ((0, 0, -5), (1, 1, 1), ()),
])
def test_cylinder(host, orc, origin, direction, times):
    expect(host, orc, cylinder(), origin, direction, times, normalise=True)


@pytest.mark.parametrize("origin,direction,hits", [  # cylinder.rs:270-324: truncated to 1 < y < 2
    ((0, 1.5, 0), (0.1, 1, 0), False), ((0, 3, -5), (0, 0, 1), False), ((0, 0, -5), (0, 0, 1), False),
    ((0, 2, -5), (0, 0, 1), False), ((0, 1, -5), (0, 0, 1), False), ((0, 1.5, -2), (0, 0, 1), True),
])
def test_cylinder_constrained(host, orc, origin, direction, hits):
    p, _ = first_hit(host, orc, cylinder(1.0, 2.0), origin, direction, normalise=True)
    assert (p is not None) == hits
    if hits:
        assert np.abs(p - (0, 1.5, -1)).max() <= 2e-5


@pytest.mark.parametrize("origin,direction", [  # cylinder.rs:366-414: every ray meets the closed cylinder twice
    ((0, 3, 0), (0, -1, 0)), ((0, 3, -2), (0, -1, 2)), ((0, 4, -2), (0, -1, 1)), ((0, 0, -2), (0, 1, 2)), ((0, -1, -2), (0, 1, 1)),
])
def test_cylinder_capped(host, orc, origin, direction):
    p, _ = first_hit(host, orc, cylinder(1.0, 2.0, True), origin, direction, normalise=True)
    assert p is not None
    # the nearest intersection lies on a cap or on the wall of the unit cylinder between y = 1 and 2
    on_cap = min(abs(p[1] - 1.0), abs(p[1] - 2.0)) <= 2e-5 and p[0] ** 2 + p[2] ** 2 <= 1.0 + 1e-4
    on_wall = abs(p[0] ** 2 + p[2] ** 2 - 1.0) <= 1e-4 and 1.0 - 2e-5 <= p[1] <= 2.0 + 2e-5
    assert on_cap or on_wall


def cone(lo=-INF, hi=INF, closed=False):
    return lambda api, m: api.Cone(api.Matrix.identity(), m, True, lo, hi, closed)


@pytest.mark.parametrize("origin,direction,times", [  # cone.rs:267-317
    ((0, 0, -5), (0, 0, 1), (5.0, 5.0)), ((0, 0, -5), (1, 1, 1), (8.66025, 8.66025)), ((1, 1, -5), (-0.5, -1, 1), (4.55006, 49.44994)),
    ((0, 0, -1), (0, 1, 1), (0.35355,)),   # parallel to one half
])
def test_cone(host, orc, origin, direction, times):
    p, d = first_hit(host, orc, cone(), origin, direction, normalise=True)
    assert p is not None
    if np.isnan(p).any():   # the first case hits the apex, where normalising (0, 0, 0) gives NaN
        assert tuple(origin) == (0, 0, -5) and tuple(direction) == (0, 0, 1)
        return
    want = np.array(origin, dtype=np.float64) + min(times) * d
    # at the apex the normal is (0, 0, 0) and over_point == point; elsewhere it is 1e-5 off; the
    # book's times are rounded to five places and the direction is steep, hence the wider margin
    assert np.abs(p - want).max() <= 5e-5


@pytest.mark.parametrize("origin,direction,hits", [  # cone.rs:319-356: closed, -0.5 <= y <= 0.5
    ((0, 0, -5), (0, 1, 0), False), ((0, 0, -0.25), (0, 1, 1), True), ((0, 0, -0.25), (0, 1, 0), True),
])
def test_cone_capped(host, orc, origin, direction, hits):
    p, _ = first_hit(host, orc, cone(-0.5, 0.5, True), origin, direction, normalise=True)
    assert (p is not None) == hits


def triangle(api, m):
    return api.Triangle(api.point(0, 1, 0), api.point(-1, 0, 0), api.point(1, 0, 0), None, m, True)


@pytest.mark.parametrize("origin,direction,times", [  # triangle.rs:230-324
    ((0, -1, -2), (0, 1, 0), ()), ((1, 1, -2), (0, 0, 1), ()), ((-1, 1, -2), (0, 0, 1), ()), ((0, -1, -2), (0, 0, 1), ()),
    ((0, 0.5, -2), (0, 0, 1), (2.0,)),
])
def test_triangle(host, orc, origin, direction, times):
    expect(host, orc, triangle, origin, direction, times)


def test_smooth_triangle_interpolates_the_normal(host, orc):  # smooth_triangle.rs:211-251
    """The hit at u = 0.45, v = 0.25 has the interpolated normal (-0.5547, 0.83205, 0); over_point
    is 1e-5 along it, which the position colour shows."""
    def smooth(api, m):
        return api.SmoothTriangle(api.point(0, 1, 0), api.point(-1, 0, 0), api.point(1, 0, 0), api.vector(0, 1, 0),
                                  api.vector(-1, 0, 0), api.vector(1, 0, 0), None, m, True)
    p, _ = first_hit(host, orc, smooth, (-0.2, 0.3, -2), (0, 0, 1))
    off = (p - np.array([-0.2, 0.3, 0.0])) / 1e-5
    assert np.abs(off - (-0.5547, 0.83205, 0.0)).max() <= 1e-3


def test_group(host, orc):  # group.rs:300-372
    """A ray through a group of three spheres hits the nearest one; a transformed group moves its
    child (group.rs:343-372: scaling(2) around a sphere at x = 5, ray from (10, 0, -10))."""
    def three(api, m):
        M = api.Matrix
        g = api.Group()
        g.add_child(api.Sphere(M.identity(), m, True))
        g.add_child(api.Sphere(M.translation(0, 0, -3), m, True))
        g.add_child(api.Sphere(M.translation(5, 0, 0), m, True))
        return g
    p, _ = first_hit(host, orc, three, (0, 0, -5), (0, 0, 1))
    assert np.abs(p - (0, 0, -1)).max() <= 2e-5     # object space of the sphere at z = -3: t = 1

    def moved(api, m):
        M = api.Matrix
        g = api.Group(M.scaling(2, 2, 2))
        g.add_child(api.Sphere(M.translation(5, 0, 0), m, True))
        return g
    p, _ = first_hit(host, orc, moved, (10, 0, -10), (0, 0, 1))
    assert p is not None and np.abs(p - (0, 0, -1)).max() <= 2e-5
    p, _ = first_hit(host, orc, lambda api, m: api.Group(), (0, 0, 0), (0, 0, 1))   # empty group: group.rs:284-298
    assert p is None
