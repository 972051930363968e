"""CSG shapes (src/geometry/csg.rs) on the GPU against the oracle: the book's CSG known answers
through World::color_at, then seeded random worlds with nested CSG shapes whose operands are
transformed quadrics, groups and other CSG shapes."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def flat(o, color):
    return o.Phong(o.Solid(o.Color(*color)), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0)


def both(host, orc, make_world, rays, remaining=5):
    o = np.array([r[0] for r in rays], dtype=np.float64)
    d = np.array([r[1] for r in rays], dtype=np.float64)
    got = host.color_at(make_world(host), o, d, remaining=remaining)
    ref = orc.color_at_batch(make_world(orc), o, d, remaining=remaining)
    assert np.abs(got - ref).max() <= 1e-11
    return got


def test_union_of_two_spheres_hits_the_first_surface(host, orc):  # csg.rs:416-442
    def world(o):
        s1 = o.Sphere(o.Matrix.identity(), flat(o, (1, 0, 0)), True)
        s2 = o.Sphere(o.Matrix.translation(0.0, 0.0, 0.5), flat(o, (0, 0, 1)), True)
        return o.World([o.CSG(o.UNION, s1, s2)], [o.PointLight(o.point(-10, 10, -10), o.Color.white())])
    got = both(host, orc, world, [((0, 0, -5), (0, 0, 1)), ((0, 0, 5), (0, 0, -1)), ((0, 2, -5), (0, 0, 1))])
    assert tuple(got[0]) == (1.0, 0.0, 0.0)   # t = 4 on s1
    assert tuple(got[1]) == (0.0, 0.0, 1.0)   # from behind: s2 first
    assert tuple(got[2]) == (0.0, 0.0, 0.0)   # misses the box (csg.rs:398-414)


@pytest.mark.parametrize("op_name,expected", [
    ("DIFFERENCE", (0.0, 0.0, 1.0)),    # left minus right: first kept surface is the far side of right
    ("INTERSECTION", (1.0, 0.0, 0.0)),  # overlap: first kept surface is left's near side (inside right)
    ("UNION", (0.0, 0.0, 1.0)),         # right's near side comes first
])
def test_operations_pick_the_surface_the_filter_keeps(host, orc, op_name, expected):  # csg.rs:362-396
    def world(o):
        left = o.Sphere(o.Matrix.identity(), flat(o, (1, 0, 0)), True)
        right = o.Sphere(o.Matrix.translation(0.0, 0.0, -0.75) * o.Matrix.scaling(0.5, 0.5, 0.5), flat(o, (0, 0, 1)), True)
        c = o.CSG(getattr(o, op_name), left, right, o.Matrix.translation(0.0, 0.0, 1.0))
        return o.World([c], [o.PointLight(o.point(-10, 10, -10), o.Color.white())])
    got = both(host, orc, world, [((0, 0, -5), (0, 0, 1))])
    assert tuple(got[0]) == expected


def random_transform(api, rng, spread=1.0):
    M = api.Matrix
    m = M.translation(*rng.uniform(-spread, spread, 3))
    k = rng.integers(0, 4)
    if k == 0:
        m = m * M.rotation_y(rng.uniform(-3, 3))
    elif k == 1:
        m = m * M.rotation_x(rng.uniform(-3, 3)) * M.scaling(*rng.uniform(0.5, 1.5, 3))
    elif k == 2:
        m = m * M.scaling(*rng.uniform(0.5, 1.6, 3))
    return m


def random_material(api, rng):
    refl = float(rng.choice([0.0, 0.0, 0.4]))
    transp = float(rng.choice([0.0, 0.0, 0.7]))
    return api.Phong(api.Solid(api.Color(*rng.uniform(0, 1, 3))), 0.2, 0.7, 0.3, 50.0, refl, transp,
                     float(rng.uniform(1.0, 2.0)))


def random_operand(api, rng, depth):
    """A leaf, a small group, or another CSG shape; never more than 24 intersections per ray in total."""
    k = rng.integers(0, 8) if depth < 2 else rng.integers(0, 5)
    t, m = random_transform(api, rng, 0.8), random_material(api, rng)
    cs = bool(rng.random() < 0.8)
    if k == 0 or k == 1:
        return api.Sphere(t, m, cs), 2
    if k == 2:
        return api.Cube(t, m, cs), 2
    if k == 3:
        lo, hi = sorted(rng.uniform(-1.2, 1.2, 2))
        return api.Cylinder(t, m, cs, float(lo), float(hi), True), 4
    if k == 4:
        lo, hi = sorted(rng.uniform(-1.2, 1.2, 2))
        return api.Cone(t, m, cs, float(lo), float(hi), True), 4
    if k == 5:
        g = api.Group(t if rng.random() < 0.5 else None)
        total = 0
        for _ in range(2):
            c, n = random_operand(api, rng, 2)
            g.add_child(c)
            total += n
        return g, total
    left, nl = random_operand(api, rng, depth + 1)
    right, nr = random_operand(api, rng, depth + 1)
    op = [api.DIFFERENCE, api.INTERSECTION, api.UNION][rng.integers(0, 3)]
    return api.CSG(op, left, right, t if rng.random() < 0.5 else None), nl + nr


def random_csg_world(api, seed):
    rng = np.random.default_rng(500 + seed)
    objects = [api.Plane(api.Matrix.translation(0, -2.0, 0), random_material(api, rng), True)]
    for _ in range(rng.integers(1, 4)):
        while True:
            state = rng.bit_generator.state
            left, nl = random_operand(api, rng, 0)
            right, nr = random_operand(api, rng, 0)
            if nl + nr <= 24:
                break
        op = [api.DIFFERENCE, api.INTERSECTION, api.UNION][rng.integers(0, 3)]
        c = api.CSG(op, left, right, random_transform(api, rng, 2.5))
        if rng.random() < 0.3:
            g = api.Group(random_transform(api, rng, 1.0))
            g.add_child(c)
            g.add_child(api.Sphere(random_transform(api, rng, 2.0), random_material(api, rng), True))
            if rng.random() < 0.5:
                g.divide(1)
            objects.append(g)
        else:
            if rng.random() < 0.5:
                c.divide(1)
            objects.append(c)
    lights = [api.PointLight(api.point(*rng.uniform(-10, 10, 3)), api.Color(*rng.uniform(0.3, 1.0, 3)))
              for _ in range(rng.integers(1, 3))]
    return api.World(objects, lights)


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_csg_scene(host, orc, seed):
    W, H = 64, 48
    depth = [5, 3, 0, 1][seed % 4]

    def camera(api):
        return api.Camera(H, W, 1.0, api.Matrix.view_transform(api.point(0, 1.0, -8), api.point(0, 0, 0),
                                                                api.vector(0, 1, 0)))
    img = host.render(random_csg_world(host, seed), camera(host), max_depth=depth)
    ref = orc.render(random_csg_world(orc, seed), camera(orc), remaining=depth, threads=8)
    finite = np.isfinite(ref)
    assert np.array_equal(np.isfinite(img), finite)
    err = np.abs(img[finite] - ref[finite]) / np.maximum(1.0, np.abs(ref[finite]))
    assert err.max() <= 1e-9, "seed %d: max rel err %.3e" % (seed, err.max())
    assert (ref[finite] != 0).any()


def test_csg_limits_are_reported(host):
    """More intersections per ray than the per-lane list holds is refused, not truncated."""
    from raytracer_amd import abi
    M = host.Matrix

    def chain(n):
        shape = host.Sphere(M.identity(), host.Phong.default(), True)
        for k in range(n):
            shape = host.CSG(host.UNION, shape, host.Sphere(M.translation(0.1 * (k + 1), 0, 0), host.Phong.default(), True))
        return shape
    cam = host.Camera(8, 8, 1.0, M.identity())
    light = [host.PointLight(host.point(0, 5, -5), host.Color.white())]
    with pytest.raises(abi.RtError) as e:
        host.render(host.World([chain(abi.RT_MAX_CSG_NEST + 1)], light), cam)   # one more nested CSG shape than allowed
    assert e.value.code == abi.RT_ERR_LIMIT
    wide = host.Group()
    for k in range(abi.RT_MAX_CSG_HITS // 2):              # with the right operand: more intersections than RT_MAX_CSG_HITS
        wide.add_child(host.Sphere(M.translation(0.1 * k, 0, 0), host.Phong.default(), True))
    with pytest.raises(abi.RtError) as e:
        host.render(host.World([host.CSG(host.UNION, wide, host.Sphere(M.identity(), host.Phong.default(), True))], light), cam)
    assert e.value.code == abi.RT_ERR_LIMIT
