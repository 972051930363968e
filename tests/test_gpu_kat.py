"""GPU known-answer and randomised parity tests through the C ABI.

The world.rs known answers that are expressible as World::color_at (reference file:line per test)
are evaluated on the device with rt_color_at and must match the book values to the reference's 1e-5
and the oracle to 1e-11.  Seeded random scenes then cover what the fixed scenes do not: every shape
kind under nested transformed groups, `divide`, nested patterns, shapes that cast no shadow, planes
inside groups (infinite boxes), up to four lights and every recursion depth."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPS = 1e-5
SQRT_2 = math.sqrt(2.0)


def default_world(o, extra=(), outer_ambient=0.1):
    M = o.Matrix
    outer = o.Sphere(M.identity(), o.Phong(o.Solid(o.Color(0.8, 1.0, 0.6)), outer_ambient, 0.7, 0.2, 200.0, 0.0, 0.0,
                                            1.0), True)
    inner = o.Sphere(M.scaling(0.5, 0.5, 0.5), o.Phong.default(), True)
    return o.World([outer, inner] + [f(o) for f in extra], [o.PointLight(o.point(-10, 10, -10), o.Color.white())])


def both(host, orc, make_world, rays, remaining=5):
    o = np.array([r[0] for r in rays], dtype=np.float64)
    d = np.array([r[1] for r in rays], dtype=np.float64)
    got = host.color_at(make_world(host), o, d, remaining=remaining)
    ref = orc.color_at_batch(make_world(orc), o, d, remaining=remaining)
    assert np.abs(got - ref).max() <= 1e-11
    return got


def test_color_at_default_world(host, orc):  # world.rs:478-503
    got = both(host, orc, default_world, [((0, 0, 5), (0, 0, 1)), ((0, 0, -5), (0, 0, 1))])
    assert np.abs(got[0]).max() == 0.0
    assert np.abs(got[1] - (0.38066, 0.47583, 0.2855)).max() <= EPS


def test_color_at_inner_sphere(host, orc):  # world.rs:506-541
    def world(o):
        M = o.Matrix
        mat = o.Phong(o.Solid(), 1.0, 0.9, 0.9, 200.0, 0.0, 0.0, 1.0)
        return o.World([o.Sphere(M.identity(), mat, True), o.Sphere(M.scaling(0.5, 0.5, 0.5), mat, True)],
                       [o.PointLight(o.point(-10, 10, -10), o.Color.white())])
    got = both(host, orc, world, [((0, 0, 0.75), (0, 0, -1))])
    assert np.abs(got[0] - 1.0).max() <= EPS


def test_reflective_plane(host, orc):  # world.rs:778-838 (shade_hit == color_at: the plane is the first hit)
    plane = lambda o: o.Plane(o.Matrix.translation(0, -1, 0), o.Phong(o.Solid(), 0.1, 0.9, 0.9, 200.0, 0.5, 0.0, 1.0), True)
    ray = ((0, 0, -3), (0, -SQRT_2 / 2, SQRT_2 / 2))
    got = both(host, orc, lambda o: default_world(o, [plane]), [ray])
    assert np.abs(got[0] - (0.87675, 0.92434, 0.82917)).max() <= EPS
    # with no bounces left the reflection contributes nothing (world.rs:841-900)
    got0 = both(host, orc, lambda o: default_world(o, [plane]), [ray], remaining=0)
    assert (got0[0] < got[0]).all()


def floor_and_ball(reflective):
    floor = lambda o: o.Plane(o.Matrix.translation(0, -1, 0),
                              o.Phong(o.Solid(o.Color.white()), 0.1, 0.9, 0.9, 200.0, reflective, 0.5, 1.5), True)
    ball = lambda o: o.Sphere(o.Matrix.translation(0, -3.5, -0.5),
                              o.Phong(o.Solid(o.Color.red()), 0.5, 0.9, 0.9, 200.0, 0.0, 0.0, 1.0), True)
    return [floor, ball]


def test_transparent_floor(host, orc):  # world.rs:1193-1266
    got = both(host, orc, lambda o: default_world(o, floor_and_ball(0.0)), [((0, 0, -3), (0, -SQRT_2 / 2, SQRT_2 / 2))])
    assert np.abs(got[0] - (0.93642, 0.68642, 0.68642)).max() <= EPS


def test_reflective_transparent_floor_uses_schlick(host, orc):  # world.rs:1371-1445
    got = both(host, orc, lambda o: default_world(o, floor_and_ball(0.5)), [((0, 0, -3), (0, -SQRT_2 / 2, SQRT_2 / 2))])
    assert np.abs(got[0] - (0.93391, 0.69643, 0.69243)).max() <= EPS


def test_refraction_through_nested_glass(host, orc):
    """n1/n2 bookkeeping (world.rs:99-132, known answers at :903-986): rays through overlapping glass
    spheres, started outside, inside one and inside two of them."""
    def world(o):
        M = o.Matrix

        def g(t, ri):
            return o.Sphere(t, o.Phong(o.Solid(o.Color(0.2, 0.3, 0.4)), 0.1, 0.5, 0.3, 100.0, 0.3, 0.9, ri), True)
        back = o.Plane(M.translation(0, 0, 6) * M.rotation_x(math.pi / 2),
                       o.Phong(o.Checker(None, None, M.scaling(0.5, 0.5, 0.5)), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0), True)
        return o.World([g(M.scaling(2, 2, 2), 1.5), g(M.translation(0, 0, -0.25), 2.0), g(M.translation(0, 0, 0.25), 2.5),
                        back], [o.PointLight(o.point(-5, 5, -8), o.Color.white())])
    rng = np.random.default_rng(3)
    rays = []
    for z in (-4.0, -1.5, -0.5, 0.0, 0.5, 1.5):
        for _ in range(40):
            d = np.array([0, 0, 1.0]) + rng.uniform(-0.35, 0.35, 3)
            rays.append(((rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), z), tuple(d / np.linalg.norm(d))))
    both(host, orc, world, rays)


# ---- randomised scenes ---------------------------------------------------------------------------------
def random_transform(api, rng, scale=1.0):
    M = api.Matrix
    m = M.translation(*rng.uniform(-3, 3, 3) * scale)
    for _ in range(rng.integers(0, 3)):
        k = rng.integers(0, 5)
        if k == 0:
            m = m * M.rotation_x(rng.uniform(-3, 3))
        elif k == 1:
            m = m * M.rotation_y(rng.uniform(-3, 3))
        elif k == 2:
            m = m * M.rotation_z(rng.uniform(-3, 3))
        elif k == 3:
            m = m * M.scaling(*rng.uniform(0.4, 1.8, 3))
        else:
            m = m * M.shearing(*rng.uniform(-0.3, 0.3, 6))
    return m


def random_pattern(api, rng, depth=0):
    C = api.Color
    k = rng.integers(0, 7) if depth < 3 else 0
    if k == 0:
        return api.Solid(C(*rng.uniform(0, 1, 3)))
    t = random_transform(api, rng, 0.3) if rng.random() < 0.7 else None
    a, b = random_pattern(api, rng, depth + 1), random_pattern(api, rng, depth + 1)
    if k == 6:
        return api.Perturbed(a, float(rng.uniform(0.05, 0.5)), t)
    return [api.Striped, api.Gradient, api.Ring, api.Checker, api.Blended][k - 1](a, b, t)


def random_material(api, rng):
    refl = float(rng.choice([0.0, 0.0, rng.uniform(0.1, 0.9)]))
    transp = float(rng.choice([0.0, 0.0, rng.uniform(0.1, 1.0)]))
    return api.Phong(random_pattern(api, rng), *[float(v) for v in rng.uniform(0, 1, 3)], float(rng.uniform(5, 300)),
                     refl, transp, float(rng.uniform(1.0, 2.5)))


def random_leaf(api, rng):
    t, m = random_transform(api, rng), random_material(api, rng)
    cs = bool(rng.random() < 0.85)
    k = rng.integers(0, 7)
    if k == 0:
        return api.Sphere(t, m, cs)
    if k == 1:
        return api.Cube(t, m, cs)
    if k in (2, 3):
        lo, hi = sorted(rng.uniform(-1.5, 1.5, 2))
        closed = bool(rng.random() < 0.6)
        if rng.random() < 0.2:
            lo, hi, closed = -math.inf, math.inf, False  # closed infinite quadrics give inf/NaN times
        return (api.Cylinder if k == 2 else api.Cone)(t, m, cs, float(lo), float(hi), closed)
    pts = [api.point(*rng.uniform(-1.5, 1.5, 3)) for _ in range(3)]
    if k == 4:
        return api.Triangle(*pts, t if rng.random() < 0.3 else None, m, cs)
    if k == 5:
        nrm = [api.vector(*rng.uniform(-1, 1, 3)) for _ in range(3)]
        return api.SmoothTriangle(*pts, *nrm, t if rng.random() < 0.3 else None, m, cs)
    return api.Plane(t, m, cs)


def random_group(api, rng, depth=0):
    g = api.Group(random_transform(api, rng) if rng.random() < 0.6 else None)
    for _ in range(rng.integers(1, 5)):
        if depth < 2 and rng.random() < 0.35:
            g.add_child(random_group(api, rng, depth + 1))
        else:
            g.add_child(random_leaf(api, rng))
    return g


def random_world(api, seed):
    rng = np.random.default_rng(seed)
    objects = []
    for _ in range(rng.integers(2, 6)):
        if rng.random() < 0.5:
            g = random_group(api, rng)
            if rng.random() < 0.7:
                g.divide(int(rng.integers(1, 4)))
            objects.append(g)
        else:
            objects.append(random_leaf(api, rng))
    lights = [api.PointLight(api.point(*rng.uniform(-12, 12, 3)), api.Color(*rng.uniform(0.1, 1.0, 3)))
              for _ in range(rng.integers(0, 5))]
    return api.World(objects, lights)


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_scene(host, orc, seed):
    W, H = 56, 40
    depth = seed % 6

    def camera(api):
        return api.Camera(H, W, 1.0, api.Matrix.view_transform(api.point(0, 1.5, -9), api.point(0, 0, 0),
                                                                api.vector(0, 1, 0)))
    img = host.render(random_world(host, seed), camera(host), max_depth=depth)
    ref = orc.render(random_world(orc, seed), camera(orc), remaining=depth, threads=8)
    finite = np.isfinite(ref)
    # a ray through a degenerate configuration can give NaN in the reference arithmetic itself; the
    # device must then agree on where that happens
    assert np.array_equal(np.isfinite(img), finite)
    err = np.abs(img[finite] - ref[finite])
    scale = np.maximum(1.0, np.abs(ref[finite]))
    assert (err / scale).max() <= 1e-9, "seed %d: max rel err %.3e" % (seed, (err / scale).max())


# ---- conservative culling must never change a result ---------------------------------------------------
def quadric_world(api, seed):
    """Finite spheres, cubes and cylinders (the leaves the device skips through padded boxes) under
    stretched, sheared and tiny transforms, some transparent so that the container bookkeeping runs."""
    rng = np.random.default_rng(1000 + seed)
    M = api.Matrix
    objects = []
    for _ in range(12):
        t = M.translation(*rng.uniform(-4, 4, 3))
        t = t * M.rotation_y(rng.uniform(-3, 3)) * M.rotation_x(rng.uniform(-3, 3))
        t = t * M.scaling(*(10.0 ** rng.uniform(-1.5, 0.5, 3)))
        if rng.random() < 0.4:
            t = t * M.shearing(*rng.uniform(-0.5, 0.5, 6))
        transp = float(rng.choice([0.0, 0.0, 0.8]))
        m = api.Phong(api.Solid(api.Color(*rng.uniform(0, 1, 3))), 0.2, 0.7, 0.3, 50.0,
                      float(rng.choice([0.0, 0.4])), transp, float(rng.uniform(1.0, 2.0)))
        k = rng.integers(0, 3)
        if k == 0:
            objects.append(api.Sphere(t, m, True))
        elif k == 1:
            objects.append(api.Cube(t, m, True))
        else:
            lo, hi = sorted(rng.uniform(-1.5, 1.5, 2))
            objects.append(api.Cylinder(t, m, True, float(lo), float(hi), bool(rng.random() < 0.5)))
    lights = [api.PointLight(api.point(*rng.uniform(-12, 12, 3)), api.Color.white()) for _ in range(2)]
    return api.World(objects, lights)


def sheared_world(api, seed):
    """Quadrics under strongly sheared transforms and under a stretch applied AFTER a rotation: maps
    whose smallest stretch is far below the length of any column of the matrix, which is what the
    padding of the cull boxes has to be sized by."""
    rng = np.random.default_rng(7000 + seed)
    M = api.Matrix
    objects = []
    while len(objects) < 10:
        rot = M.rotation_y(rng.uniform(-3, 3)) * M.rotation_z(rng.uniform(-3, 3))
        if rng.random() < 0.5:
            t = M.scaling(*(10.0 ** rng.uniform(-1.0, 0.7, 3))) * rot          # stretch after rotation
        else:
            t = rot * M.shearing(*rng.uniform(-3.0, 3.0, 6)) * M.scaling(*(10.0 ** rng.uniform(-0.7, 0.3, 3)))
        t = M.translation(*rng.uniform(-4, 4, 3)) * t
        try:
            t.inverse()
        except ValueError:
            continue  # |det| < 1e-5: the reference would panic on it
        m = api.Phong(api.Solid(api.Color(*rng.uniform(0, 1, 3))), 0.2, 0.7, 0.3, 50.0,
                      float(rng.choice([0.0, 0.4])), float(rng.choice([0.0, 0.0, 0.8])), float(rng.uniform(1.0, 2.0)))
        k = len(objects) % 3
        if k == 0:
            objects.append(api.Sphere(t, m, True))
        elif k == 1:
            objects.append(api.Cube(t, m, True))
        else:
            objects.append(api.Cylinder(t, m, True, -1.0, 1.5, True))
    lights = [api.PointLight(api.point(*rng.uniform(-12, 12, 3)), api.Color.white()) for _ in range(2)]
    return api.World(objects, lights)


def triangle_fan_group(api, rng):
    """A small closed mesh (an octahedron subdivided once: 32 triangles) in a transformed group,
    divided like the OBJ meshes are, so that sibling runs get cluster boxes and groups get pruned."""
    M = api.Matrix
    verts = [np.array(v, dtype=np.float64) for v in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))]
    faces = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    tris = []
    for a, b, c in faces:
        pa, pb, pc = verts[a], verts[b], verts[c]
        mab, mbc, mca = [(x + y) / np.linalg.norm(x + y) for x, y in ((pa, pb), (pb, pc), (pc, pa))]
        tris += [(pa, mab, mca), (mab, pb, mbc), (mca, mbc, pc), (mab, mbc, mca)]
    mat = api.Phong(api.Solid(api.Color(*rng.uniform(0, 1, 3))), 0.2, 0.7, 0.3, 50.0, 0.3, 0.0, 1.0)
    g = api.Group(M.translation(*rng.uniform(-3, 3, 3)) * M.rotation_y(rng.uniform(-3, 3)) * M.scaling(*rng.uniform(0.5, 2.0, 3)))
    for k, (p1, p2, p3) in enumerate(tris):
        pts = [api.point(*p) for p in (p1, p2, p3)]
        if k % 2:
            g.add_child(api.SmoothTriangle(*pts, *[api.vector(*p) for p in (p1, p2, p3)], None, mat, True))
        else:
            g.add_child(api.Triangle(*pts, None, mat, True))
    g.divide(1)
    return g


def grouped_world(api, seed):
    """quadric_world's objects inside nested, transformed, divided groups, plus a triangle mesh."""
    flat = quadric_world(api, seed)
    rng = np.random.default_rng(2000 + seed)
    M = api.Matrix
    outer = api.Group(M.rotation_z(rng.uniform(-0.5, 0.5)))
    inner = api.Group(M.translation(*rng.uniform(-1, 1, 3)) * M.scaling(*rng.uniform(0.7, 1.4, 3)))
    for k, obj in enumerate(flat.objects):
        (inner if k % 2 else outer).add_child(obj)
    outer.add_child(inner)
    outer.add_child(triangle_fan_group(api, rng))
    outer.divide(1)
    return api.World([outer], flat.lights)


def mixed_world(api, seed):
    """Thirty-odd objects directly in the world, the way scenes are built without groups: finite
    quadrics, transformed triangles, small transformed groups (some holding a cone), a CSG shape, and
    a plane and a cone in between, which end the runs of siblings that share cluster boxes."""
    rng = np.random.default_rng(3000 + seed)
    M = api.Matrix
    base = quadric_world(api, seed)
    objects = list(base.objects)

    def mat():
        return api.Phong(api.Solid(api.Color(*rng.uniform(0, 1, 3))), 0.2, 0.7, 0.3, 50.0,
                         float(rng.choice([0.0, 0.3])), float(rng.choice([0.0, 0.0, 0.7])), 1.5)

    def place():
        return M.translation(*rng.uniform(-4, 4, 3)) * M.rotation_y(rng.uniform(-3, 3)) * M.scaling(*rng.uniform(0.2, 0.8, 3))
    for k in range(20):
        kind = k % 5
        if kind == 0:
            pts = [api.point(*rng.uniform(-1, 1, 3)) for _ in range(3)]
            objects.append(api.Triangle(*pts, place(), mat(), True))
        elif kind == 1:
            g = api.Group(place())
            g.add_child(api.Sphere(M.translation(-1, 0, 0), mat(), True))
            g.add_child(api.Cube(M.translation(1, 0, 0) * M.scaling(0.5, 0.5, 0.5), mat(), True))
            if k % 2:
                g.add_child(api.Cone(M.translation(0, 1.5, 0), mat(), True, -1.0, 0.0, bool(k % 4 == 1)))
            objects.append(g)
        elif kind == 2:
            objects.append(api.CSG(api.DIFFERENCE, api.Cube(M.identity(), mat(), True),
                                   api.Sphere(M.translation(0.4, 0.4, -0.4), mat(), True), place()))
        elif kind == 3:
            objects.append(api.Sphere(place(), mat(), True))
        else:
            objects.append(api.Cylinder(place(), mat(), True, -1.0, 1.0, True))
        if k == 7:
            objects.append(api.Plane(M.translation(0, -5, 0), mat(), True))
        if k == 13:
            objects.append(api.Cone(place(), mat(), True, -1.0, 0.5, True))
    return api.World(objects, base.lights)


@pytest.mark.parametrize("kind", ["flat", "grouped", "mixed", "sheared"])
@pytest.mark.parametrize("seed", list(range(8)))
def test_culling_is_conservative(host, orc, seed, kind):
    """Rays from 10 to 10^7 units away aimed at and just past the silhouettes of the objects: the
    padded boxes (single leaves, runs of siblings) and the pruning of groups by their box times
    must keep every ray the exact arithmetic counts as a hit, including the false hits that
    cancellation produces for far origins."""
    make = {"flat": quadric_world, "grouped": grouped_world, "mixed": mixed_world, "sheared": sheared_world}[kind]
    rng = np.random.default_rng(seed)
    origins, dirs = [], []
    for dist in (10.0, 1e3, 1e5, 1e7):
        for _ in range(300):
            target = rng.uniform(-5, 5, 3)
            o = rng.normal(size=3)
            o = o / np.linalg.norm(o) * dist
            if rng.random() < 0.3:
                o[rng.integers(0, 3)] = 0.0  # axis-parallel planes of the boxes
            d = target - o
            if rng.random() < 0.2:
                d[rng.integers(0, 3)] = 0.0  # a direction component the slab tests treat as parallel
            origins.append(o)
            dirs.append(d / np.linalg.norm(d))
    o = np.array(origins)
    d = np.array(dirs)
    got = host.color_at(make(host, seed), o, d, remaining=3)
    ref = orc.color_at_batch(make(orc, seed), o, d, remaining=3)
    finite = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), finite)
    err = np.abs(got[finite] - ref[finite]) / np.maximum(1.0, np.abs(ref[finite]))
    assert err.max() <= 1e-9, "seed %d: max rel err %.3e" % (seed, err.max())
    assert (ref[finite] != 0.0).any()


def test_cone_in_group_hit_outside_its_box(host, orc):
    """For a ray parallel to one half of a cone the reference reports the wall intersection without
    the minimum < y < maximum check (cone.rs:119-127), so a truncated cone inside a group can be
    hit at a time outside the interval in which the ray is inside the group's box.  The group is
    still entered whenever the ray's line meets the box (group.rs:139), so that hit must win over a
    farther one found earlier, even though the box lies entirely behind it."""
    def world(o):
        M = o.Matrix
        flat = lambda c: o.Phong(o.Solid(o.Color(*c)), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0)
        g = o.Group()
        g.add_child(o.Cone(M.identity(), flat((0.9, 0.1, 0.1)), True, -1.0, 0.0, False))
        # a wall between the early wall intersection and the box: it is hit first in scene order
        wall = o.Plane(M.translation(0, 0, -1.5) * M.rotation_x(math.pi / 2), flat((0.1, 0.1, 0.9)), True)
        return o.World([wall, g], [o.PointLight(o.point(-10, 10, -10), o.Color.white())])
    d = (0.0, SQRT_2 / 2, SQRT_2 / 2)
    rays = [((0.0, -4.5, -5.0), d), ((0.1, -4.5, -5.0), d), ((0.0, -4.4, -5.0), d)]
    got = both(host, orc, world, rays, remaining=0)
    assert np.abs(got[0] - (0.9, 0.1, 0.1)).max() <= 1e-12   # the cone's unchecked intersection at t = 3.36
    for sign in (1.0, -1.0):  # and from the other side, where the box is ahead and the hit behind
        o = np.array(rays[0][0]) + sign * 20.0 * np.array(d)
        both(host, orc, world, [(tuple(o), tuple(-sign * np.array(d)))], remaining=0)


def scattered(o, k):
    """Small spheres far out on alternating sides, listed between the objects of interest, so that
    sorting the world's objects in space pays and the device does not visit them in list order."""
    M = o.Matrix
    x = (10.0 + 2 * k) * (1 if k % 2 else -1)
    return o.Sphere(M.translation(x, 6.0, 0.5 * k) * M.scaling(0.3, 0.3, 0.3),
                    o.Phong(o.Solid(o.Color(0.5, 0.5, 0.5)), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0), True)


@pytest.mark.parametrize("swap", [False, True])
def test_equal_times_go_to_the_first_object_in_list_order(host, orc, swap):
    """Two cubes share the face x = 1 and axis-parallel rays meet it at exactly the same time in both
    object spaces (t = 4).  Intersection::hit takes the first of equal times in the stable order of
    World::intersect_world (world.rs:55-68), i.e. the object listed first, although the other cube's
    box lies lower on every axis and a spatial ordering of the records visits it first."""
    def world(o):
        M = o.Matrix
        flat = lambda c: o.Phong(o.Solid(o.Color(*c)), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0)
        a = o.Cube(M.identity(), flat((0.9, 0.1, 0.1)), True)                                  # [-1, 1]^3
        b = o.Cube(M.translation(-1, -1, -1) * M.scaling(2, 2, 2), flat((0.1, 0.1, 0.9)), True)  # [-3, 1]^3
        first, second = (b, a) if swap else (a, b)
        objects = [scattered(o, 0), first, scattered(o, 1), scattered(o, 2), second, scattered(o, 3), scattered(o, 4),
                   scattered(o, 5)]
        return o.World(objects, [o.PointLight(o.point(-10, 10, -10), o.Color.white())])
    rng = np.random.default_rng(17)
    rays = [((5.0, float(y), float(z)), (-1.0, 0.0, 0.0)) for y, z in rng.uniform(-0.99, 0.99, (200, 2))]
    got = both(host, orc, world, rays, remaining=0)
    want = (0.1, 0.1, 0.9) if swap else (0.9, 0.1, 0.1)
    assert np.abs(got - want).max() <= 1e-12


@pytest.mark.parametrize("swap", [False, True])
def test_equal_times_in_the_container_walk(host, orc, swap):
    """The same two cubes made of glass with different refractive indices.  A ray started inside both
    has their two entries behind it at exactly the same time (t = -0.5), so n1 at the next surface
    (world.rs:99-132) is the index of whichever cube the sorted list holds last: the one listed later
    in the world.  A ray from outside meets both at t = 4 and enters the one listed first."""
    def world(o):
        M = o.Matrix

        def glass(t, ri, c):
            return o.Cube(t, o.Phong(o.Solid(o.Color(*c)), 0.1, 0.4, 0.3, 100.0, 0.2, 0.9, ri), True)
        a = glass(M.identity(), 1.5, (0.4, 0.2, 0.2))
        b = glass(M.translation(-1, -1, -1) * M.scaling(2, 2, 2), 2.0, (0.2, 0.4, 0.2))
        first, second = (b, a) if swap else (a, b)
        back = o.Plane(M.translation(-8, 0, 0) * M.rotation_z(math.pi / 2),
                       o.Phong(o.Checker(None, None, M.scaling(0.5, 0.5, 0.5)), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0), True)
        objects = [scattered(o, 0), first, scattered(o, 1), back, scattered(o, 2), second, scattered(o, 3), scattered(o, 4),
                   scattered(o, 5)]
        return o.World(objects, [o.PointLight(o.point(-5, 5, -8), o.Color.white())])
    rng = np.random.default_rng(29)
    rays = []
    for x in (5.0, 0.5):
        rays += [((x, float(y), float(z)), (-1.0, 0.0, 0.0)) for y, z in rng.uniform(-0.9, 0.9, (40, 2))]
    both(host, orc, world, rays)


def test_rays_through_group_box_edges(host, orc):
    """The group bounds test (group.rs:139, bounding_box.rs:80-98) decides by tmin > tmax.  The device
    first compares uncorrected quotients and only falls back to the exact ones when they are within
    a few ulps; rays through the edges and corners of a group's box (where tmin == tmax up to
    rounding) must still take the reference's side of every such tie."""
    def world(o):
        M = o.Matrix
        g = o.Group()
        g.add_child(o.Cube(M.translation(0.3, -0.2, 0.1) * M.scaling(1.3, 0.7, 0.9),
                           o.Phong(o.Solid(o.Color(0.9, 0.2, 0.2)), 0.3, 0.7, 0.0, 200.0, 0.0, 0.0, 1.0), True))
        g.add_child(o.Sphere(M.translation(0.3, -0.2, 0.1) * M.scaling(0.5, 0.5, 0.5),
                             o.Phong(o.Solid(o.Color(0.2, 0.9, 0.2)), 0.3, 0.7, 0.0, 200.0, 0.0, 0.0, 1.0), True))
        outer = o.Group()
        outer.add_child(g)
        return o.World([outer], [o.PointLight(o.point(-10, 10, -10), o.Color.white())])
    lo = np.array([0.3 - 1.3, -0.2 - 0.7, 0.1 - 0.9])
    hi = np.array([0.3 + 1.3, -0.2 + 0.7, 0.1 + 0.9])
    rng = np.random.default_rng(5)
    origins, dirs = [], []
    for _ in range(3000):
        p = np.where(rng.random(3) < 0.5, lo, hi).astype(np.float64)
        free = rng.integers(0, 4)          # 3: a corner; else a point on the edge along that axis
        if free < 3:
            p[free] = rng.uniform(lo[free], hi[free])
        d = rng.uniform(0.2, 1.0, 3) * rng.choice([-1.0, 1.0], 3)
        d /= np.linalg.norm(d)
        o = p - rng.uniform(1.0, 5.0) * d
        o = np.nextafter(o, o + rng.choice([-1.0, 0.0, 1.0], 3))  # straddle the tie by an ulp
        origins.append(o)
        dirs.append(d)
    o, d = np.array(origins), np.array(dirs)
    got = host.color_at(world(host), o, d, remaining=1)
    ref = orc.color_at_batch(world(orc), o, d, remaining=1)
    assert np.array_equal(got != 0, ref != 0)   # the same rays hit
    assert np.abs(got - ref).max() <= 1e-11
    assert 0.02 < (ref != 0).any(axis=1).mean() < 0.98


def test_shadow_rays_skipped_only_when_they_cannot_matter(host, orc):
    """The device does not trace a shadow ray when `ambient` and `ambient + diffuse + specular`
    (phong.rs:79-115) are the same colour: light behind the surface, or no diffuse and specular
    term.  With an infinite colour channel 0 * inf is NaN, the two differ, and the ray must be traced:
    the floor below is lit NaN except in the sphere's shadow, where it is inf."""
    def world(o):
        M = o.Matrix
        glow = o.Phong(o.Solid(o.Color(float("inf"), 0.5, 0.25)), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0)
        plain = o.Phong(o.Solid(o.Color(0.2, 0.5, 0.25)), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0)
        floor = o.Plane(M.identity(), glow, True)
        slab = o.Plane(M.translation(0.0, -5.0, 0.0), plain, True)   # ambient only, bounded: never needs a shadow ray
        ball = o.Sphere(M.translation(0.0, 1.5, 0.0), o.Phong.default(), True)
        return o.World([floor, slab, ball], [o.PointLight(o.point(0, 10, 0), o.Color.white()),
                                            o.PointLight(o.point(0, -20, 0), o.Color(0.5, 0.5, 0.5))])
    W, H = 64, 48

    def camera(api):
        return api.Camera(H, W, 1.0, api.Matrix.view_transform(api.point(0, 4.0, -8), api.point(0, 0, 0),
                                                                api.vector(0, 1, 0)))
    img, st = host.render(world(host), camera(host), stats=True)
    ref, counters = orc.render(world(orc), camera(orc), threads=4, counters=True)
    assert np.array_equal(np.isnan(img), np.isnan(ref)) and np.isnan(ref).any()
    assert np.array_equal(np.isinf(img), np.isinf(ref)) and np.isinf(ref).any()
    ok = np.isfinite(ref)
    assert np.abs(img[ok] - ref[ok]).max() <= 1e-11
    # skipped rays still count: the totals are the reference's
    assert st.rays_shadow == counters["rays_shadow"]
