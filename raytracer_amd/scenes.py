"""Scene definitions for the measured configurations, written against the mirror interface.

Every function takes an `Api` (see mirror.py) so that the same description can be built on the
host library (for the HIP path) and on the test oracle.
"""
import math
import os


def build_world(api):
    """The default scene of the reference's binary (src/lib.rs:135-323): a white ambient-only
    backdrop plane and one group of a glass sphere plus seventeen cubes, `divide(1)`, two lights."""
    M, Phong, Solid, Color = api.Matrix, api.Phong, api.Solid, api.Color

    def matte(color):
        return Phong(Solid(color), 0.1, 0.7, 0.0, 200.0, 0.1, 0.0, 1.0)

    white = matte(Color.white())
    blue = matte(Color(0.537, 0.831, 0.914))
    red = matte(Color(0.941, 0.322, 0.388))
    purple = matte(Color(0.373, 0.404, 0.550))

    standard = M.scaling(0.5, 0.5, 0.5) * M.translation(1.0, -1.0, 1.0)
    large = M.scaling(3.5, 3.5, 3.5) * standard
    medium = M.scaling(3.0, 3.0, 3.0) * standard
    small = M.scaling(2.0, 2.0, 2.0) * standard

    plane = api.Plane(
        M.translation(0.0, 0.0, 500.0) * M.rotation_x(math.pi / 2.0),
        Phong(Solid(Color.white()), 1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0),
        True,
    )

    group = api.Group()
    glass = Phong(Solid(Color(0.373, 0.404, 0.550)), 0.0, 0.2, 1.0, 200.0, 0.7, 0.7, 1.5)
    cubes = [
        ((4.0, 0.0, 0.0), medium, white),
        ((8.5, 1.5, -0.5), large, blue),
        ((0.0, 0.0, 4.0), large, red),
        ((4.0, 0.0, 4.0), small, white),
        ((7.5, 0.5, 4.0), medium, purple),
        ((-0.25, 0.25, 8.0), medium, white),
        ((4.0, 1.0, 7.5), large, blue),
        ((10.0, 2.0, 7.5), medium, red),
        ((8.0, 2.0, 12.0), small, white),
        ((20.0, 1.0, 9.0), small, white),
        ((-0.5, -5.0, 0.25), large, blue),
        ((4.0, -4.0, 0.0), large, red),
        ((8.5, -4.0, 0.0), large, white),
        ((0.0, -4.0, 4.0), large, white),
        ((-0.5, -4.5, 8.0), large, purple),
        ((0.0, -8.0, 4.0), large, white),
        ((-0.5, 8.5, 8.0), large, white),
    ]
    children = [api.Sphere(large, glass, True)]
    for (tx, ty, tz), size, mat in cubes:
        children.append(api.Cube(M.translation(tx, ty, tz) * size, mat, True))
    group.add_children(children)
    group.divide(1)

    lights = [
        api.PointLight(api.point(50.0, 100.0, -50.0), Color(1.0, 1.0, 1.0)),
        api.PointLight(api.point(-400.0, 50.0, -10.0), Color(0.2, 0.2, 0.2)),
    ]
    return api.World([plane, group], lights)


def default_camera(api, width, height):
    """The camera `run` builds (src/lib.rs:59-68).  Note the argument order: the reference passes
    (height, width) as (hsize, vsize), and the canvas is camera.width() x camera.height() =
    vsize x hsize, so the image is `width` pixels wide."""
    M = api.Matrix
    return api.Camera(
        height,
        width,
        0.785,
        M.view_transform(api.point(-6.0, 6.0, -10.0), api.point(6.0, 0.0, 6.0),
                         api.vector(-0.45, 1.0, 0.0)),
    )


def glass_world(api):
    """A reflective + refractive scene that exercises every pattern and the quadrics: glass sphere
    with an air bubble, mirror sphere with a ring pattern, Perlin-perturbed gradient, blended
    pattern, closed cylinder and cone, checkered reflective floor.  Build-defined (the reference
    ships no such scene); its oracle is the CPU restatement."""
    M, Phong, Solid, Color = api.Matrix, api.Phong, api.Solid, api.Color
    floor = api.Plane(
        M.identity(),
        Phong(api.Checker(Solid(Color.white()), Solid(Color(0.15, 0.15, 0.15)), M.scaling(1, 1, 1)),
              0.1, 0.9, 0.0, 200.0, 0.3, 0.0, 1.0),
        True,
    )
    wall = api.Plane(
        M.translation(0.0, 0.0, 10.0) * M.rotation_x(math.pi / 2.0),
        Phong(api.Striped(Solid(Color(0.9, 0.3, 0.3)), Solid(Color(0.3, 0.3, 0.9)),
                          M.scaling(0.5, 0.5, 0.5)),
              1.0, 0.0, 0.0, 200.0, 0.0, 0.0, 1.0),
        True,
    )
    glass = api.Sphere(
        M.translation(0.0, 1.0, 0.0),
        Phong(Solid(Color(0.05, 0.05, 0.05)), 0.0, 0.1, 1.0, 300.0, 0.9, 0.9, 1.5),
        True,
    )
    bubble = api.Sphere(
        M.translation(0.0, 1.0, 0.0) * M.scaling(0.5, 0.5, 0.5),
        Phong(Solid(Color.white()), 0.0, 0.0, 0.0, 200.0, 0.0, 1.0, 1.0),
        True,
    )
    mirror = api.Sphere(
        M.translation(2.5, 1.0, 1.5),
        Phong(api.Ring(Solid(Color(0.8, 0.8, 0.2)), Solid(Color(0.2, 0.6, 0.2)),
                       M.scaling(0.25, 0.25, 0.25)),
              0.1, 0.6, 0.9, 200.0, 0.8, 0.0, 1.0),
        True,
    )
    noisy = api.Sphere(
        M.translation(-2.5, 1.0, 1.5),
        Phong(api.Perturbed(api.Gradient(Solid(Color.red()), Solid(Color.blue()),
                                         M.translation(-1.0, 0.0, 0.0) * M.scaling(2, 2, 2)),
                            0.2, M.identity()),
              0.1, 0.8, 0.3, 100.0, 0.1, 0.0, 1.0),
        True,
    )
    blend = api.Sphere(
        M.translation(-1.0, 0.4, -2.0) * M.scaling(0.4, 0.4, 0.4),
        Phong(api.Blended(api.Striped(), api.Checker(), M.identity()),
              0.1, 0.8, 0.3, 100.0, 0.0, 0.0, 1.0),
        True,
    )
    cyl = api.Cylinder(M.translation(1.5, 0.0, -2.0), api.Phong.default(), True, 0.0, 1.0, True)
    cone = api.Cone(M.translation(3.5, 1.0, -1.0), api.Phong.default(), True, -1.0, 0.0, True)
    lights = [
        api.PointLight(api.point(-10.0, 10.0, -10.0), Color.white()),
        api.PointLight(api.point(5.0, 8.0, -6.0), Color(0.3, 0.3, 0.3)),
    ]
    return api.World([floor, wall, glass, bubble, mirror, noisy, blend, cyl, cone], lights)


def glass_camera(api, width, height):
    M = api.Matrix
    return api.Camera(height, width, math.pi / 3.0,
                      M.view_transform(api.point(0.0, 2.5, -8.0), api.point(0.0, 1.0, 0.0),
                                       api.vector(0.0, 1.0, 0.0)))


OBJ_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                       "obj")


def mesh_world(api, variant="teapot", obj_dir=OBJ_DIR, divide=True):
    """An OBJ mesh (the reference's tests/obj_files fixtures) parsed like
    src/scene/obj_file_parser.rs does, grouped, `divide(1)` (as src/lib.rs:308 does for its group),
    over a floor plane with one light.  Variants: teapot (flat triangles), smooth_teapot (smooth
    triangles, Z-up so it is rotated and scaled), dragon (smooth triangles)."""
    M, Phong, Solid, Color = api.Matrix, api.Phong, api.Solid, api.Color
    mat = Phong(Solid(Color(0.8, 0.33, 0.2)), 0.1, 0.7, 0.3, 200.0, 0.0, 0.0, 1.0)
    if variant == "smooth_teapot":
        t = M.rotation_x(-math.pi / 2.0) * M.scaling(0.2, 0.2, 0.2)
    else:
        t = M.identity()
    parser = api.ObjFileParser.parse_obj_file(os.path.join(obj_dir, variant + ".obj"), M.identity(),
                                              mat, True)
    group = parser.obj_to_group(t)
    if divide:
        group.divide(1)
    floor = api.Plane(M.translation(0.0, -0.2, 0.0), Phong.default(), True)
    lights = [api.PointLight(api.point(-10.0, 10.0, -10.0), Color.white())]
    return api.World([floor, group], lights)


def mesh_camera(api, width, height):
    M = api.Matrix
    return api.Camera(height, width, math.pi / 3.0,
                      M.view_transform(api.point(0.0, 3.0, -9.0), api.point(0.0, 1.5, 0.0),
                                       api.vector(0.0, 1.0, 0.0)))


def csg_world(api):
    """Constructive solid geometry (src/geometry/csg.rs): a die (cube ∩ sphere, minus three
    cylinders through its faces), a glass lens (sphere ∩ sphere) and a union of a cone and a sphere,
    on a checkered floor.  Build-defined (the reference's binary renders no CSG shape); its oracle
    is the CPU restatement."""
    M, Phong, Solid, Color = api.Matrix, api.Phong, api.Solid, api.Color
    floor = api.Plane(M.identity(),
                      Phong(api.Checker(Solid(Color.white()), Solid(Color(0.2, 0.2, 0.25)), M.identity()),
                            0.1, 0.9, 0.0, 200.0, 0.2, 0.0, 1.0), True)
    ivory = Phong(Solid(Color(0.9, 0.85, 0.7)), 0.1, 0.8, 0.4, 80.0, 0.05, 0.0, 1.0)
    red = Phong(Solid(Color(0.8, 0.1, 0.1)), 0.1, 0.8, 0.3, 50.0, 0.0, 0.0, 1.0)
    body = api.CSG(api.INTERSECTION, api.Cube(M.identity(), ivory, True),
                   api.Sphere(M.scaling(1.4, 1.4, 1.4), ivory, True))
    bores = api.Group()
    for rot in (M.identity(), M.rotation_x(math.pi / 2.0), M.rotation_z(math.pi / 2.0)):
        bores.add_child(api.Cylinder(rot * M.scaling(0.45, 1.0, 0.45), red, True, -1.5, 1.5, True))
    die = api.CSG(api.DIFFERENCE, body, bores,
                  M.translation(-1.6, 1.0, 0.0) * M.rotation_y(0.6) * M.rotation_x(0.3))
    glass = Phong(Solid(Color(0.05, 0.08, 0.05)), 0.0, 0.1, 1.0, 300.0, 0.6, 0.9, 1.5)
    lens = api.CSG(api.INTERSECTION, api.Sphere(M.translation(0.0, 0.0, -0.6), glass, True),
                   api.Sphere(M.translation(0.0, 0.0, 0.6), glass, True),
                   M.translation(1.4, 1.0, -1.0) * M.rotation_y(-0.4))
    blue = Phong(Solid(Color(0.2, 0.3, 0.9)), 0.1, 0.8, 0.5, 100.0, 0.1, 0.0, 1.0)
    top = api.CSG(api.UNION, api.Cone(M.translation(0.0, 1.0, 0.0), blue, True, -1.0, 0.0, True),
                  api.Sphere(M.translation(0.0, 1.0, 0.0) * M.scaling(0.4, 0.4, 0.4), blue, True),
                  M.translation(2.2, 0.0, 2.0))
    lights = [api.PointLight(api.point(-8.0, 10.0, -8.0), Color.white()),
              api.PointLight(api.point(6.0, 6.0, -4.0), Color(0.3, 0.3, 0.3))]
    return api.World([floor, die, lens, top], lights)


def field_world(api, side=16):
    """side x side small spheres, cubes and closed cylinders standing on a checkered floor, all
    added to the world directly (no groups), the way scenes are usually written against
    World::new; every tenth one is a mirror, every seventh is glass.  Build-defined; its oracle is
    the CPU restatement."""
    M, Phong, Solid, Color = api.Matrix, api.Phong, api.Solid, api.Color
    floor = api.Plane(M.identity(),
                      Phong(api.Checker(Solid(Color(0.9, 0.9, 0.9)), Solid(Color(0.3, 0.3, 0.35)), M.identity()),
                            0.1, 0.9, 0.0, 200.0, 0.1, 0.0, 1.0), True)
    objects = [floor]
    for k in range(side * side):
        i, j = k % side, k // side
        x, z = (i - (side - 1) / 2.0) * 1.2, (j - (side - 1) / 2.0) * 1.2
        r = 0.3 + 0.15 * ((k * 7) % 5) / 4.0
        colour = Color(0.25 + 0.7 * ((k * 3) % 7) / 6.0, 0.25 + 0.7 * ((k * 5) % 11) / 10.0, 0.25 + 0.7 * ((k * 2) % 5) / 4.0)
        if k % 10 == 3:
            mat = Phong(Solid(colour), 0.1, 0.4, 0.9, 200.0, 0.7, 0.0, 1.0)
        elif k % 7 == 5:
            mat = Phong(Solid(Color(0.05, 0.05, 0.08)), 0.0, 0.1, 1.0, 300.0, 0.5, 0.9, 1.5)
        else:
            mat = Phong(Solid(colour), 0.1, 0.8, 0.3, 80.0, 0.0, 0.0, 1.0)
        kind = k % 3
        if kind == 0:
            objects.append(api.Sphere(M.translation(x, r, z) * M.scaling(r, r, r), mat, True))
        elif kind == 1:
            objects.append(api.Cube(M.translation(x, r, z) * M.rotation_y(0.1 * k) * M.scaling(r, r, r), mat, True))
        else:
            objects.append(api.Cylinder(M.translation(x, 0.0, z) * M.scaling(r, 1.0, r), mat, True, 0.0, 2.0 * r, True))
    lights = [api.PointLight(api.point(-12.0, 14.0, -10.0), Color.white()),
              api.PointLight(api.point(8.0, 9.0, -12.0), Color(0.3, 0.3, 0.3))]
    return api.World(objects, lights)


def field_camera(api, width, height):
    M = api.Matrix
    # aimed below and beside the field: with (height, width) in the order src/lib.rs:59-61 passes them,
    # a wide frame's centre is up and to the right of the view axis
    return api.Camera(height, width, math.pi / 3.0,
                      M.view_transform(api.point(0.0, 11.0, -15.0), api.point(-4.5, -8.0, -2.0),
                                       api.vector(0.0, 1.0, 0.0)))


def loose_triangles_world(api, n_grouped=20000, n_loose=300, seed=41):
    """Small triangles in random order: n_grouped as direct children of one undivided group and
    n_loose more directly in the world.  The reference tests each of them against every ray; for
    the device library this is the scene whose traversal records take longest to lay out (long runs of
    siblings to cut and sort).  Build-defined; its oracle is the CPU restatement."""
    import numpy as np
    rng = np.random.default_rng(seed)
    n = n_grouped + n_loose
    centres = rng.uniform(-4, 4, (n, 3))
    offs = rng.uniform(-0.15, 0.15, (n, 3, 3))
    mat = api.Phong(api.Solid(api.Color(0.7, 0.5, 0.3)), 0.2, 0.7, 0.2, 50.0, 0.1, 0.0, 1.0)
    g = api.Group()
    loose = []
    for k in range(n):
        pts = [api.point(*(centres[k] + offs[k, j])) for j in range(3)]
        t = api.Triangle(*pts, None, mat, True)
        if k < n_grouped:
            g.add_child(t)
        else:
            loose.append(t)
    return api.World([g] + loose, [api.PointLight(api.point(-10, 10, -10), api.Color.white())])


WORKLOADS = {
    "field": (field_world, field_camera),
    "default": (build_world, default_camera),
    "glass": (glass_world, glass_camera),
    "teapot": (lambda api: mesh_world(api, "teapot"), mesh_camera),
    "smooth_teapot": (lambda api: mesh_world(api, "smooth_teapot"), mesh_camera),
    "dragon": (lambda api: mesh_world(api, "dragon"), mesh_camera),
    "csg": (csg_world, glass_camera),
}
