"""The benchmarked scenes seen from random viewpoints: cameras inside the scene, close to surfaces,
far away and looking straight along an axis, random fields of view and ragged frame shapes, depths 0
to 5; every pixel and the ray counts against the oracle.  tools/fuzz_views.py runs the same recipe for
as many seeds as one likes."""
import numpy as np
import pytest

from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def random_view(api, key):
    """A camera drawn from the seed `key` (the same for the host library and the oracle)."""
    rng = np.random.default_rng(key)
    W, H = int(rng.integers(17, 200)), int(rng.integers(9, 120))
    dist = float(10.0 ** rng.uniform(-0.5, 2.0))
    o = rng.normal(size=3)
    o = o / np.linalg.norm(o) * dist
    if rng.random() < 0.7:
        o[1] = abs(o[1])  # mostly above the floor
    target = rng.uniform(-3, 3, 3) * np.array([1.0, 0.5, 1.0]) + np.array([0.0, 1.0, 0.0])
    if rng.random() < 0.15:
        axis = int(rng.integers(0, 3))  # straight along an axis: direction components that are exact zeros
        target = o.copy()
        target[axis] -= np.sign(o[axis]) if o[axis] != 0 else 1.0
    up = (0.0, 1.0, 0.0) if abs(target[0] - o[0]) + abs(target[2] - o[2]) > 1e-9 else (0.0, 0.0, 1.0)
    fov = float(rng.uniform(0.2, 2.4))
    return api.Camera(H, W, fov, api.Matrix.view_transform(api.point(*o), api.point(*target), api.vector(*up))), (W, H)


def compare_view(host, orc, host_world, orc_world, name, seed):
    """Returns None when the frame and the ray counts agree, else a description of the difference."""
    depth = seed % 6
    key = [seed, sum(map(ord, name))]
    hc, size = random_view(host, key)
    oc, _ = random_view(orc, key)
    img, st = host.render(host_world, hc, max_depth=depth, stats=True)
    ref, cnt = orc.render(orc_world, oc, remaining=depth, threads=16, counters=True)
    finite = np.isfinite(ref)
    same_mask = np.array_equal(np.isfinite(img), finite)
    err = np.abs(img[finite] - ref[finite]) / np.maximum(1.0, np.abs(ref[finite]))
    worst = err.max() if err.size else 0.0
    rays = (st.rays_primary, st.rays_shadow, st.rays_reflect, st.rays_refract) == \
        (cnt["rays_primary"], cnt["rays_shadow"], cnt["rays_reflect"], cnt["rays_refract"])
    if same_mask and worst <= 1e-9 and rays:
        return None
    return "seed %d %s %dx%d depth %d: max rel err %.3e, finite masks equal %s, ray counts equal %s" % (
        seed, name, size[0], size[1], depth, worst, same_mask, rays)


@pytest.mark.parametrize("name", sorted(scenes.WORKLOADS))
def test_random_views_of_the_benchmarked_scenes(host, orc, name):
    mw, _ = scenes.WORKLOADS[name]
    host_world, orc_world = mw(host), mw(orc)
    for seed in range(12):
        assert compare_view(host, orc, host_world, orc_world, name, seed) is None
