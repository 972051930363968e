"""Edges of the render path on the device: empty and degenerate inputs, ragged frames, ranks without
rows, the documented limits."""
import numpy as np
import pytest

from raytracer_amd import abi, scenes

pytestmark = pytest.mark.gpu


def camera(api, W, H):
    return api.Camera(H, W, 1.0, api.Matrix.view_transform(api.point(0, 1.5, -6), api.point(0, 0.5, 0), api.vector(0, 1, 0)))


def some_objects(api):
    M = api.Matrix
    return [api.Sphere(M.translation(0, 1, 0), api.Phong.default(), True),
            api.Plane(M.identity(), api.Phong.default(), True)]


def light(api, k=0):
    return api.PointLight(api.point(-10 + 5 * k, 10, -10), api.Color(0.5, 0.5, 0.5))


def test_world_without_objects(host, orc):  # World::color_at with no hit is black (world.rs:198-212)
    img = host.render(host.World([], [light(host)]), camera(host, 20, 12))
    ref = orc.render(orc.World([], [light(orc)]), camera(orc, 20, 12))
    assert img.shape == (12, 20, 3) and not img.any() and np.array_equal(img, ref)


def test_world_without_lights(host, orc):   # shade_hit sums over the lights (world.rs:170-195): nothing to sum
    img = host.render(host.World(some_objects(host), []), camera(host, 20, 12))
    ref = orc.render(orc.World(some_objects(orc), []), camera(orc, 20, 12))
    assert not img.any() and np.array_equal(img, ref)


@pytest.mark.parametrize("size", [(1, 1), (8, 8), (9, 9), (7, 64), (65, 3)])
def test_small_and_ragged_frames(host, orc, size):
    W, H = size
    make = lambda api: api.World(some_objects(api), [light(api)])
    img = host.render(make(host), camera(host, W, H))
    ref = orc.render(make(orc), camera(orc, W, H))
    assert img.shape == (H, W, 3)
    assert np.abs(img - ref).max() <= 1e-11
    u8 = host.render_u8(make(host), camera(host, W, H))
    assert np.array_equal(u8, np.vectorize(host.clamp, otypes=[np.uint8])(img))


def test_ranks_without_rows(host):
    """A 16-row frame has two tiles of rows; with eight ranks, six of them own nothing."""
    make = lambda api: api.World(some_objects(api), [light(api)])
    whole = host.render(make(host), camera(host, 24, 16))
    for rank in range(8):
        part = host.render(make(host), camera(host, 24, 16), rank=rank, world_size=8)
        rows = host.global_rows(16, rank, 8)
        assert part.shape == (len(rows), 24, 3)
        assert len(rows) == (8 if rank < 2 else 0)
        if len(rows):
            assert np.array_equal(part, whole[rows])


def test_depth_zero_and_limits(host, orc):
    make = lambda api, n: api.World(some_objects(api), [light(api, k) for k in range(n)])
    most = abi.RT_MAX_LIGHTS
    img = host.render(make(host, most), camera(host, 32, 20), max_depth=0)   # as many lights as are allowed
    ref = orc.render(make(orc, most), camera(orc, 32, 20), remaining=0)
    assert np.abs(img - ref).max() <= 1e-11
    with pytest.raises(abi.RtError):
        host.render(make(host, most + 1), camera(host, 32, 20))
    with pytest.raises(abi.RtError):
        host.render(make(host, 1), camera(host, 32, 20), max_depth=6)
