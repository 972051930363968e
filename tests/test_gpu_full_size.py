"""The measured configurations at their full sizes.  The oracle cannot render 8 or 33 million pixels
in a test, so parity is checked on a random sample of pixels (each traced by the oracle through
Camera::ray_for_pixel and World::color_at) plus whole-frame properties that do not depend on the
size: the clamped-byte output equals Canvas::clamp of the f64 output everywhere, and the frame
assembled from cyclic row tiles of several ranks equals the single-rank frame bit for bit."""
import numpy as np
import pytest

from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def sampled_reference(orc, world, cam, xs, ys):
    o = np.empty((len(xs), 3))
    d = np.empty((len(xs), 3))
    for k, (x, y) in enumerate(zip(xs, ys)):
        ray = orc.ray_for_pixel(cam, int(x), int(y))
        o[k] = ray.origin[:3]
        d[k] = ray.direction[:3]
    return orc.color_at_batch(world, o, d, remaining=5)


def quantise(img):
    return np.clip(np.ceil(255.0 * img), 0, 255).astype(np.int64)


def test_default_scene_3840x2160(host, orc):
    W, H = 3840, 2160
    img, st = host.render(scenes.build_world(host), scenes.default_camera(host, W, H), stats=True)
    assert img.shape == (H, W, 3) and np.isfinite(img).all()
    assert st.rays_primary == W * H
    rng = np.random.default_rng(7)
    n = 4000
    xs, ys = rng.integers(0, W, n), rng.integers(0, H, n)
    # the glass sphere is where the deep recursion happens: oversample its neighbourhood
    xs[: n // 2] = rng.integers(int(0.35 * W), int(0.65 * W), n // 2)
    ys[: n // 2] = rng.integers(int(0.25 * H), int(0.75 * H), n // 2)
    ref = sampled_reference(orc, scenes.build_world(orc), scenes.default_camera(orc, W, H), xs, ys)
    got = img[ys, xs]
    assert np.abs(got - ref).max() <= 1e-11
    assert (quantise(got) != quantise(ref)).sum() == 0
    # bytes = Canvas::clamp of the colours, everywhere
    img8 = host.render_u8(scenes.build_world(host), scenes.default_camera(host, W, H))
    assert np.array_equal(img8, quantise(img).astype(np.uint8))


def test_8k_frame_from_eight_ranks(host, orc):
    """7680x4320 as the multi-GPU runs partition it: each of 8 ranks renders its cyclic 8-row tiles;
    a sample of pixels from every rank's rows is checked against the oracle, and two ranks' buffers
    are compared bit for bit with the corresponding rows of a 2-rank partition."""
    W, H = 7680, 4320
    world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
    oworld, ocam = scenes.build_world(orc), scenes.default_camera(orc, W, H)
    rng = np.random.default_rng(11)
    parts = {}
    for rank in (0, 3, 7):
        part = host.render(world, cam, rank=rank, world_size=8)
        rows = host.global_rows(H, rank, 8)
        assert part.shape == (len(rows), W, 3)
        parts[rank] = (part, rows)
        k = rng.integers(0, len(rows), 500)
        xs = rng.integers(0, W, 500)
        ref = sampled_reference(orc, oworld, ocam, xs, rows[k])
        assert np.abs(part[k, xs] - ref).max() <= 1e-11
    # rank 0 of 8 owns tiles 0, 8, 16, ...; rank 0 of 2 owns tiles 0, 2, 4, ...: shared rows must agree exactly
    half = host.render(world, cam, rank=0, world_size=2)
    half_rows = host.global_rows(H, 0, 2)
    part, rows = parts[0]
    pos = {int(r): i for i, r in enumerate(half_rows)}
    idx = np.array([pos[int(r)] for r in rows])
    assert np.array_equal(half[idx], part)


@pytest.mark.parametrize("name", ["glass", "csg", "smooth_teapot", "dragon"])
def test_other_workloads_1920x1080(host, orc, name):
    """The glass/pattern, CSG and mesh configurations at the size they are benchmarked at: a sample
    of pixels against the oracle, and bytes = Canvas::clamp of the colours everywhere."""
    W, H = 1920, 1080
    mw, mc = scenes.WORKLOADS[name]
    img, st = host.render(mw(host), mc(host, W, H), stats=True)
    assert img.shape == (H, W, 3) and st.rays_primary == W * H
    rng = np.random.default_rng(13)
    n = 1500
    xs, ys = rng.integers(0, W, n), rng.integers(0, H, n)
    xs[: n // 2] = rng.integers(int(0.3 * W), int(0.7 * W), n // 2)   # where the objects are
    ys[: n // 2] = rng.integers(int(0.3 * H), int(0.8 * H), n // 2)
    ref = sampled_reference(orc, mw(orc), mc(orc, W, H), xs, ys)
    got = img[ys, xs]
    assert np.abs(got - ref).max() <= 1e-11
    assert (quantise(got) != quantise(ref)).sum() == 0
    img8 = host.render_u8(mw(host), mc(host, W, H))
    assert np.array_equal(img8, quantise(img).astype(np.uint8))


def test_two_frames_in_flight(host):
    """Consecutive frames on two streams, each with its own scene handle and buffer (what bench.py does
    on several GPUs so that the next frame's wavefronts fill the GPU while this frame's longest ones
    finish): every frame equals the same frame rendered alone."""
    torch = pytest.importorskip("torch")
    from raytracer_amd.distributed import OverlappedGather
    W, H = 960, 540
    dev = torch.device("cuda", 0)
    mw, mc = scenes.WORKLOADS["default"]
    worlds = [mw(host), mw(host)]
    cams = [mc(host, W, H), scenes.glass_camera(host, W, H)]   # two different views of the scene
    want = [host.render_u8(worlds[0], cam) for cam in cams]
    buffers = [torch.empty((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    pipe = OverlappedGather(None, buffers, streams)
    for pair in range(4):
        order = [pair % 2, 1 - pair % 2]   # which view goes to which slot alternates

        def render(buf, slot):
            host.render_device(worlds[slot], cams[order[slot]], buf.data_ptr(),
                               stream=torch.cuda.current_stream().cuda_stream, device=0, u8=True)
        pipe.step(render)
        pipe.step(render)
        pipe.wait()
        for slot in range(2):
            assert np.array_equal(buffers[slot].cpu().numpy(), want[order[slot]]), (pair, slot)
