"""The measured configurations at their full sizes, whole frames: the oracle renders a 3840x2160
frame in seconds on the GPU box's host cores, so every pixel of the benchmarked frames is compared
(colours to 1e-11, identical clamped bytes, identical ray counts), plus properties that do not
depend on the size: the clamped-byte output equals Canvas::clamp of the f64 output everywhere, and
the frame assembled from cyclic row tiles of several ranks equals the single-rank frame bit for bit."""
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
    """The headline frame, every one of its 8,294,400 pixels against the oracle."""
    W, H = 3840, 2160
    img, st = host.render(scenes.build_world(host), scenes.default_camera(host, W, H), stats=True)
    assert img.shape == (H, W, 3) and np.isfinite(img).all()
    assert st.rays_primary == W * H
    ref, cnt = orc.render(scenes.build_world(orc), scenes.default_camera(orc, W, H), threads=16, counters=True)
    assert np.abs(img - ref).max() <= 1e-11
    assert np.array_equal(quantise(img), quantise(ref))
    assert (st.rays_primary, st.rays_shadow, st.rays_reflect, st.rays_refract) == \
        (cnt["rays_primary"], cnt["rays_shadow"], cnt["rays_reflect"], cnt["rays_refract"])
    # bytes = Canvas::clamp of the colours, everywhere
    img8 = host.render_u8(scenes.build_world(host), scenes.default_camera(host, W, H))
    assert np.array_equal(img8, quantise(img).astype(np.uint8))


def test_8k_frame_from_eight_ranks(host, orc):
    """7680x4320 as the multi-GPU runs partition it: each of 8 ranks renders its cyclic 8-row tiles
    (here one after the other on one GPU); the rows of all ranks together are the oracle's frame,
    all 33,177,600 pixels of it, and a 2-rank partition gives the same rows bit for bit."""
    W, H = 7680, 4320
    world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
    ref = orc.render(scenes.build_world(orc), scenes.default_camera(orc, W, H), threads=16)
    seen = np.zeros(H, dtype=bool)
    part0 = rows0 = None
    for rank in range(8):
        part = host.render(world, cam, rank=rank, world_size=8)
        rows = host.global_rows(H, rank, 8)
        assert part.shape == (len(rows), W, 3) and not seen[rows].any()
        seen[rows] = True
        assert np.abs(part - ref[rows]).max() <= 1e-11
        assert np.array_equal(quantise(part), quantise(ref[rows]))
        if rank == 0:
            part0, rows0 = part, rows
    assert seen.all()
    # rank 0 of 8 owns tiles 0, 8, 16, ...; rank 0 of 2 owns tiles 0, 2, 4, ...: shared rows must agree exactly
    half = host.render(world, cam, rank=0, world_size=2)
    half_rows = host.global_rows(H, 0, 2)
    pos = {int(r): i for i, r in enumerate(half_rows)}
    idx = np.array([pos[int(r)] for r in rows0])
    assert np.array_equal(half[idx], part0)


@pytest.mark.parametrize("name", ["glass", "csg", "field", "teapot", "smooth_teapot", "dragon"])
def test_other_workloads_1920x1080(host, orc, name):
    """The glass/pattern, CSG, loose-object and mesh configurations at the size they are benchmarked
    at: every pixel and the ray counts against the oracle, and bytes = Canvas::clamp of the colours."""
    W, H = 1920, 1080
    mw, mc = scenes.WORKLOADS[name]
    img, st = host.render(mw(host), mc(host, W, H), stats=True)
    assert img.shape == (H, W, 3) and st.rays_primary == W * H
    ref, cnt = orc.render(mw(orc), mc(orc, W, H), threads=16, counters=True)
    assert np.abs(img - ref).max() <= 1e-11
    assert np.array_equal(quantise(img), quantise(ref))
    assert (st.rays_shadow, st.rays_reflect, st.rays_refract) == \
        (cnt["rays_shadow"], cnt["rays_reflect"], cnt["rays_refract"])
    img8 = host.render_u8(mw(host), mc(host, W, H))
    assert np.array_equal(img8, quantise(img).astype(np.uint8))


@pytest.mark.parametrize("name", ["glass", "dragon", "csg", "teapot", "smooth_teapot"])
def test_glass_and_mesh_3840x2160(host, orc, name):
    """The glass/pattern scene, the CSG scene and the meshes at the 3840x2160 their timings are quoted at."""
    W, H = 3840, 2160
    mw, mc = scenes.WORKLOADS[name]
    img, st = host.render(mw(host), mc(host, W, H), stats=True)
    assert img.shape == (H, W, 3) and st.rays_primary == W * H
    ref, cnt = orc.render(mw(orc), mc(orc, W, H), threads=16, counters=True)
    assert np.abs(img - ref).max() <= 1e-11
    assert np.array_equal(quantise(img), quantise(ref))
    assert (st.rays_shadow, st.rays_reflect, st.rays_refract) == \
        (cnt["rays_shadow"], cnt["rays_reflect"], cnt["rays_refract"])


@pytest.mark.parametrize("name", ["default", "glass", "csg", "field", "teapot", "smooth_teapot", "dragon"])
def test_ppm_of_every_configuration_1000x1000(host, orc, name):
    """The parity gate as the metric words it, at the reference's own 1000x1000: the PPM text the
    device renders and formats (rt_render_ppm) against Canvas::to_ppm of the oracle's frame.  The
    gate allows 0.01 % of the pixels to differ by 1/255 per channel where ties flip; none do, so the
    texts are compared byte for byte (header, values and the 70-column line breaks of canvas.rs:62-126)."""
    W, H = 1000, 1000
    mw, mc = scenes.WORKLOADS[name]
    text = host.render_ppm(mw(host), mc(host, W, H))
    ref = orc.render(mw(orc), mc(orc, W, H), threads=16)
    assert text.startswith(b"P3\n1000 1000\n255\n")
    assert text == orc.to_ppm(ref)


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


def test_ragged_frame_with_cost_ordered_tiles(host, orc):
    """1003x517: neither side a multiple of the 8-pixel tile, and enough tiles (8190) for the cost
    pre-pass and the reordered dispatch; every pixel against the oracle."""
    W, H = 1003, 517
    mw, mc = scenes.WORKLOADS["default"]
    img = host.render(mw(host), mc(host, W, H))
    ref = orc.render(mw(orc), mc(orc, W, H), threads=16)
    assert img.shape == ref.shape == (H, W, 3)
    assert np.abs(img - ref).max() <= 1e-11
    assert np.array_equal(quantise(img), quantise(ref))


def test_rows_of_a_40960x23040_frame(host, orc):
    """Far beyond the benchmarked sizes: 943,718,400 pixels (a 22.6 GB f64 frame) as 720 ranks would share it.
    Three of the ranks' tiles, 1.3 M pixels each, rendered here; sampled pixels against the oracle's rays
    through the same pixels (pixel coordinates up to 40959 x 23039, pixel indices beyond 2^29)."""
    W, H, N = 40960, 23040, 720
    world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
    oworld, ocam = scenes.build_world(orc), scenes.default_camera(orc, W, H)
    rng = np.random.default_rng(5)
    for rank in (0, 333, N - 1):
        rows = np.asarray(host.global_rows(H, rank, N))
        part = host.render(world, cam, rank=rank, world_size=N)
        assert part.shape == (len(rows), W, 3) and len(rows) == 32 and np.isfinite(part).all()
        k = rng.integers(0, len(rows), 1500)
        x = rng.integers(0, W, 1500)
        x[:4] = (0, W - 1, 0, W - 1)
        k[:4] = (0, 0, len(rows) - 1, len(rows) - 1)
        ref = sampled_reference(orc, oworld, ocam, x, rows[k])
        got = part[k, x]
        assert np.abs(got - ref).max() <= 1e-11
        assert np.array_equal(quantise(got), quantise(ref))
    assert rows[-1] == H - 1  # the last rank owns the frame's last tile
