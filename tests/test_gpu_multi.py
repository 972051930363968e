"""One frame on several ranks from one call (rt_multi_create / rt_render_multi*), the frame's way to
host memory (page-locked or ordinary destination), and scene handles across devices.

A one-GPU box exercises the several-rank code with ranks that share the GPU: the partition into
cyclic 8-row tiles, every rank's launch on its own stream and scene copy, the strided placement of
each rank's tiles in host memory and the gather kernel that assembles the frame on rank 0's GPU are
the same code; only the peer links are missing."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi, scenes

pytestmark = pytest.mark.gpu


def test_multi_with_one_rank_is_rt_render(host):
    """The wiring of the ABI: n_ranks = 1 must give rt_render's bytes and ray counts."""
    W, H = 640, 360
    world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
    want, st = host.render(world, cam, stats=True)
    multi = host.multi(world, 1)
    got, mst = multi.render(cam, stats=True)
    assert np.array_equal(got, want)
    assert (mst.rays_primary, mst.rays_shadow, mst.rays_reflect, mst.rays_refract) == \
        (st.rays_primary, st.rays_shadow, st.rays_reflect, st.rays_refract)
    assert mst.kernel_ms > 0 and mst.total_ms >= mst.exchange_ms >= 0
    assert np.array_equal(multi.render(cam, u8=True), host.render_u8(world, cam))
    multi.close()


@pytest.mark.parametrize("ranks", [2, 3, 8])
@pytest.mark.parametrize("size", [(1003, 517), (256, 16)])
def test_ranks_sharing_the_gpu_assemble_the_same_frame(host, ranks, size):
    """Ragged frame (neither side a multiple of 8, last tile short) and a frame with fewer tiles
    than ranks; f64 and bytes; ordinary and page-locked destination."""
    W, H = size
    world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
    want, st = host.render(world, cam, stats=True)
    want8 = host.render_u8(world, cam)
    multi = host.multi(world, ranks, devices=[0] * ranks)
    got, mst = multi.render(cam, stats=True)
    assert np.array_equal(got, want)
    assert mst.total_rays() == st.total_rays()
    assert np.array_equal(multi.render(cam, u8=True), want8)
    pin = host.pinned_empty((H, W, 3), np.float64)
    pin.array.fill(np.nan)
    assert np.array_equal(multi.render(cam, out=pin.array), want)
    pin8 = host.pinned_empty((H, W, 3), np.uint8)
    assert np.array_equal(multi.render(cam, u8=True, out=pin8.array), want8)
    pin.close()
    pin8.close()
    multi.close()


@pytest.mark.parametrize("u8", [False, True])
def test_frame_assembled_on_rank_0s_gpu(host, u8):
    torch = pytest.importorskip("torch")
    # the last tile is short; at 1003x517 the byte frame's last tile (and the whole frame) is not a
    # whole number of 4-byte words, so the rank that owns it ends on one to three single bytes
    for W, H in ((1000, 517), (1003, 517)):
        world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
        want = host.render_u8(world, cam) if u8 else host.render(world, cam)
        for ranks in (1, 3):
            multi = host.multi(world, ranks, devices=[0] * ranks)
            frame = torch.zeros((H, W, 3), dtype=torch.uint8 if u8 else torch.float64, device="cuda:0")
            guard = torch.zeros(64, dtype=torch.uint8, device="cuda:0")  # (a neighbour that must stay untouched)
            st = multi.render_device(cam, frame.data_ptr(), u8=u8)
            assert np.array_equal(frame.cpu().numpy(), want)
            assert not guard.any()
            assert st.rays_primary == W * H and st.exchange_ms >= 0
            multi.close()
        host.reset()


def test_large_frame_through_the_staging_ring_and_straight_to_pinned_memory(host):
    """1920x1080 f64 is 50 MB: several laps of the 4 x 8 MB ring for an ordinary destination."""
    W, H = 1920, 1080
    mw, mc = scenes.WORKLOADS["glass"]
    world, cam = mw(host), mc(host, W, H)
    want = host.render(world, cam)
    assert np.isfinite(want).all() and want.any()
    pin = host.pinned_empty((H, W, 3), np.float64)
    prm = abi.rt_params(5, 0, 1, 0)
    scene = host.scene_handle(world, 0)
    block = host.camera_block(cam)
    rc = host.abi.rt_render(scene, C.byref(block), C.byref(prm), pin.array.ctypes.data_as(C.POINTER(C.c_double)), None)
    assert rc == 0
    assert np.array_equal(pin.array, want)
    # an unaligned window into ordinary memory
    raw = np.zeros(H * W * 3 + 1, dtype=np.float64)
    out = raw[1:].reshape(H, W, 3)
    rc = host.abi.rt_render(scene, C.byref(block), C.byref(prm), out.ctypes.data_as(C.POINTER(C.c_double)), None)
    assert rc == 0 and np.array_equal(out, want) and raw[0] == 0.0
    text = host.render_ppm(world, cam)
    assert text == host.to_ppm(want)
    pin.close()


def test_scene_handle_survives_changes_to_other_worlds_and_to_its_own(host):
    """A raw scene handle taken from a world keeps describing that world as it was: building or
    changing other worlds, and even changing this one, must not leave it dangling."""
    W, H = 160, 96
    world = scenes.build_world(host)
    cam = scenes.default_camera(host, W, H)
    want = host.render(world, cam)
    handle = host.scene_handle(world, 0)
    block = host.camera_block(cam)
    prm = abi.rt_params(5, 0, 1, 0)

    def through_handle():
        out = np.empty((H, W, 3))
        assert host.abi.rt_render(handle, C.byref(block), C.byref(prm), out.ctypes.data_as(C.POINTER(C.c_double)), None) == 0
        return out

    other = scenes.csg_world(host)              # rth_csg, rth_group_add_child on shapes of another world
    g = host.Group()
    g.add_child(host.Sphere())
    g.divide(1)
    host.render(other, scenes.glass_camera(host, 64, 40))
    assert np.array_equal(through_handle(), want)
    # the handle's own world gets another light: new renders see it, the old handle still works
    M = host.Matrix
    grp = world.objects[1]
    grp.add_child(host.Sphere(M.translation(0, 3, 0), host.Phong.default(), True))
    changed = host.render(world, cam)
    assert not np.array_equal(changed, want)
    assert np.array_equal(through_handle(), want)


def test_same_world_on_two_devices(host):
    """render(device=0) then render(device=1) must run on the GPU asked for."""
    if host.abi.rt_device_count() < 2:
        pytest.skip("needs two GPUs")
    torch = pytest.importorskip("torch")
    W, H = 320, 200
    world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
    a = host.render(world, cam, device=0)
    b = host.render(world, cam, device=1)
    assert np.array_equal(a, b)
    buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda:1")
    host.render_device(world, cam, buf.data_ptr(), device=1)
    torch.cuda.synchronize(1)
    assert np.array_equal(buf.cpu().numpy(), a)
    multi = host.multi(world, 2)
    assert np.array_equal(multi.render(cam), a)
    frame = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda:0")
    multi.render_device(cam, frame.data_ptr())
    assert np.array_equal(frame.cpu().numpy(), a)
    multi.close()


@pytest.mark.parametrize("width", [64, 12, 7680])
def test_rank_0_places_byte_rows_in_frame_order_on_the_device(host, width):
    """What rank 0 of the one-process-per-GPU gather does with the rows it has received, on device
    tensors: widths whose rows are whole 8-byte words are moved as words, others byte by byte."""
    torch = pytest.importorskip("torch")
    from raytracer_amd.distributed import FrameGather
    dev = torch.device("cuda", 0)
    H, ranks = 45 if width < 100 else 4320, 3
    full = torch.randint(0, 256, (H, width, 3), dtype=torch.uint8, device=dev)
    g = FrameGather(host, width, H, 0, ranks, dev, dtype=torch.uint8)
    for r in range(1, ranks):
        rows = torch.as_tensor(np.asarray(host.global_rows(H, r, ranks), dtype=np.int64), device=dev)
        g.parts[r].copy_(full[rows])
    own = torch.as_tensor(np.asarray(host.global_rows(H, 0, ranks), dtype=np.int64), device=dev)
    for _ in range(2):  # both of the gather's frame buffers
        g.frame = g.frames[g.calls % len(g.frames)]
        g.calls += 1
        g.frame.zero_()
        assert torch.equal(g._place(full[own].contiguous()), full)
