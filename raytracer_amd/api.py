"""The shipped host interface: scene construction (mirror.Api bound to librt_host.so) plus `render`
and `color_at`, which run on the GPU through the C ABI.  There is no CPU fallback: without the built
libraries or without a GPU these calls raise."""
import ctypes as C
import os

import numpy as np

from . import abi
from .mirror import Api, c_d, c_dp, c_i64

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.environ.get("RT_HOST_LIB") or os.path.join(HERE, "host", "librt_host.so")  # RT_HOST_LIB: diagnostic builds


class HostApi(Api):
    def __init__(self):
        if not os.path.exists(HOST_LIB):
            raise RuntimeError("librt_host.so is not built: run __graft_entry__.build()")
        self.abi = abi.load()  # librt_amd.so first (RTLD_GLOBAL), the host library links against it
        lib = C.CDLL(HOST_LIB)
        super().__init__(lib, "rth_")
        P = C.POINTER
        sig = {
            "last_error": (C.c_char_p, []),
            "camera_block": (C.c_int, [c_i64, P(abi.rt_camera)]),
            "flatten": (C.c_int, [c_i64, P(C.c_int32)]),
            "flat_nodes": (C.c_int, [c_i64, P(abi.rt_node)]),
            "flat_aabbs": (C.c_int, [c_i64, P(abi.rt_aabb)]),
            "flat_xforms": (C.c_int, [c_i64, P(abi.rt_xform)]),
            "scene_desc": (C.c_int, [c_i64, P(abi.rt_scene_desc)]),
            "world_scene": (C.c_int, [c_i64, C.c_int, P(C.c_void_p)]),
            "render": (C.c_int, [c_i64, c_i64, C.c_int, P(abi.rt_params), c_dp, P(abi.rt_stats)]),
            "render_device": (C.c_int, [c_i64, c_i64, C.c_int, P(abi.rt_params), C.c_void_p, C.c_void_p]),
            "render_u8": (C.c_int, [c_i64, c_i64, C.c_int, P(abi.rt_params), P(C.c_uint8), P(abi.rt_stats)]),
            "render_device_u8": (C.c_int, [c_i64, c_i64, C.c_int, P(abi.rt_params), C.c_void_p, C.c_void_p]),
            "color_at": (C.c_int, [c_i64, C.c_int, c_i64, c_dp, c_dp, C.c_int, c_dp, P(abi.rt_stats)]),
            "clamp_u8": (C.c_int, [c_d]),
            "to_ppm": (c_i64, [c_dp, c_i64, c_i64]),
            "to_ppm_u8": (c_i64, [P(C.c_uint8), c_i64, c_i64]),
            "ppm_copy": (None, [C.c_char_p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, "rth_" + name)
            fn.restype = res
            fn.argtypes = args

    def _check(self, rc):
        if rc != 0:
            raise abi.RtError(rc, (self._f("last_error")() or b"").decode())

    # ---- flatten inspection -------------------------------------------------------------------
    def flatten(self, world):
        """Flattens the world; returns the table sizes."""
        counts = (C.c_int32 * 9)()
        self._check(self._f("flatten")(world.h, counts))
        keys = ["nodes", "xforms", "aabbs", "ranges", "tris", "materials", "patterns", "pattern_xforms",
                "lights"]
        return dict(zip(keys, [int(c) for c in counts]))

    def flat_nodes(self, world):
        n = self.flatten(world)["nodes"]
        out = (abi.rt_node * max(n, 1))()
        self._check(self._f("flat_nodes")(world.h, out))
        return [out[i] for i in range(n)]

    def flat_aabbs(self, world):
        n = self.flatten(world)["aabbs"]
        out = (abi.rt_aabb * max(n, 1))()
        self._check(self._f("flat_aabbs")(world.h, out))
        return [(tuple(out[i].min), tuple(out[i].max)) for i in range(n)]

    def scene_desc(self, world):
        d = abi.rt_scene_desc()
        self._check(self._f("scene_desc")(world.h, C.byref(d)))
        return d

    def camera_block(self, camera):
        cam = abi.rt_camera()
        self._check(self._f("camera_block")(camera.h, C.byref(cam)))
        return cam

    def scene_handle(self, world, device=0):
        h = C.c_void_p()
        self._check(self._f("world_scene")(world.h, device, C.byref(h)))
        return h

    # ---- the render path ------------------------------------------------------------------------
    def render(self, world, camera, max_depth=5, device=0, rank=0, world_size=1, stats=False):
        """render(world, camera) (src/lib.rs:77-133) on the GPU -> float64 array (rows, W, 3).
        With world_size > 1 only this rank's cyclic row tiles are rendered, in compact order."""
        prm = abi.rt_params(max_depth, rank, world_size, 0)
        W, H = camera.width(), camera.height()
        rows = self.abi.rt_local_rows(H, C.byref(prm))
        if rows < 0:
            raise abi.RtError(abi.RT_ERR_INVALID, self.abi.rt_last_error().decode())
        img = np.empty((rows, W, 3), dtype=np.float64)
        st = abi.rt_stats()
        self._check(self._f("render")(world.h, camera.h, device, C.byref(prm), img.ctypes.data_as(c_dp),
                                      C.byref(st)))
        return (img, st) if stats else img

    def render_u8(self, world, camera, max_depth=5, device=0, rank=0, world_size=1, stats=False):
        """render followed by Canvas::clamp (src/window/canvas.rs:50-60) on the GPU -> uint8 array
        (rows, W, 3): the values to_ppm prints."""
        prm = abi.rt_params(max_depth, rank, world_size, 0)
        W, H = camera.width(), camera.height()
        rows = self.abi.rt_local_rows(H, C.byref(prm))
        if rows < 0:
            raise abi.RtError(abi.RT_ERR_INVALID, self.abi.rt_last_error().decode())
        img = np.empty((rows, W, 3), dtype=np.uint8)
        st = abi.rt_stats()
        self._check(self._f("render_u8")(world.h, camera.h, device, C.byref(prm),
                                         img.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(st)))
        return (img, st) if stats else img

    def render_ppm(self, world, camera, max_depth=5, device=0, stats=False):
        """render + Canvas::to_ppm (src/lib.rs:77-133, src/window/canvas.rs:62-126), both on the GPU
        -> the bytes main.rs writes to the output file."""
        prm = abi.rt_params(max_depth, 0, 1, 0)
        cam = self.camera_block(camera)
        scene = self.scene_handle(world, device)
        cap = 32 + cam.height * (cam.width * 12 + 1)
        buf = C.create_string_buffer(cap)
        n = C.c_uint64()
        st = abi.rt_stats()
        rc = self.abi.rt_render_ppm(scene, C.byref(cam), C.byref(prm), buf, cap, C.byref(n), C.byref(st))
        if rc != 0:
            raise abi.RtError(rc, (self.abi.rt_last_error() or b"").decode())
        text = buf.raw[:n.value]
        return (text, st) if stats else text

    def ppm_encode(self, img8, device=0, capacity=None):
        """Canvas::to_ppm (src/window/canvas.rs:62-126) on the GPU for an (H, W, 3) uint8 image."""
        img8 = np.ascontiguousarray(img8, dtype=np.uint8)
        h, w = img8.shape[:2]
        cap = 32 + h * (w * 12 + 1) if capacity is None else capacity
        buf = C.create_string_buffer(max(cap, 1))
        n = C.c_uint64()
        rc = self.abi.rt_ppm_encode(device, img8.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, buf, cap, C.byref(n))
        if rc != 0:
            err = abi.RtError(rc, (self.abi.rt_last_error() or b"").decode())
            err.needed = n.value
            raise err
        return buf.raw[:n.value]

    def render_device(self, world, camera, d_ptr, stream=None, max_depth=5, device=0, rank=0, world_size=1,
                      u8=False):
        """Asynchronous render into device memory at address d_ptr (e.g. a torch tensor's data_ptr):
        3 doubles per pixel, or with u8=True the 3 clamped bytes per pixel."""
        prm = abi.rt_params(max_depth, rank, world_size, 0)
        self._check(self._f("render_device_u8" if u8 else "render_device")(
            world.h, camera.h, device, C.byref(prm), C.c_void_p(d_ptr), C.c_void_p(stream or 0)))

    def color_at(self, world, origins, directions, remaining=5, device=0, stats=False):
        """World::color_at (src/scene/world.rs:198-212) for a batch of rays on the GPU."""
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        assert o.shape == d.shape
        out = np.empty_like(o)
        st = abi.rt_stats()
        self._check(self._f("color_at")(world.h, device, o.shape[0], o.ctypes.data_as(c_dp),
                                        d.ctypes.data_as(c_dp), remaining, out.ctypes.data_as(c_dp),
                                        C.byref(st)))
        return (out, st) if stats else out

    # ---- several GPUs from one process -----------------------------------------------------------
    def multi(self, world, n_ranks, devices=None):
        """One handle that owns a copy of `world` on every GPU it uses (rt_multi_create).  devices:
        HIP ordinals per rank (may repeat), default 0 .. n_ranks-1."""
        return Multi(self, world, n_ranks, devices)

    def pinned_empty(self, shape, dtype):
        """A numpy array in page-locked host memory (rt_host_alloc): the GPU writes it directly."""
        return PinnedArray(self, shape, dtype)

    def local_rows(self, height, rank, world_size):
        prm = abi.rt_params(0, rank, world_size, 0)
        return int(self.abi.rt_local_rows(height, C.byref(prm)))

    def global_rows(self, height, rank, world_size):
        """Frame row of every compact row of `rank`."""
        prm = abi.rt_params(0, rank, world_size, 0)
        n = self.local_rows(height, rank, world_size)
        return np.array([self.abi.rt_global_row(i, C.byref(prm)) for i in range(n)], dtype=np.int64)

    # ---- canvas -----------------------------------------------------------------------------------
    def clamp(self, v):
        return self._f("clamp_u8")(float(v))

    def to_ppm(self, img):
        """Canvas::to_ppm (src/window/canvas.rs:62-94) for an (H, W, 3) float64 image, or for a
        uint8 image that already holds the clamped values."""
        h, w = img.shape[:2]
        if img.dtype == np.uint8:
            img = np.ascontiguousarray(img)
            n = self._f("to_ppm_u8")(img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h)
        else:
            img = np.ascontiguousarray(img, dtype=np.float64)
            n = self._f("to_ppm")(img.ctypes.data_as(c_dp), w, h)
        buf = C.create_string_buffer(n)
        self._f("ppm_copy")(buf)
        return buf.raw[:n]


class _PinnedBlock:
    """One rt_host_alloc block, returned to the library when the last reference goes."""

    def __init__(self, host, n):
        self.free = host.abi.rt_host_free
        self.ptr = host.abi.rt_host_alloc(max(n, 1))
        if not self.ptr:
            raise abi.RtError(abi.RT_ERR_HIP, (host.abi.rt_last_error() or b"").decode())

    def __del__(self):
        try:
            if self.ptr:
                self.free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class PinnedArray:
    """`.array` is a numpy view of one rt_host_alloc block.  The block belongs to the buffer object every
    view of it is based on (the array itself, slices, what Multi.render(out=...) returns), so it is only
    freed once close() has been called (or this object is gone) AND no view is left: reading a view
    after close() stays valid."""

    def __init__(self, host, shape, dtype):
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        block = _PinnedBlock(host, n)
        self.ptr = block.ptr
        buf = (C.c_char * max(n, 1)).from_address(block.ptr)
        buf._block = block  # numpy keeps `buf` as the base of every view
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        self.array = None
        self.ptr = None


class Multi:
    """render (src/lib.rs:77-133) over several GPUs of this process: rt_multi_create and the
    rt_render_multi* entry points."""

    def __init__(self, host, world, n_ranks, devices=None):
        self.host = host
        self.world = world  # keeps the tables alive while the handle is created
        desc = host.scene_desc(world)
        dev = None
        if devices is not None:
            assert len(devices) == n_ranks
            dev = (C.c_int32 * n_ranks)(*[int(d) for d in devices])
        self.h = C.c_void_p()
        rc = host.abi.rt_multi_create(C.byref(desc), n_ranks, dev, C.byref(self.h))
        if rc != 0:
            raise abi.RtError(rc, (host.abi.rt_last_error() or b"").decode())
        self.n_ranks = n_ranks

    def _check(self, rc):
        if rc != 0:
            raise abi.RtError(rc, (self.host.abi.rt_last_error() or b"").decode())

    def render(self, camera, max_depth=5, out=None, u8=False, stats=False):
        """The whole frame in host memory -> (H, W, 3) float64, or uint8 with u8=True.  `out`: an
        array to fill (for example PinnedArray.array)."""
        cam = self.host.camera_block(camera)
        dtype = np.uint8 if u8 else np.float64
        if out is None:
            out = np.empty((cam.height, cam.width, 3), dtype=dtype)
        assert out.shape == (cam.height, cam.width, 3) and out.dtype == dtype and out.flags.c_contiguous
        st = abi.rt_multi_stats()
        if u8:
            self._check(self.host.abi.rt_render_multi_u8(self.h, C.byref(cam), max_depth,
                                                         out.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(st)))
        else:
            self._check(self.host.abi.rt_render_multi(self.h, C.byref(cam), max_depth, out.ctypes.data_as(c_dp),
                                                      C.byref(st)))
        return (out, st) if stats else out

    def render_device(self, camera, d_ptr, max_depth=5, u8=False):
        """The whole frame in device memory at d_ptr on rank 0's GPU; returns the stats."""
        cam = self.host.camera_block(camera)
        st = abi.rt_multi_stats()
        self._check(self.host.abi.rt_render_multi_device(self.h, C.byref(cam), max_depth, C.c_void_p(d_ptr),
                                                         1 if u8 else 0, C.byref(st)))
        return st

    def close(self):
        if self.h:
            self.host.abi.rt_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_API = None


def load():
    global _API
    if _API is None:
        _API = HostApi()
    return _API
