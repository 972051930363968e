"""Kernel time of ONE rank's share of a frame split over `world_size` ranks (cyclic 8-row tiles), on
this GPU: what each GPU of an N-GPU run has to render.  usage: bench_rank.py W H world_size [rank] [steps]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from raytracer_amd import api, scenes  # noqa: E402

W, H, ws = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rank = int(sys.argv[4]) if len(sys.argv) > 4 else 0
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 60
host = api.load()
world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
rows = host.local_rows(H, rank, ws)
buf = torch.empty((rows, W, 3), dtype=torch.uint8, device="cuda:0")
scene = host.scene_handle(world, 0)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    host.render_device(world, cam, buf.data_ptr(), stream=stream, rank=rank, world_size=ws, u8=True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    host.render_device(world, cam, buf.data_ptr(), stream=stream, rank=rank, world_size=ws, u8=True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / steps * 1e3
ms = (ctypes.c_float * steps)()
n = host.abi.rt_scene_launch_times(scene, ms, steps)
print("%dx%d rank %d of %d: %d rows, kernel %.3f ms, wall %.3f ms per frame" % (W, H, rank, ws, rows, sum(ms[:n]) / n, wall))
