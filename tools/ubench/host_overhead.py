"""Host time of one step of bench.py's loop against the GPU time of one rank's share of the 7680x4320
frame at N = 8: the loop only keeps up while enqueueing a step takes less than rendering it.
usage (GPU box): python tools/ubench/host_overhead.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from raytracer_amd import api, scenes  # noqa: E402
from raytracer_amd.distributed import FrameGather, OverlappedGather  # noqa: E402

host = api.load()
dev = torch.device("cuda", 0)
W, H, N = 7680, 4320, 8
world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
rows = host.local_rows(H, 0, N)
bufs = [torch.empty((rows, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]


def render(buf, slot=0):
    host.render_device(world, cam, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, device=0,
                       rank=0, world_size=N, u8=True)


def timed(step, steps=300):
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3


enq, tot = timed(lambda: render(bufs[0]))
print("render_device alone:            enqueue %.3f ms per step, %.3f ms per step with the GPU" % (enq, tot))
# rank 0's placement of the peers' rows on the side stream, without the exchange itself
g = FrameGather(host, W, H, 0, N, dev, dtype=torch.uint8)
pipe = OverlappedGather(g, bufs)
g_call = FrameGather.__call__


def place_only(self, local):
    self.frame = self.frames[self.calls % len(self.frames)]
    self.calls += 1
    return self._place(local)


FrameGather.__call__ = place_only
enq, tot = timed(lambda: pipe.step(render))
print("render + row placement (side stream): enqueue %.3f ms per step, %.3f ms per step with the GPU" % (enq, tot))
