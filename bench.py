"""Benchmark of the render path: Mrays/s (primary + secondary) and frame time.

  python bench.py --gpus 1 --steps K --warmup W            single GPU, default scene at 3840x2160
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
                                                          N GPUs, default scene at 7680x4320, the
                                                          frame split into cyclic 8-row tiles and
                                                          gathered to rank 0 over RCCL

A step renders one frame from tables already resident in HBM into HBM (at N > 1 it also gathers the
ranks' tiles into rank 0's full frame).  Rank 0 prints ONE JSON line.

`roofline` prices the render kernel against the FP64 vector peak: achieved = algorithmic f64
operations of the frame (counted by the CPU oracle with fixed per-event costs, profiles/work_*.json)
/ the kernel's mean device time from HIP events on its launch stream.
`cpu_baseline` times the CPU restatement of the reference in its cost-faithful mode (inverse
recomputed per use, secondary rays re-traced per light, 12 worker threads like src/lib.rs:23) on a
bounded sample of rows of the same frame, on this machine's host cores.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 128 FLOP/clk x 2.4 GHz, FMA counted as 2 (datasheet)
HBM_PEAK_GBS = 8000.0


def load_work(name, W, H):
    path = os.path.join(ROOT, "profiles", "work_%s_%dx%d.json" % (name, W, H))
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return None


def cpu_baseline(name, W, H, budget_s=15.0):
    """Cost-faithful CPU restatement on evenly spaced 2-row bands of the same frame."""
    from oracle import binding
    from raytracer_amd import scenes
    orc = binding.load()
    mw, mc = scenes.WORKLOADS[name]
    world, cam = mw(orc), mc(orc, W, H)
    threads = 12  # NUM_OF_THREADS, src/lib.rs:23
    band = 2
    n_bands = 64
    starts = [int((k + 0.5) * H / n_bands) // band * band for k in range(n_bands)]
    rays = 0
    seconds = 0.0
    seconds_fast = 0.0
    rows = 0
    for r0 in starts:
        r1 = min(r0 + band, H)
        orc.set_faithful(False)
        t = time.perf_counter()
        _, cnt = orc.render(world, cam, threads=threads, rows=(r0, r1), counters=True)
        seconds_fast += time.perf_counter() - t
        orc.set_faithful(True)
        t = time.perf_counter()
        orc.render(world, cam, threads=threads, rows=(r0, r1))
        seconds += time.perf_counter() - t
        rays += cnt["rays_primary"] + cnt["rays_shadow"] + cnt["rays_reflect"] + cnt["rays_refract"]
        rows += r1 - r0
        if seconds > budget_s:
            break
    orc.set_faithful(False)
    orc.reset()
    return {
        "value": rays / seconds / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d rows (%d evenly spaced bands of %d) of the %dx%d %s frame, %.1f s; rays counted "
                  "once per hit as on the GPU, timed in the cost-faithful mode" % (rows, len(starts), band, W, H, name, seconds),
        "host_cpus": os.cpu_count(),
        # the same rows with every inverse computed once per shape instead of once per use (what the
        # GPU path's host side does too): how much of the reference's CPU cost is that recomputation
        "value_with_cached_inverses": rays / seconds_fast / 1e6 if seconds_fast > 0 else None,
    }


def load_traffic(workload, W, H):
    """HBM bytes per launch measured with rocprofv3 PMC passes on this workload (profiles/traffic.json), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    entry = json.load(open(path)).get("%s_%dx%d" % (workload, W, H))
    return entry["bytes_per_launch"] if entry else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # a frame is under a millisecond; the first few dozen run before the clocks settle
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="default")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-depth", type=int, default=5, help="diagnostic: `remaining` of primary rays (the metric uses 5)")
    ap.add_argument("--gather", choices=["u8", "f64"], default="u8",
                    help="N > 1: what rank 0 collects per frame: the clamped bytes the PPM prints (default) or the f64 colours")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: gather each frame before rendering the next")
    ap.add_argument("--gather-mode", choices=["p2p", "collective"], default="p2p",
                    help="N > 1: grouped send/recv of exactly each rank's rows (default) or one padded dist.gather")
    ap.add_argument("--in-flight", type=int, choices=[1, 2], default=0,
                    help="frames rendering at once: 2 = consecutive frames on two streams (each with its own scene "
                         "handle and buffer), so that the next frame's wavefronts fill the GPU while the long wavefronts "
                         "of this one finish.  Default: 1 on one GPU, 2 on several (a rank's share of the frame is then "
                         "not much longer than its longest wavefront)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, api, scenes

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world_size != args.gpus:
        raise SystemExit("--gpus %d needs a torch.distributed launch with WORLD_SIZE=%d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")  # "gloo" only to rehearse N > 1 on one GPU
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()  # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    stat_dev = dev if backend == "nccl" else torch.device("cpu")

    # N = 1: the single-GPU configuration (3840x2160); N > 1: the 8K frame split across ranks.
    W = args.width or (3840 if world_size == 1 else 7680)
    H = args.height or (2160 if world_size == 1 else 4320)

    in_flight = args.in_flight or (2 if world_size > 1 and not args.no_overlap else 1)
    host = api.load()
    make_world, make_camera = scenes.WORKLOADS[args.workload]
    # one scene handle per frame in flight: launches on one handle must be stream-ordered
    worlds = [make_world(host) for _ in range(in_flight)]
    world, cam = worlds[0], make_camera(host, W, H)
    scenes_h = [host.scene_handle(w, device=local_rank) for w in worlds]
    scene = scenes_h[0]
    rows = host.local_rows(H, rank, world_size)
    u8 = world_size > 1 and args.gather == "u8"
    dtype = torch.uint8 if u8 else torch.float64
    buffers = [torch.empty((rows, W, 3), dtype=dtype, device=dev) for _ in range(2 if world_size > 1 or in_flight > 1 else 1)]

    from raytracer_amd.distributed import FrameGather, OverlappedGather
    gather = FrameGather(host, W, H, rank, world_size, dev, dtype=dtype, mode=args.gather_mode) if world_size > 1 else None
    render_streams = [torch.cuda.Stream(device=dev) for _ in range(2)] if in_flight > 1 else None
    pipe = (OverlappedGather(gather, buffers, render_streams)
            if (gather is not None and not args.no_overlap) or in_flight > 1 else None)

    def render(buf, slot=0):
        stream = torch.cuda.current_stream().cuda_stream
        host.render_device(worlds[slot % in_flight], cam, buf.data_ptr(), stream=stream, device=local_rank, rank=rank,
                           world_size=world_size, max_depth=args.max_depth, u8=u8)

    def step():
        if pipe is not None:
            pipe.step(render)  # RCCL send/recv of frame k's tile buffers overlaps the render of frame k + 1
        else:
            render(buffers[0])
            if gather is not None:
                gather(buffers[0])  # RCCL send/recv of the compact tile buffers to rank 0, rows into frame order

    def sync():
        torch.cuda.synchronize()  # all streams of this device, the gather's side stream included
        if world_size > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0

    # kernel device times of the timed launches, from HIP events on the launch stream
    cap = min(args.steps, 256)
    ms = (ctypes.c_float * cap)()
    got = host.abi.rt_scene_launch_times(scene, ms, min(cap, max(args.steps // in_flight, 1)))
    kernel_ms = sum(ms[i] for i in range(got)) / max(got, 1)
    st = abi.rt_stats()
    host.abi.rt_scene_ray_counts(scene, ctypes.byref(st))
    rays_local = st.total_rays()

    t = torch.tensor([elapsed, kernel_ms, float(rays_local)], dtype=torch.float64, device=stat_dev)
    if world_size > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, kernel_ms, rays = float(tmax[0]), float(tmax[1]), float(tsum[2])
    else:
        rays = float(rays_local)

    if rank == 0:
        work = load_work(args.workload, W, H) if args.max_depth == 5 else None
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "Mrays/s (primary+secondary), %s scene %dx%d" % (args.workload, W, H),
            "value": rays * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if world_size > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s scene%s, %dx%d, depth %d" % (
                    args.workload, " (src/lib.rs build_world)" if args.workload == "default" else " (raytracer_amd/scenes.py)",
                    W, H, args.max_depth),
                "rays_per_frame": int(rays),
                "partition": "cyclic 8-row tiles over %d rank(s)%s" % (
                    world_size,
                    ", RCCL gather of %s rows to rank 0%s" % (
                        "clamped u8" if u8 else "f64", "" if args.no_overlap else ", overlapped with the next frame's render")
                    if world_size > 1 else "") + (", %d frames in flight on separate streams" % in_flight if in_flight > 1 else ""),
            },
            "kernel_ms": kernel_ms,  # mean launch duration; with 2 frames in flight launches overlap and each takes longer
            "frames_in_flight": in_flight,
        }
        if work is not None:
            flops = float(work["flops"])
            # with two frames in flight a launch shares the GPU with its neighbour for most of its life:
            # the rate is then what the GPU sustains over the timed region, not flops / one launch's duration
            per_frame_ms = ms_per_step if in_flight > 1 else kernel_ms
            achieved = flops / world_size / (per_frame_ms * 1e-3) / 1e12 if per_frame_ms > 0 else 0.0
            out["roofline"] = {
                "bound": "valu-fp64",
                "achieved": achieved,
                "peak": FP64_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
                "traffic": load_traffic(args.workload, W, H) if world_size == 1 else None,
                "peak_without_fma": FP64_VECTOR_PEAK_TFLOPS / 2,
                "frac_without_fma": achieved / (FP64_VECTOR_PEAK_TFLOPS / 2),
                "algorithmic_flops_per_launch": flops / world_size,
                "oracle_rays_per_frame": work["rays"],
            }
            # the secondary roof: the only streamed bytes are the frame itself (scene tables stay in cache)
            frame_bytes = rows * W * 3 * (1 if u8 else 8)
            traffic = out["roofline"]["traffic"]
            out["roofline"]["hbm"] = {
                "peak_gbs": HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": frame_bytes,
                "algorithmic_gbs": frame_bytes / (per_frame_ms * 1e-3) / 1e9 if per_frame_ms > 0 else 0.0,
                "measured_gbs": traffic / (per_frame_ms * 1e-3) / 1e9 if traffic and per_frame_ms > 0 else None,
            }
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, W, H)
        print(json.dumps(out))
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
