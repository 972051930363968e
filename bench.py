"""Benchmark of the render path: Mrays/s (primary + secondary) and frame time.

  python bench.py [--gpus N] [--steps K] [--warmup W]

One experiment at every N: the reference's default scene (src/lib.rs:135-323) at 7680x4320, depth 5,
split into cyclic 8-row tiles over N ranks (one process per GPU), every rank storing the clamped
bytes the PPM prints; at N > 1 rank 0 collects the ranks' tiles over RCCL while the next frame
renders.  With N > 1 and no torch.distributed launcher around it, the script starts its own N worker
processes (fresh interpreters, started before anything touches a GPU).  Rank 0 prints ONE JSON line.

A step renders one frame from tables already resident in HBM into HBM.  Besides the timed region
the line carries:
  frame_latency_ms, kernel_ms, gather_ms, d2h_ms   one frame at a time, nothing overlapped
  roofline      the render kernel against the FP64 vector peak: algorithmic f64 operations of the
                frame (counted by the CPU oracle, profiles/work_*.json) / the kernel's mean launch
                duration (HIP events on its stream); HBM traffic and executed operations from the
                rocprofv3 PMC passes committed under profiles/ (profiles/pmc.json)
and at N = 1 only:
  single_gpu_4k the same scene at 3840x2160 with the f64 frame store (the single-GPU configuration)
  e2e           wall time per frame of the host-memory entry points rt_render / rt_render_u8 /
                rt_render_ppm, and the measured page-locked PCIe rate
  scene_build   host-side time to parse, divide, flatten, lay out and upload two large scenes
  cpu_baseline  the CPU restatement of the reference in its cost-faithful mode (inverse recomputed
                per use, secondary rays re-traced per light, 12 worker threads like src/lib.rs:23) on
                a bounded sample of rows of the same frame, on this machine's host cores.
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


def load_json(*parts):
    path = os.path.join(ROOT, *parts)
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return None


def load_work(name, W, H):
    return load_json("profiles", "work_%s_%dx%d.json" % (name, W, H))


def load_pmc(workload, W, H):
    """What rocprofv3's PMC passes measured per launch on this configuration (profiles/pmc.json)."""
    table = load_json("profiles", "pmc.json") or {}
    return table.get("%s_%dx%d" % (workload, W, H))


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
        "sample": "%d rows (evenly spaced bands of %d) of the %dx%d %s frame, %.1f s; rays counted "
                  "once per hit as on the GPU, timed in the cost-faithful mode" % (rows, band, W, H, name, seconds),
        "host_cpus": os.cpu_count(),
        "cargo_on_path": bool(__import__("shutil").which("cargo")),  # the Rust reference itself needs it (and its crates)
        # the same rows with every inverse computed once per shape instead of once per use (what the
        # GPU path's host side does too): how much of the reference's CPU cost is that recomputation
        "value_with_cached_inverses": rays / seconds_fast / 1e6 if seconds_fast > 0 else None,
    }


N_SIMD = 1024                   # 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9


def table_bytes(host, scene):
    """Device bytes of the scene's tables (all of them, and the traversal records alone)."""
    from raytracer_amd import abi
    info = abi.rt_build_info()
    host.abi.rt_scene_build_info(scene, ctypes.byref(info))
    return int(info.table_bytes), int(info.walk_bytes)


def scaled_pmc(pmc, share):
    """Counters of a whole-frame launch, scaled to a rank that renders `share` of the frame's tiles (the
    tiles are dealt cyclically, so every rank gets the same mix)."""
    if pmc is None or share == 1.0:
        return pmc
    out = dict(pmc)
    for k in ("bytes_per_launch", "valu_busy_quads", "valu_insts", "valu_insts_in_2clock_classes", "executed_flops",
              "smem_insts", "scalar_cache_requests", "scalar_cache_misses"):
        if out.get(k) is not None:
            out[k] = out[k] * share
    out["source"] = "%s, scaled by %.4g (this rank's share of the frame)" % (pmc.get("source"), share)
    return out


FETCH_CLOCKS = 235  # dependent 128-byte scalar fetch, 20 wavefronts per CU (tools/ubench/smem_latency, profiles/r3/issue_calib.txt)


def load_walks(workload, W, H):
    """Walks and records visited of one frame, from the -DRT_PROFILE build (tools/phase_set.sh -> profiles/walks.json)."""
    table = load_json("profiles", "walks.json") or {}
    return table.get("%s_%dx%d" % (workload, W, H))


def roofline_block(work, pmc, flops_share, kernel_ms, frame_bytes, W, H, tables=None, rays=None, walks=None):
    """The render kernel against the roofs of the machine.

    flops_share: this launch's part of the frame's algorithmic operations.  The kernel has no dense
    contraction (no MFMA) and its tables stay in the caches, so neither textbook roof is near; what it
    can run into is the issue rate of the vector ALUs and, on scenes with long walks, the latency of
    the chain of dependent scalar record fetches.  `bound` names the one that is nearest, `frac` is
    the fraction of it (<= 1): live kernel time, counters of the rocprofv3 PMC passes of the same
    command (profiles/pmc.json; tools/pmc_to_json.py documents units, corrections and the calibration).
    """
    sec = kernel_ms * 1e-3
    fp64 = flops_share / sec / 1e12 if sec > 0 else 0.0
    r = {"kernel_ms": kernel_ms, "traffic": pmc["bytes_per_launch"] if pmc else None}
    cand = {}
    if pmc and pmc.get("valu_busy_quads") and sec > 0:
        busy = 4.0 * pmc["valu_busy_quads"]  # SIMD clocks spent issuing vector instructions, per launch
        vi = {
            "busy_simd_clocks_per_launch": busy,
            "frac": busy / (N_SIMD * PEAK_CLOCK_HZ * sec),
            "frac_is": "4 x SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs x 2.4 GHz x live kernel time); the counter charges "
                       "2-clock 32-bit instructions (v_mov, v_and, f32 add / multiply-add) a whole quad, so this is an upper "
                       "bound; frac_min charges 2 clocks for every instruction outside the classes known to take 4 or more "
                       "(F64, CVT, INT32, INT64, TRANS_F32): a lower bound (compares, selects, min / max are outside them too)",
            "frac_from_counters_alone": pmc.get("valu_issue_frac"),  # clocks from GRBM_GUI_ACTIVE of the same pass
            "valu_insts_per_launch": pmc.get("valu_insts"),
            "f64_share_of_valu_insts": pmc.get("f64_share_of_valu_insts"),
            "mean_active_lanes": pmc.get("mean_active_lanes"),
        }
        if pmc.get("valu_insts_in_2clock_classes") is not None:
            vi["frac_min"] = (busy - 2.0 * pmc["valu_insts_in_2clock_classes"]) / (N_SIMD * PEAK_CLOCK_HZ * sec)
        if rays:
            vi["valu_insts_per_ray"] = 64.0 * pmc["valu_insts"] / rays if pmc.get("valu_insts") else None  # lane instructions
        r["valu_issue"] = vi
        cand["valu-issue"] = (vi["frac"], busy / sec / 1e9, N_SIMD * PEAK_CLOCK_HZ / 1e9, "G SIMD-clocks/s")
    # ---- the walk's chain of dependent record fetches (scalar loads through the scalar cache) ---------------
    if pmc and pmc.get("scalar_cache_requests") and sec > 0:
        sf = {
            "smem_insts_per_launch": pmc.get("smem_insts"),
            "scalar_cache_requests_per_launch": pmc["scalar_cache_requests"],
            "scalar_cache_requests_per_s": pmc["scalar_cache_requests"] / sec,
            "scalar_cache_miss_rate": (pmc.get("scalar_cache_misses") or 0.0) / pmc["scalar_cache_requests"],
            "salu_issue_frac": 4.0 * pmc["salu_insts"] / (N_SIMD * PEAK_CLOCK_HZ * sec) if pmc.get("salu_insts") else None,
            "mean_resident_waves_per_simd": pmc.get("mean_resident_waves_per_simd"),
        }
        if walks:
            # from the -DRT_PROFILE build of the same frame (profiles/walks.json): every record costs one dependent
            # fetch, 235 clocks on a busy CU when it hits the scalar cache (tools/ubench/smem_latency)
            per_wave = walks["records_visited"] / walks["waves"]
            sf.update({
                "walks_per_launch": walks["walks"], "records_visited_per_launch": walks["records_visited"],
                "records_per_walk": walks["records_per_walk"],
                "fetch_chain_clocks_per_wave": per_wave * FETCH_CLOCKS,
                "mean_wave_clocks": walks["mean_wave_clocks"],
                "fetch_chain_share_of_wave_time": per_wave * FETCH_CLOCKS / walks["mean_wave_clocks"],
                "fetch_chain_share_is": "records visited per wavefront x %d clocks (a dependent 128-byte scalar fetch that hits, on a "
                                        "busy CU) / mean wavefront lifetime; a lower bound when fetches miss the scalar cache" % FETCH_CLOCKS,
                "time_share_of_walks": walks["share"]["nearest_hit_walks"] + walks["share"]["shadow_walks"],
            })
            cand["scalar-fetch-chain"] = (sf["fetch_chain_share_of_wave_time"], per_wave * FETCH_CLOCKS, walks["mean_wave_clocks"],
                                          "clocks per wavefront")
        r["scalar_fetch"] = sf
    # ---- FP64: operations of the reference's algorithm per second, and what the ALUs executed -----------
    f = {
        "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
        "algorithmic_flops_per_launch": flops_share,
        "reference_work_rate": fp64,
        "reference_work_rate_is": "f64 operations of the REFERENCE's algorithm for this frame (oracle count) per second.  The "
                                  "kernel skips work whose outcome is known (cull boxes, pruned subtrees, f32 box tests), so "
                                  "this is not ALU utilisation and may exceed the peak on meshes; `frac_executed` is utilisation",
        "reference_work_rate_over_peak": fp64 / FP64_VECTOR_PEAK_TFLOPS,
        "peak_without_fma": FP64_VECTOR_PEAK_TFLOPS / 2,
        "reference_work_rate_over_peak_without_fma": fp64 / (FP64_VECTOR_PEAK_TFLOPS / 2),
    }
    if pmc and pmc.get("executed_flops") and sec > 0:
        ex = pmc["executed_flops"] / sec / 1e12
        f["executed_flops_per_launch"] = pmc["executed_flops"]  # 64 lanes x (f64 add + mul + 2 fma + transcendental) wave instructions
        f["executed_tflops"] = ex
        f["frac_executed"] = ex / FP64_VECTOR_PEAK_TFLOPS
        cand["valu-fp64"] = (f["frac_executed"], ex, FP64_VECTOR_PEAK_TFLOPS, "TFLOP/s")
    r["fp64"] = f
    # ---- HBM: compulsory bytes (every table once, the frame once) against what moved ------------------------
    if tables is not None:
        compulsory = tables[0] + frame_bytes
        h = {
            "peak_gbs": HBM_PEAK_GBS,
            "compulsory_bytes_per_launch": compulsory,
            "compulsory_bytes_is": "every table of the scene read once (%d B, of which traversal records %d B) + the frame "
                                   "stored once (%d B)" % (tables[0], tables[1], frame_bytes),
            "compulsory_gbs": compulsory / sec / 1e9 if sec > 0 else 0.0,
            "compulsory_frac_of_peak": compulsory / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else 0.0,
        }
        if pmc and sec > 0:
            h["measured_bytes_per_launch"] = pmc["bytes_per_launch"]  # 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction)
            h["measured_gbs"] = pmc["bytes_per_launch"] / sec / 1e9
            h["measured_frac_of_peak"] = h["measured_gbs"] / HBM_PEAK_GBS
            h["measured_over_compulsory"] = pmc["bytes_per_launch"] / compulsory
            cand["hbm"] = (h["measured_frac_of_peak"], h["measured_gbs"], HBM_PEAK_GBS, "GB/s")
        r["hbm"] = h
    # ---- the roof that is nearest ---------------------------------------------------------------------------
    if cand:
        name = max(cand, key=lambda k: cand[k][0])
        r["bound"] = name
        r["frac"], r["achieved"], r["peak"], r["unit"] = cand[name]
        r["fractions"] = {k: v[0] for k, v in cand.items()}
    else:
        # no counters for this configuration (N > 1, or one that was never profiled): only the reference-work rate
        r["bound"] = "valu-fp64"
        r["achieved"], r["peak"], r["unit"] = fp64, FP64_VECTOR_PEAK_TFLOPS, "TFLOP/s"
        r["frac"] = fp64 / FP64_VECTOR_PEAK_TFLOPS
        r["frac_is"] = "no PMC counters for this configuration: reference-algorithm f64 operations per second over the peak"
    if pmc:
        r["pmc_source"] = pmc.get("source")
    return r


def launch_times(host, scene, n):
    cap = max(1, min(n, 256))
    ms = (ctypes.c_float * cap)()
    got = host.abi.rt_scene_launch_times(scene, ms, cap)
    return [ms[i] for i in range(got)]


def timed_renders(host, world, cam, steps, warmup, torch, dev, u8, local_rank, max_depth=5):
    """Frames one at a time on the current stream: (wall ms per frame, mean launch ms, rays per frame)."""
    from raytracer_amd import abi
    W, H = cam.width(), cam.height()
    buf = torch.empty((H, W, 3), dtype=torch.uint8 if u8 else torch.float64, device=dev)
    scene = host.scene_handle(world, device=local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    def go():
        host.render_device(world, cam, buf.data_ptr(), stream=stream, device=local_rank, max_depth=max_depth, u8=u8)
    for _ in range(warmup):
        go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        go()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    ms = launch_times(host, scene, steps)
    st = abi.rt_stats()
    host.abi.rt_scene_ray_counts(scene, ctypes.byref(st))
    return wall, sum(ms) / max(len(ms), 1), st.total_rays(), buf


def single_gpu_4k(host, torch, dev, local_rank):
    """The single-GPU configuration: default scene, 3840x2160, f64 frame store, one frame at a time."""
    from raytracer_amd import scenes
    W, H = 3840, 2160
    world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
    wall, kernel_ms, rays, _ = timed_renders(host, world, cam, 100, 20, torch, dev, False, local_rank)
    work = load_work("default", W, H)
    out = {"workload": "default scene, 3840x2160, depth 5, f64 store, 1 frame in flight", "steps": 100,
           "ms_per_step": wall, "kernel_ms": kernel_ms, "rays_per_frame": rays, "value": rays / wall / 1e3, "unit": "Mrays/s"}
    if work:
        tables = table_bytes(host, host.scene_handle(world, device=local_rank))
        out["roofline"] = roofline_block(work, load_pmc("default", W, H), float(work["flops"]), kernel_ms, W * H * 24, W, H,
                                         tables=tables, rays=rays, walks=load_walks("default", W, H))
    return out


def single_gpu_8k_f64(host, torch, dev, local_rank):
    """The headline frame with the f64 store (what render() returns, 796 MB) instead of the clamped bytes."""
    from raytracer_amd import scenes
    W, H = 7680, 4320
    world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
    wall, kernel_ms, rays, buf = timed_renders(host, world, cam, 30, 5, torch, dev, False, local_rank)
    del buf
    torch.cuda.empty_cache()
    return {"workload": "default scene, 7680x4320, depth 5, f64 store, 1 frame in flight", "steps": 30,
            "ms_per_step": wall, "kernel_ms": kernel_ms, "rays_per_frame": rays, "value": rays / wall / 1e3, "unit": "Mrays/s"}


def e2e_block(host, torch, dev, local_rank, calls=12):
    """Wall time per frame of the entry points that hand the frame to HOST memory, at 3840x2160."""
    import numpy as np
    from raytracer_amd import abi, scenes
    W, H = 3840, 2160
    world, cam = scenes.build_world(host), scenes.default_camera(host, W, H)
    scene = host.scene_handle(world, device=local_rank)
    block = host.camera_block(cam)
    prm = abi.rt_params(5, 0, 1, 0)
    P = ctypes.POINTER

    def mean_ms(fn):
        fn()
        fn()  # the first calls allocate staging and fault the destination's pages in
        t0 = time.perf_counter()
        for _ in range(calls):
            fn()
        return (time.perf_counter() - t0) / calls * 1e3

    out = {"workload": "default scene, 3840x2160, depth 5", "calls": calls}
    pageable = np.empty((H, W, 3), dtype=np.float64)
    pinned = host.pinned_empty((H, W, 3), np.float64)
    pageable8 = np.empty((H, W, 3), dtype=np.uint8)
    pinned8 = host.pinned_empty((H, W, 3), np.uint8)
    for key, arr, fn, typ in (("rt_render_ms", pageable, host.abi.rt_render, ctypes.c_double),
                              ("rt_render_pinned_ms", pinned.array, host.abi.rt_render, ctypes.c_double),
                              ("rt_render_u8_ms", pageable8, host.abi.rt_render_u8, ctypes.c_uint8),
                              ("rt_render_u8_pinned_ms", pinned8.array, host.abi.rt_render_u8, ctypes.c_uint8)):
        ptr = arr.ctypes.data_as(P(typ))

        def call(fn=fn, ptr=ptr):
            rc = fn(scene, ctypes.byref(block), ctypes.byref(prm), ptr, None)
            assert rc == 0, host.abi.rt_last_error()
        out[key] = mean_ms(call)
    cap = 32 + H * (W * 12 + 1)
    text = ctypes.create_string_buffer(cap)
    n = ctypes.c_uint64()

    def ppm():
        rc = host.abi.rt_render_ppm(scene, ctypes.byref(block), ctypes.byref(prm), text, cap, ctypes.byref(n), None)
        assert rc == 0, host.abi.rt_last_error()
    out["rt_render_ppm_ms"] = mean_ms(ppm)
    out["ppm_bytes"] = int(n.value)
    # the PCIe rate a page-locked destination gets: the same 199 MB with the copy engine alone
    frame = torch.empty((H, W, 3), dtype=torch.float64, device=dev)
    host_pin = torch.empty((H, W, 3), dtype=torch.float64, pin_memory=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host_pin.copy_(frame, non_blocking=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        host_pin.copy_(frame, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    copy_ms = e0.elapsed_time(e1) / 5
    out["frame_bytes"] = W * H * 24
    out["pinned_copy_ms"] = copy_ms
    out["pinned_pcie_gbs"] = W * H * 24 / (copy_ms * 1e-3) / 1e9
    _, kernel_ms, _, _ = timed_renders(host, world, cam, 20, 5, torch, dev, False, local_rank)
    out["kernel_ms"] = kernel_ms
    out["kernel_plus_pinned_copy_ms"] = kernel_ms + copy_ms
    out["rt_render_over_kernel_plus_copy"] = out["rt_render_ms"] / (kernel_ms + copy_ms)
    out["rt_render_pinned_over_kernel_plus_copy"] = out["rt_render_pinned_ms"] / (kernel_ms + copy_ms)
    pinned.close()
    pinned8.close()
    del pageable, pageable8, frame, host_pin
    # the headline frame's way home: 7680x4320 clamped bytes to page-locked memory, per call
    W, H = 7680, 4320
    cam8 = scenes.default_camera(host, W, H)
    block8 = host.camera_block(cam8)
    big = host.pinned_empty((H, W, 3), np.uint8)
    ptr8 = big.array.ctypes.data_as(P(ctypes.c_uint8))

    def call8():
        rc = host.abi.rt_render_u8(scene, ctypes.byref(block8), ctypes.byref(prm), ptr8, None)
        assert rc == 0, host.abi.rt_last_error()
    out["rt_render_u8_pinned_7680x4320_ms"] = mean_ms(call8)
    _, k8, _, _ = timed_renders(host, world, cam8, 10, 3, torch, dev, True, local_rank)
    out["kernel_7680x4320_u8_ms"] = k8
    out["pinned_copy_7680x4320_u8_ms"] = W * H * 3 / (out["pinned_pcie_gbs"] * 1e9) * 1e3
    big.close()
    out["how"] = ("frames of 8 MB or more are rendered in five bands of tile rows (1/16, 3/16, 8/16, 3/16, 1/16) on two "
                  "streams and every band's rows start across PCIe when the band is done (RT_BANDS=0 turns that off)")
    return out


def scene_build_block(host, local_rank):
    """Host-side cost of getting a large scene onto the GPU: OBJ parse, Group::divide, flatten, the
    library's record layout, upload."""
    import math
    from raytracer_amd import abi, scenes
    out = {}

    def finish(world, entry):
        t = time.perf_counter()
        counts = host.flatten(world)
        entry["flatten_ms"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        scene = host.scene_handle(world, device=local_rank)
        entry["scene_create_ms"] = (time.perf_counter() - t) * 1e3
        info = abi.rt_build_info()
        host.abi.rt_scene_build_info(scene, ctypes.byref(info))
        entry.update({"validate_ms": info.validate_ms, "records_ms": info.records_ms, "upload_ms": info.upload_ms,
                      "nodes": counts["nodes"], "triangles": counts["tris"], "records": info.n_records,
                      "cull_records": info.n_cull_records})
        return entry

    M, Phong, Solid, Color = host.Matrix, host.Phong, host.Solid, host.Color
    mat = Phong(Solid(Color(0.8, 0.33, 0.2)), 0.1, 0.7, 0.3, 200.0, 0.0, 0.0, 1.0)
    t = time.perf_counter()
    parser = host.ObjFileParser.parse_obj_file(os.path.join(scenes.OBJ_DIR, "dragon.obj"), M.identity(), mat, True)
    group = parser.obj_to_group(M.identity())
    parse_ms = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    group.divide(1)
    divide_ms = (time.perf_counter() - t) * 1e3
    floor = host.Plane(M.translation(0.0, -0.2, 0.0), Phong.default(), True)
    world = host.World([floor, group], [host.PointLight(host.point(-10.0, 10.0, -10.0), Color.white())])
    out["dragon"] = finish(world, {"parse_ms": parse_ms, "divide_ms": divide_ms})
    t = time.perf_counter()
    world = scenes.loose_triangles_world(host)
    out["loose_triangles_20300"] = finish(world, {"construct_ms": (time.perf_counter() - t) * 1e3, "divide_ms": 0.0})
    for e in out.values():
        e["total_ms"] = sum(v for k, v in e.items() if k in ("parse_ms", "construct_ms", "divide_ms", "flatten_ms", "scene_create_ms"))
    return out


def parent(args, argv):
    """No launcher around us: start one worker per GPU and relay rank 0's line."""
    from raytracer_amd.launch import spawn_ranks
    status, out = spawn_ranks([os.path.abspath(__file__)] + argv, args.gpus, timeout=3000)
    for line in out.splitlines():
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.exit(status)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="default")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region and the latency pass (what the profiling scripts run)")
    ap.add_argument("--max-depth", type=int, default=5, help="diagnostic: `remaining` of primary rays (the metric uses 5)")
    ap.add_argument("--store", choices=["u8", "f64"], default="u8",
                    help="what every rank stores (and rank 0 collects) per pixel: the clamped bytes the PPM prints "
                         "(default) or the f64 colours")
    ap.add_argument("--gather", choices=["u8", "f64"], default=None, help="old name of --store")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: gather each frame before rendering the next")
    ap.add_argument("--gather-mode", choices=["p2p", "collective"], default="p2p",
                    help="N > 1: grouped send/recv of exactly each rank's rows (default) or one padded dist.gather")
    ap.add_argument("--in-flight", type=int, choices=[1, 2], default=1,
                    help="frames rendering at once on a GPU: 2 = consecutive frames on two streams, each with its own "
                         "scene handle and buffer (the next frame's wavefronts fill the GPU while this frame's longest "
                         "ones finish).  The same default, 1, at every N")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs from ONE process through rt_render_multi_u8 / rt_render_multi (frame to page-locked host "
                         "memory) instead of one process per GPU")
    args = ap.parse_args()
    if args.gather:
        args.store = args.gather

    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched and not args.single_process:
        return parent(args, sys.argv[1:])

    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, api, scenes

    world_size = int(os.environ.get("WORLD_SIZE", "1")) if not args.single_process else 1
    rank = int(os.environ.get("RANK", "0")) if not args.single_process else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if not args.single_process else 0
    if not args.single_process and args.gpus != world_size:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world_size))
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

    W = args.width or 7680
    H = args.height or 4320
    host = api.load()
    make_world, make_camera = scenes.WORKLOADS[args.workload]
    if args.single_process:
        return single_process(args, host, make_world, make_camera, W, H)

    in_flight = args.in_flight
    overlap = world_size > 1 and not args.no_overlap
    # one scene handle per frame in flight: launches on one handle must be stream-ordered
    worlds = [make_world(host) for _ in range(in_flight)]
    cam = make_camera(host, W, H)
    scenes_h = [host.scene_handle(w, device=local_rank) for w in worlds]
    scene = scenes_h[0]
    rows = host.local_rows(H, rank, world_size)
    u8 = args.store == "u8"
    dtype = torch.uint8 if u8 else torch.float64
    buffers = [torch.empty((rows, W, 3), dtype=dtype, device=dev) for _ in range(2)]

    from raytracer_amd.distributed import FrameGather, OverlappedGather
    gather = FrameGather(host, W, H, rank, world_size, dev, dtype=dtype, mode=args.gather_mode) if world_size > 1 else None
    render_streams = [torch.cuda.Stream(device=dev) for _ in range(2)] if in_flight > 1 else None
    pipe = OverlappedGather(gather, buffers, render_streams) if overlap or in_flight > 1 else None

    def render(buf, slot=0):
        stream = torch.cuda.current_stream().cuda_stream
        host.render_device(worlds[slot % in_flight], cam, buf.data_ptr(), stream=stream, device=local_rank, rank=rank,
                           world_size=world_size, max_depth=args.max_depth, u8=u8)

    def step():
        if pipe is not None:
            pipe.step(render)  # the exchange of frame k's tile buffers overlaps the render of frame k + 1
        else:
            render(buffers[0])
            if gather is not None:
                gather(buffers[0])

    def sync():
        torch.cuda.synchronize()  # all streams of this device, the gather's side stream included
        if world_size > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- the timed region ----------------------------------------------------------------------------
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    per_scene = max(args.steps // in_flight, 1)
    timed_ms = launch_times(host, scene, per_scene)
    kernel_ms_timed = sum(timed_ms) / max(len(timed_ms), 1)
    st = abi.rt_stats()
    host.abi.rt_scene_ray_counts(scene, ctypes.byref(st))
    rays_local = st.total_rays()

    # ---- one frame at a time, nothing overlapped: latency and its parts ---------------------------
    L = max(2, min(args.steps, 30))
    lat = []
    g_events = []
    for _ in range(L):
        sync()
        t = time.perf_counter()
        render(buffers[0])
        if gather is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gather(buffers[0])
            e1.record()
            g_events.append((e0, e1))
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t) * 1e3)
    clean_ms = launch_times(host, scene, L)
    kernel_ms = sum(clean_ms) / max(len(clean_ms), 1)
    latency_ms = sum(lat) / len(lat)
    gather_ms = sum(a.elapsed_time(b) for a, b in g_events) / len(g_events) if g_events else 0.0
    d2h_ms = 0.0
    if rank == 0:
        frame = gather.frame if gather is not None else buffers[0]
        pinned = torch.empty(frame.shape, dtype=frame.dtype, pin_memory=True)
        pinned.copy_(frame, non_blocking=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            pinned.copy_(frame, non_blocking=True)
        e1.record()
        torch.cuda.synchronize()
        d2h_ms = e0.elapsed_time(e1) / 5
        del pinned

    t = torch.tensor([elapsed, kernel_ms, kernel_ms_timed, float(rays_local)], dtype=torch.float64, device=stat_dev)
    if world_size > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, kernel_ms, kernel_ms_timed, rays = float(tmax[0]), float(tmax[1]), float(tmax[2]), float(tsum[3])
    else:
        rays = float(rays_local)

    if rank == 0:
        work = load_work(args.workload, W, H) if args.max_depth == 5 else None
        ms_per_step = elapsed / args.steps * 1e3
        exchange = ("RCCL" if backend == "nccl" else backend) + " " + (
            "grouped send/recv" if args.gather_mode == "p2p" else "gather")
        out = {
            "metric": "Mrays/s (primary+secondary), %s scene %dx%d" % (args.workload, W, H),
            "value": rays * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",  # the frame is the same at every N
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s scene%s, %dx%d, depth %d, %s frame store" % (
                    args.workload, " (src/lib.rs build_world)" if args.workload == "default" else " (raytracer_amd/scenes.py)",
                    W, H, args.max_depth, "clamped u8" if u8 else "f64"),
                "rays_per_frame": int(rays),
                "partition": "cyclic 8-row tiles over %d rank(s)%s" % (
                    world_size,
                    ", %s of %s rows to rank 0%s" % (exchange, "clamped u8" if u8 else "f64",
                                                    ", overlapped with the next frame's render" if overlap else "")
                    if world_size > 1 else ""),
                "frames_in_flight": in_flight,
            },
            # one frame at a time, nothing overlapped (mean of %d frames after the timed region)
            "frame_latency_ms": latency_ms,
            "kernel_ms": kernel_ms,        # mean launch duration (HIP events), longest rank
            "gather_ms": gather_ms,        # rank 0: exchange + placing the rows (device time)
            "d2h_ms": d2h_ms,              # rank 0: the assembled frame to page-locked host memory
            "latency_frames": L,
            "kernel_ms_in_timed_region": kernel_ms_timed,
            # `value` of round 1's record was measured at 3840x2160 with the f64 store; since round 2 the line is the
            # 7680x4320 frame with the clamped-byte store (the multi-GPU configuration, the same at every N).  The
            # old configuration is `single_gpu_4k`, the f64 store at this size `single_gpu_8k_f64`.
            "metric_changed_from": "round 1: default scene 3840x2160, f64 store (now under single_gpu_4k)",
        }
        if work is not None:
            frame_bytes = rows * W * 3 * (1 if u8 else 8)
            # (the PMC passes are whole-frame launches on one GPU; a rank of N renders 1/N of the tiles)
            out["roofline"] = roofline_block(work, scaled_pmc(load_pmc(args.workload, W, H), 1.0 / world_size),
                                             float(work["flops"]) / world_size, kernel_ms, frame_bytes, W, H,
                                             tables=table_bytes(host, scene), rays=rays / world_size,
                                             walks=load_walks(args.workload, W, H) if world_size == 1 else None)
            out["roofline"]["oracle_rays_per_frame"] = work["rays"]
        if world_size == 1 and args.workload == "default" and not args.no_extras:
            host.reset()
            out["single_gpu_4k"] = single_gpu_4k(host, torch, dev, local_rank)
            host.reset()
            out["single_gpu_8k_f64"] = single_gpu_8k_f64(host, torch, dev, local_rank)
            host.reset()
            out["e2e"] = e2e_block(host, torch, dev, local_rank)
            host.reset()
            out["scene_build"] = scene_build_block(host, local_rank)
            host.reset()
        if world_size == 1 and not args.no_cpu_baseline and not args.no_extras:
            out["cpu_baseline"] = cpu_baseline(args.workload, W, H)
        print(json.dumps(out))
    if world_size > 1:
        dist.destroy_process_group()


def single_process(args, host, make_world, make_camera, W, H):
    """N GPUs from one process: rt_multi_create + rt_render_multi(_u8) into page-locked host memory.  Prints the
    same keys as the process-per-GPU line, so that a scaling run of either form reads like the other."""
    import numpy as np
    n = args.gpus
    world, cam = make_world(host), make_camera(host, W, H)
    multi = host.multi(world, n)
    u8 = args.store == "u8"
    out_arr = host.pinned_empty((H, W, 3), np.uint8 if u8 else np.float64)
    for _ in range(args.warmup):
        multi.render(cam, max_depth=args.max_depth, out=out_arr.array, u8=u8)
    t0 = time.perf_counter()
    parts = []
    for _ in range(args.steps):
        _, st = multi.render(cam, max_depth=args.max_depth, out=out_arr.array, u8=u8, stats=True)
        parts.append((st.kernel_ms, st.exchange_ms, st.total_ms))
    elapsed = time.perf_counter() - t0
    rays = st.total_rays()
    mean = lambda k: sum(p[k] for p in parts) / len(parts)
    # every call is one frame at a time already (the call returns when the frame is in host memory)
    L = max(2, min(args.steps, 30))
    lat = []
    for _ in range(L):
        t = time.perf_counter()
        _, st = multi.render(cam, max_depth=args.max_depth, out=out_arr.array, u8=u8, stats=True)
        lat.append(((time.perf_counter() - t) * 1e3, st.kernel_ms, st.exchange_ms))
    latency_ms, kernel_ms, download_ms = (sum(p[k] for p in lat) / len(lat) for k in range(3))
    out = {
        "metric": "Mrays/s (primary+secondary), %s scene %dx%d" % (args.workload, W, H),
        "value": rays * args.steps / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s scene%s, %dx%d, depth %d, %s frame to page-locked HOST memory" % (
                       args.workload, " (src/lib.rs build_world)" if args.workload == "default" else " (raytracer_amd/scenes.py)",
                       W, H, args.max_depth, "clamped u8" if u8 else "f64"),
                   "rays_per_frame": int(rays),
                   "partition": "cyclic 8-row tiles over %d GPU(s) of ONE process (rt_render_multi%s): every GPU copies its "
                                "tiles over its own PCIe link, no exchange between GPUs" % (n, "_u8" if u8 else ""),
                   "frames_in_flight": 1},
        "frame_latency_ms": latency_ms,   # one call: render on every GPU + the frame in host memory
        "kernel_ms": kernel_ms,           # longest render launch over the GPUs (HIP events)
        "gather_ms": 0.0,                 # nothing is gathered on a GPU in this form
        "download_ms": download_ms,       # from the end of the longest render to the last byte in host memory
        "d2h_ms": download_ms,
        "latency_frames": L,
        "kernel_ms_in_timed_region": mean(0), "download_ms_in_timed_region": mean(1), "call_ms": mean(2),
    }
    work = load_work(args.workload, W, H) if args.max_depth == 5 else None
    if work is not None:
        rows = host.local_rows(H, 0, n)
        frame_bytes = rows * W * 3 * (1 if u8 else 8)
        tables = table_bytes(host, host.scene_handle(world, device=0))
        out["roofline"] = roofline_block(work, scaled_pmc(load_pmc(args.workload, W, H), 1.0 / n), float(work["flops"]) / n,
                                         kernel_ms, frame_bytes, W, H, tables=tables, rays=rays / n,
                                         walks=load_walks(args.workload, W, H) if n == 1 else None)
        out["roofline"]["oracle_rays_per_frame"] = work["rays"]
    print(json.dumps(out))
    out_arr.close()
    multi.close()


if __name__ == "__main__":
    main()
