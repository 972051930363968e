"""Extended differential run of the randomised whole-path tests: the scene generators of
tests/test_gpu_kat.py::test_random_scene (every shape kind, nested and divided groups, nested patterns,
reflective and transparent materials, 0 to 4 lights) and tests/test_gpu_csg.py::test_random_csg_scene
for many more seeds, small frames at depths 0 to 5, every colour against the oracle.
usage (GPU box): python tools/fuzz_scenes.py first_seed last_seed
(tests/test_gpu_fuzz_slice.py calls run() for a short range of seeds)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from oracle import binding  # noqa: E402
from raytracer_amd import abi, api  # noqa: E402
import test_gpu_kat as k  # noqa: E402
import test_gpu_csg as c  # noqa: E402



def cam_kat(a):
    return a.Camera(40, 56, 1.0, a.Matrix.view_transform(a.point(0, 1.5, -9), a.point(0, 0, 0), a.vector(0, 1, 0)))


def cam_csg(a):
    return a.Camera(48, 64, 1.0, a.Matrix.view_transform(a.point(0, 1.0, -8), a.point(0, 0, 0), a.vector(0, 1, 0)))



def run(first, last):
    """Returns the number of scenes with a mismatch."""
    orc, host = binding.load(), api.load()
    bad = refused = 0
    t0 = time.time()
    for seed in range(first, last):
        for kind, make, cam in (("scene", k.random_world, cam_kat), ("csg", c.random_csg_world, cam_csg)):
            depth = seed % 6
            try:
                img = host.render(make(host, seed), cam(host), max_depth=depth)
            except abi.RtError as e:
                if e.code != abi.RT_ERR_LIMIT:
                    raise
                refused += 1  # more CSG intersections per ray than the per-lane list holds: reported, not rendered
                host.reset()
                continue
            ref = orc.render(make(orc, seed), cam(orc), remaining=depth, threads=8)
            host.reset()
            orc.reset()
            finite = np.isfinite(ref)
            same_mask = np.array_equal(np.isfinite(img), finite)
            err = np.abs(img[finite] - ref[finite]) / np.maximum(1.0, np.abs(ref[finite]))
            worst = err.max() if err.size else 0.0
            if not same_mask or worst > 1e-9:
                bad += 1
                print("MISMATCH seed %d %s depth %d: max rel err %.3e, finite masks equal %s" % (seed, kind, depth, worst, same_mask),
                      flush=True)
        if (seed - first) % 50 == 49:
            print("seed %d done, %.0f s, %d mismatching scenes, %d refused for a limit" % (seed, time.time() - t0, bad, refused),
                  flush=True)
    print("seeds %d..%d: %d mismatching scenes, %d refused for a limit" % (first, last - 1, bad, refused))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2])) else 0)
