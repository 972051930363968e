"""Extended run of tests/test_gpu_views.py: the benchmarked scenes seen from random viewpoints (cameras
inside the scene, close to surfaces, far away, looking along the axes; random fields of view and frame
shapes; depths 0 to 5), whole small frames and ray counts against the oracle.
usage (GPU box): python tools/fuzz_views.py first_seed last_seed
(tests/test_gpu_fuzz_slice.py calls run() for a short range of seeds)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding  # noqa: E402
from raytracer_amd import api, scenes  # noqa: E402
import test_gpu_views as v  # noqa: E402


def run(first, last):
    """Returns the number of frames with a mismatch."""
    orc, host = binding.load(), api.load()
    bad = 0
    t0 = time.time()
    worlds = {name: (mw(host), mw(orc)) for name, (mw, _) in scenes.WORKLOADS.items()}
    for seed in range(first, last):
        for name, (hw, ow) in worlds.items():
            diff = v.compare_view(host, orc, hw, ow, name, seed)
            if diff:
                bad += 1
                print("MISMATCH " + diff, flush=True)
        if (seed - first) % 20 == 19:
            print("seed %d done, %.0f s, %d mismatching frames" % (seed, time.time() - t0, bad), flush=True)
    print("seeds %d..%d x %d workloads: %d mismatching frames" % (first, last - 1, len(worlds), bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2])) else 0)
