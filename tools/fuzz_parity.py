"""Extended differential run of the culling stress test: the same scene generators and ray recipe as
tests/test_gpu_kat.py::test_culling_is_conservative for many more seeds.  usage (GPU box):
python tools/fuzz_parity.py first_seed last_seed
(tests/test_gpu_fuzz_slice.py calls run() for a short range of seeds)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from oracle import binding  # noqa: E402
from raytracer_amd import api  # noqa: E402
import test_gpu_kat as k  # noqa: E402

import test_gpu_box_margins as bm  # noqa: E402


def run(first, last):
    """Returns the number of scenes with a mismatch."""
    orc, host = binding.load(), api.load()
    makers = {"flat": k.quadric_world, "grouped": k.grouped_world, "mixed": k.mixed_world, "sheared": k.sheared_world}
    bad_total = 0
    t0 = time.time()
    for seed in range(first, last):
        for kind, make in makers.items():
            rng = np.random.default_rng(seed)
            origins, dirs = [], []
            for dist in (10.0, 1e3, 1e5, 1e7):
                for _ in range(300):
                    target = rng.uniform(-5, 5, 3)
                    o = rng.normal(size=3)
                    o = o / np.linalg.norm(o) * dist
                    if rng.random() < 0.3:
                        o[rng.integers(0, 3)] = 0.0
                    d = target - o
                    if rng.random() < 0.2:
                        d[rng.integers(0, 3)] = 0.0
                    if rng.random() < 0.1:
                        d[rng.integers(0, 3)] = rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-7, -4) * dist  # nearly parallel components
                    origins.append(o)
                    dirs.append(d / np.linalg.norm(d))
            o, d = np.array(origins), np.array(dirs)
            got = host.color_at(make(host, seed), o, d, remaining=3)
            ref = orc.color_at_batch(make(orc, seed), o, d, remaining=3)
            host.reset()
            orc.reset()
            finite = np.isfinite(ref)
            same_mask = np.array_equal(np.isfinite(got), finite)
            err = np.abs(got[finite] - ref[finite]) / np.maximum(1.0, np.abs(ref[finite]))
            worst = err.max() if err.size else 0.0
            if not same_mask or worst > 1e-9:
                bad_total += 1
                rows = np.argwhere((np.abs(got - ref) / np.maximum(1.0, np.abs(ref)) > 1e-9).any(axis=1))[:3, 0]
                print("MISMATCH seed %d %s: max rel err %.3e, finite masks equal %s, rays %s" % (seed, kind, worst, same_mask, rows.tolist()), flush=True)
                for r in rows:
                    print("   o", o[r].tolist(), "d", d[r].tolist(), "got", got[r].tolist(), "ref", ref[r].tolist(), flush=True)
        if (seed - first) % 10 == 9:
            print("seed %d done, %.0f s, %d mismatching scenes so far" % (seed, time.time() - t0, bad_total), flush=True)
    print("seeds %d..%d: %d mismatching scenes" % (first, last - 1, bad_total))

    # second part: the transformed-group scenes of tests/test_gpu_box_margins.py, colours equal bit for bit
    bad_first = bad_total
    bad_total = 0
    for seed in range(first, first + max(1, (last - first) // 10)):
        rng = np.random.default_rng(100 + seed)
        origins, dirs = [], []
        for _ in range(300):
            t = rng.uniform(-7, 7, 3) * np.array([1.0, 0.5, 0.6])
            for dist in (14.0, 300.0, 1e5):
                o = rng.normal(size=3)
                o = o / np.linalg.norm(o) * dist
                for eps in (0.0, 1e-12, 1e-7, 1e-5, 1e-3):
                    d = t + eps * rng.choice([-1.0, 0.0, 1.0], 3) * dist - o
                    if rng.random() < 0.15:
                        d[rng.integers(0, 3)] = 0.0
                    if rng.random() < 0.1:
                        d[rng.integers(0, 3)] = rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-7, -4) * dist
                    n = np.linalg.norm(d)
                    if n > 0:
                        origins.append(o)
                        dirs.append(d / n)
        o, d = np.array(origins), np.array(dirs)
        got = host.color_at(bm.nested_world(host, seed), o, d, remaining=3)
        ref = orc.color_at_batch(bm.nested_world(orc, seed), o, d, remaining=3)
        host.reset()
        orc.reset()
        same = ((got == ref) | (np.isnan(got) & np.isnan(ref))).all(axis=1)
        if not same.all():
            bad_total += 1
            rows = np.argwhere(~same)[:3, 0]
            print("MISMATCH nested seed %d: %d rays, first %s" % (seed, (~same).sum(), rows.tolist()), flush=True)
            for r in rows:
                print("   o", o[r].tolist(), "d", d[r].tolist(), "got", got[r].tolist(), "ref", ref[r].tolist(), flush=True)
        if (seed - first) % 10 == 9:
            print("nested seed %d done, %d mismatching scenes so far" % (seed, bad_total), flush=True)
    print("nested seeds %d..%d: %d mismatching scenes" % (first, first + max(1, (last - first) // 10) - 1, bad_total))
    return bad_first + bad_total


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]), int(sys.argv[2])) else 0)
