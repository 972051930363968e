"""Extended randomised parity run against the CPU oracle: the seeded scene generators of the GPU tests
with many more seeds than the test suite uses.  usage (on a GPU box, repo root): python tools/stress_gpu.py"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from oracle import binding
from raytracer_amd import api
import test_gpu_kat as K
import test_gpu_csg as C
host, orc = api.load(), binding.load()
bad = 0
for seed in range(40, 400):
    try:
        K.test_random_scene(host, orc, seed)
    except AssertionError as e:
        bad += 1; print("random_scene", seed, str(e)[:200])
    host.reset(); orc.reset()
for seed in range(8, 60):
    for kind in ("flat", "grouped", "mixed"):
        try:
            K.test_culling_is_conservative(host, orc, seed, kind)
        except AssertionError as e:
            bad += 1; print("culling", seed, kind, str(e)[:200])
        host.reset(); orc.reset()
for seed in range(24, 150):
    try:
        C.test_random_csg_scene(host, orc, seed)
    except AssertionError as e:
        bad += 1; print("csg", seed, str(e)[:200])
    host.reset(); orc.reset()
print("failures", bad)
sys.exit(1 if bad else 0)
