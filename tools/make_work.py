"""Counts the algorithmic work of one frame with the CPU oracle and writes it to
profiles/work_<workload>_<W>x<H>.json.  bench.py prices the kernel against these figures, so the
numbers it reports do not depend on anything the GPU measures about itself.

usage: python tools/make_work.py default 3840 2160
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import binding  # noqa: E402
from raytracer_amd import scenes  # noqa: E402


def main():
    name, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    orc = binding.load()
    mw, mc = scenes.WORKLOADS[name]
    world, cam = mw(orc), mc(orc, W, H)
    t = time.time()
    _, cnt = orc.render(world, cam, counters=True)
    cnt["seconds_oracle_fast_mode"] = round(time.time() - t, 2)
    cnt["rays"] = cnt["rays_primary"] + cnt["rays_shadow"] + cnt["rays_reflect"] + cnt["rays_refract"]
    cnt["workload"] = name
    cnt["width"], cnt["height"] = W, H
    path = os.path.join(ROOT, "profiles", "work_%s_%dx%d.json" % (name, W, H))
    with open(path, "w") as f:
        json.dump(cnt, f, indent=1, sort_keys=True)
    print(path, cnt)


if __name__ == "__main__":
    main()
