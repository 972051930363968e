"""The library's diagnostic switches must not change a pixel: every mechanism they turn off (cull
boxes, sorted siblings, cost-ordered tiles, the short form of scale-and-translation transforms,
strips and work sharing) only decides what work is skipped or when it is done.
The library reads them once per process, so each setting renders in a child process."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, sys
sys.path.insert(0, %r)
from raytracer_amd import api, scenes
rt = api.load()
for name, (W, H) in (("default", (1024, 576)), ("glass", (320, 200)), ("csg", (200, 120)), ("teapot", (160, 96)),
                     ("field", (192, 108))):
    mw, mc = scenes.WORKLOADS[name]
    img = rt.render(mw(rt), mc(rt, W, H))
    print(name, hashlib.sha256(img.tobytes()).hexdigest())
""" % ROOT


def render_hashes(env_extra):
    env = dict(os.environ)
    for k in ("RT_NO_CULL", "RT_REORDER", "RT_TILE_ORDER", "RT_DIAG", "RT_SPLIT", "RT_PROBE", "RT_TUNE"):
        env.pop(k, None)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return dict(line.split() for line in out.stdout.strip().splitlines())


def test_switches_do_not_change_a_pixel():
    base = render_hashes({})
    assert len(base) == 5
    for env in ({"RT_NO_CULL": "1"}, {"RT_REORDER": "0"}, {"RT_TILE_ORDER": "0"}, {"RT_DIAG": "0"}, {"RT_SPLIT": "1"},
                {"RT_SPLIT": "4", "RT_PROBE": "0"}, {"RT_SPLIT": "2", "RT_PROBE": "3"},
                {"RT_NO_CULL": "1", "RT_REORDER": "0", "RT_TILE_ORDER": "0"}, {"RT_TUNE": "0.5,1,1,1,1.0"},
                {"RT_TUNE": "6,1,1.5,1.2,0.3"}):
        assert render_hashes(env) == base, env
