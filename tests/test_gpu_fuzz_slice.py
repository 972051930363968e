"""A bounded slice of the three differential fuzz tools, so that the claims they back -- the box tests
in single precision with a bounded margin decide nothing the reference's f64 tests would decide
differently, cull boxes never remove a hit, the whole path agrees with the oracle on random scenes and
views -- are checked by the suite itself and not only by logs of longer runs.  Fresh seed ranges (the
other tests use small seeds; the long runs recorded under profiles/ used 0 .. 104,000)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_culling_stress_rays_and_edge_rays_200_seeds():
    # 200 seeds x 4 scene generators x 1,200 rays from 10 .. 1e7 units away (zero and nearly parallel
    # components), then 20 transformed-group scenes x 4,500 rays on and next to box edges, bit for bit
    assert tool("fuzz_parity").run(700000, 700200) == 0


def test_random_whole_path_and_csg_scenes_200_seeds():
    # every shape kind, nested and divided groups, nested patterns, mirrors and glass, 0-4 lights; CSG trees
    assert tool("fuzz_scenes").run(700000, 700200) == 0


def test_benchmarked_scenes_from_random_viewpoints_200_seeds():
    # 200 seeds x the seven benchmarked scenes: colours and ray counts of ragged frames at depths 0-5
    assert tool("fuzz_views").run(700000, 700200) == 0
