import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import binding
from raytracer_amd import api
import test_gpu_box_margins as t
orc = binding.load(); host = api.load()
np.set_printoptions(precision=17, linewidth=200)
def run(name, make, o, d):
    got = host.color_at(make(host), o, d, remaining=2)
    ref = orc.color_at_batch(make(orc), o, d, remaining=2)
    same = ((got == ref) | (np.isnan(got) & np.isnan(ref))).all(axis=1)
    bad = np.argwhere(~same)[:, 0]
    print(name, "bad", len(bad), "of", len(o))
    for i in bad[:8]:
        print("  ray", i, "o", o[i], "d", d[i]); print("     got", got[i], "ref", ref[i])
    # remaining=0 to see whether the primary hit differs
    got0 = host.color_at(make(host), o[bad[:8]], d[bad[:8]], remaining=0)
    ref0 = orc.color_at_batch(make(orc), o[bad[:8]], d[bad[:8]], remaining=0)
    print("  depth0 got", got0.tolist()); print("  depth0 ref", ref0.tolist())
import inspect
src = inspect.getsource(t.test_rays_beyond_the_range_of_the_shortcut)
body = src[src.index('    rng ='):src.index('    same_colours')]
import textwrap
ns = {'np': np}
exec(textwrap.dedent(body), ns)
run("beyond", t.boxes_world, ns['o'], ns['d'])
