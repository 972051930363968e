"""Record-visit counters of one frame from a -DRT_PROFILE_MIX build.  usage: RT_AMD_LIB=variants/mix.so python tools/mix_counts.py workload W H"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes  # noqa: E402
name, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
host = api.load()
mw, mc = scenes.WORKLOADS[name]
img, st = host.render(mw(host), mc(host, W, H), stats=True)
print(name, W, H, "kernel_ms", st.kernel_ms, file=sys.stderr)
