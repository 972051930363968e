#!/bin/bash
# usage (under gpurun, repo root): tools/ab_xf.sh lib1.so lib2.so ...  -- kernel ms of the frames whose scenes have transformed groups or CSG shapes
run() {
  timeout -k 10 150 python bench.py --no-extras --steps ${STEPS:-30} --warmup 6 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.3f' % d['kernel_ms'], end=' ')"
}
for lib in "$@"; do
  export RT_AMD_LIB=$PWD/$lib
  printf "%-22s " $lib
  run --store f64 --workload csg --width 3840 --height 2160
  run --store f64 --workload csg --width 1920 --height 1080
  run --store f64 --workload smooth_teapot --width 3840 --height 2160
  run --store f64 --workload smooth_teapot --width 1920 --height 1080
  run --store f64 --width 3840 --height 2160
  echo
done
echo "columns: csg 4K, csg 1080p, smooth teapot 4K, 1080p, default 4K"
