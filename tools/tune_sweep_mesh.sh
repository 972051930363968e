#!/bin/bash
# usage (under gpurun): tools/tune_sweep_mesh.sh "need,cost_quad,cost_tri,cost_group,worth" ...  -- kernel ms of the mesh frames per setting
run() {
  timeout -k 10 150 python bench.py --no-extras --steps 40 --warmup 8 --store f64 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%.3f' % d['kernel_ms'], end=' ')"
}
for t in "$@"; do
  export RT_TUNE=$t
  printf "%-24s " "$t"
  run --workload teapot --width 1920 --height 1080
  run --workload smooth_teapot --width 1920 --height 1080
  run --workload dragon --width 1920 --height 1080
  run --workload teapot --width 3840 --height 2160
  run --workload smooth_teapot --width 3840 --height 2160
  run --workload dragon --width 3840 --height 2160
  echo
done
