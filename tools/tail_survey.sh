#!/bin/bash
# usage (under gpurun, repo root): tools/tail_survey.sh  -- ms per frame with one and with two frames in flight for every
# benchmarked configuration: the difference is what a frame spends filling and draining the GPU.
run() {
  timeout -k 10 150 python bench.py --no-extras --steps 40 --warmup 8 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.3f' % d['ms_per_step'], end=' ')"
}
for cfg in default:3840x2160 default:1920x1080 glass:3840x2160 glass:1920x1080 csg:3840x2160 csg:1920x1080 field:1920x1080 teapot:3840x2160 smooth_teapot:3840x2160 dragon:3840x2160 dragon:1920x1080; do
  w=${cfg%%:*}; size=${cfg##*:}; W=${size%%x*}; H=${size##*x}
  printf "%-28s " $cfg
  run --store f64 --workload $w --width $W --height $H --in-flight 1
  run --store f64 --workload $w --width $W --height $H --in-flight 2
  echo
done
