#!/bin/bash
# usage (under gpurun, repo root): tools/bench_quick.sh  -- kernel ms of the main configurations, one line each
# (environment variables such as RT_SPLIT or RT_AMD_LIB apply to every run)
set -e
run() {
  timeout -k 10 150 python bench.py --no-extras --steps ${STEPS:-60} --warmup 10 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-34s kernel %.3f ms  latency %.3f ms  step %.3f ms  %.0f Mrays/s' % (sys.argv[1], d['kernel_ms'], d['frame_latency_ms'], d['ms_per_step'], d['value']))" "$*"
}
run --store f64 --width 3840 --height 2160
run --store f64 --width 1920 --height 1080
run --store f64 --width 960 --height 540
run
for w in glass csg field dragon; do run --store f64 --workload $w --width 1920 --height 1080; done
run --store f64 --workload glass --width 3840 --height 2160
