#!/bin/bash
# usage (under gpurun, repo root): tools/bench_ab4.sh lib1.so lib2.so ...  -- a short list of configurations per library
set -e
run() {
  timeout -k 10 150 python bench.py --no-extras --steps ${STEPS:-40} --warmup 10 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   %-50s kernel %.3f ms' % (sys.argv[1], d['kernel_ms']))" "$*"
}
for lib in "$@"; do
  echo "== $lib"
  export RT_AMD_LIB=$lib
  run --store f64 --width 3840 --height 2160
  run --store f64 --width 1920 --height 1080
  run --store f64 --workload field --width 1920 --height 1080
  run --store f64 --workload glass --width 1920 --height 1080
  run --store f64 --workload dragon --width 3840 --height 2160
done
