#!/bin/bash
# usage (under gpurun, repo root): tools/ab_all.sh lib1.so lib2.so ...  -- kernel ms of every benchmarked configuration per library
run() {
  timeout -k 10 150 python bench.py --no-extras --steps ${STEPS:-30} --warmup 6 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.3f' % d['kernel_ms'], end=' ')"
}
for lib in "$@"; do
  export RT_AMD_LIB=$PWD/$lib
  printf "%-22s " $lib
  run --store f64 --width 3840 --height 2160
  run --store f64 --width 1920 --height 1080
  run
  for w in glass csg teapot smooth_teapot dragon; do run --store f64 --workload $w --width 3840 --height 2160; done
  for w in glass csg field dragon; do run --store f64 --workload $w --width 1920 --height 1080; done
  echo
done
echo "columns: default 4K, 1080p, 8K(u8) | 4K: glass csg teapot smooth_teapot dragon | 1080p: glass csg field dragon"
