#!/bin/bash
# usage (under gpurun, repo root): tools/ab_csg.sh lib1.so lib2.so ...  -- kernel ms of the CSG frames per library
run() {
  timeout -k 10 150 python bench.py --no-extras --steps ${STEPS:-30} --warmup 6 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.3f' % d['kernel_ms'], end=' ')"
}
for lib in "$@"; do
  export RT_AMD_LIB=$PWD/$lib
  printf "%-22s " $lib
  for size in "3840 2160" "1920 1080" "960 540"; do set -- $size; run --store f64 --workload csg --width $1 --height $2; done
  run --store f64 --workload field --width 1920 --height 1080
  echo
done
echo "columns: csg 4K, 1080p, 960x540, field 1080p"
