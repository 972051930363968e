#!/bin/bash
# usage (under gpurun, repo root): tools/bench_all.sh <outdir>  -- the bench line of every workload, one JSON file each
set -e
OUT=$1; mkdir -p $OUT
timeout -k 10 300 python bench.py --steps 50 --warmup 3 > $OUT/bench_default_3840x2160.json
for w in glass csg field teapot smooth_teapot dragon; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --workload $w --width 1920 --height 1080 > $OUT/bench_${w}_1920x1080.json
done
for w in glass csg dragon; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --workload $w --no-cpu-baseline > $OUT/bench_${w}_3840x2160.json
done
