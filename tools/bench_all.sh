#!/bin/bash
# usage (under gpurun, repo root): tools/bench_all.sh <outdir>  -- the bench line of every workload, one JSON file each
set -e
OUT=$1; mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/bench_default_7680x4320.json
timeout -k 10 300 python bench.py --no-extras --store f64 --width 3840 --height 2160 > $OUT/bench_default_3840x2160.json
timeout -k 10 300 python bench.py --no-extras --store f64 --width 1920 --height 1080 > $OUT/bench_default_1920x1080.json
timeout -k 10 300 python bench.py --single-process --gpus 1 --steps 30 --warmup 5 > $OUT/bench_single_process_1gpu_u8.json
timeout -k 10 300 python bench.py --single-process --gpus 1 --steps 30 --warmup 5 --store f64 --width 3840 --height 2160 > $OUT/bench_single_process_1gpu_4k_f64.json
for w in glass csg field teapot smooth_teapot dragon; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-extras --store f64 --workload $w --width 1920 --height 1080 > $OUT/bench_${w}_1920x1080.json
done
for w in glass csg teapot smooth_teapot dragon; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-extras --store f64 --workload $w --width 3840 --height 2160 > $OUT/bench_${w}_3840x2160.json
done
