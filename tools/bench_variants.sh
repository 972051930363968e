#!/bin/bash
# usage (under gpurun, repo root): tools/bench_variants.sh v1 v2 ...   -- benches variants/<v>.so in turn
set -e
cp raytracer_amd/csrc/librt_amd.so /tmp/orig.so
for v in "$@"; do
  cp variants/$v.so raytracer_amd/csrc/librt_amd.so
  echo "== $v"
  timeout -k 10 120 python bench.py --no-extras --store f64 --width 3840 --height 2160 --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default4k', round(d['kernel_ms'],3), round(d['value']))"
  for w in glass csg field teapot smooth_teapot dragon; do
    timeout -k 10 120 python bench.py --no-extras --store f64 --steps 10 --workload $w --width 1920 --height 1080 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['kernel_ms'],3), round(d['value']))"
  done
done
cp /tmp/orig.so raytracer_amd/csrc/librt_amd.so
