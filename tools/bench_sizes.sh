#!/bin/bash
# usage (under gpurun, repo root): tools/bench_sizes.sh v1 v2 ...  -- default scene at several frame sizes per variants/<v>.so
cp raytracer_amd/csrc/librt_amd.so /tmp/orig.so
for v in "$@"; do
  cp variants/$v.so raytracer_amd/csrc/librt_amd.so
  line="$v :"
  for hw in "7680 4320" "3840 2160" "1920 1080" "960 540"; do
    set -- $hw
    r=$(timeout -k 10 100 python bench.py --no-extras --store f64 --steps 50 --warmup 5 --width $1 --height $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],3))")
    line="$line $1x$2 $r"
  done
  r=$(timeout -k 10 100 python bench.py --no-extras --store f64 --steps 50 --warmup 5 --workload glass --width 3840 --height 2160 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],3))")
  echo "$line glass4k $r"
done
cp /tmp/orig.so raytracer_amd/csrc/librt_amd.so
