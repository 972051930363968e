#!/bin/bash
# usage (under gpurun, repo root): tools/bench_env.sh VAR v1 v2 ...  -- every workload's kernel ms per value of an environment knob
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  line="$VAR=$v :"
  r=$(timeout -k 10 120 python bench.py --no-extras --store f64 --width 3840 --height 2160 --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],3))") || exit 1
  line="$line default4k $r"
  for w in glass csg field teapot smooth_teapot dragon; do
    r=$(timeout -k 10 120 python bench.py --no-extras --store f64 --steps 10 --workload $w --width 1920 --height 1080 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],3))") || exit 1
    line="$line $w $r"
  done
  echo "$line"
done
