#!/bin/bash
# usage (under gpurun, repo root): tools/trace_one.sh <tag> [bench args...]  -- rocprofv3 kernel stats of a short bench run
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONPATH=$ROOT TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $ROOT/bench.py --steps 20 --warmup 3 --no-extras "$@" > $OUT.log 2>&1
cut -d, -f1-4 $OUT/t_kernel_stats.csv | sed 's/(anonymous namespace):://g' | cut -c1-150
