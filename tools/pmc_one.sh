#!/bin/bash
# usage (under gpurun, repo root): tools/pmc_one.sh <tag> "<counters>" [bench args...]
# One rocprofv3 counter pass over a short bench run; prints the per-dispatch means.
set -e
TAG=$1; CNT=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONPATH=$ROOT TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc $CNT --output-format csv -d $OUT -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-extras "$@" > $OUT.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/p_counter_collection.csv
