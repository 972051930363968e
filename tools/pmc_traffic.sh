#!/bin/bash
# usage (under gpurun, repo root): tools/pmc_traffic.sh lib.so [bench args]  -- L2<->memory bytes of the render kernel
# (FETCH_SIZE and WRITE_SIZE in passes of their own; tools/pmc_summary.py prints the sums per kernel)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export RT_AMD_LIB=$ROOT/$1; shift
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-extras "$@" > $OUT/$c.log 2>&1 || exit 1
  python $ROOT/tools/pmc_summary.py $OUT/$c/p_counter_collection.csv
done
