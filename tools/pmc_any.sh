#!/bin/bash
# usage (under gpurun, repo root): tools/pmc_any.sh "COUNTER ..." [bench args]  -- one --pmc pass over a 2-frame bench run
ROOT=${GRAFT_REPO_ROOT:-$PWD}
C=$1; shift
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_any
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-extras "$@" > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
python $ROOT/tools/pmc_summary.py $OUT/p_counter_collection.csv
