#!/bin/bash
# usage (under gpurun, repo root): tools/traffic_env.sh "<bench args>" "ENV=VALUE ..." ["ENV=VALUE ..."] ...
# HBM traffic counters (FETCH_SIZE, WRITE_SIZE in KB, SQ_INSTS_VMEM) per rt_kernel launch for one bench
# configuration under several settings of the library's environment switches ("-" = none).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
ARGS=$1; shift
export PYTHONPATH=$ROOT TMPDIR=/tmp
cd /tmp
n=0
for setting in "$@"; do
  n=$((n+1))
  for counters in "FETCH_SIZE" "WRITE_SIZE SQ_INSTS_VMEM SQ_WAVES"; do
    out=$ROOT/gpurun_out/traffic_env/${n}_${counters%% *}
    rm -rf $out; mkdir -p $out
    ( [ "$setting" != "-" ] && export $setting; timeout -s KILL 120 rocprofv3 --pmc $counters --output-format csv -d $out -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-extras $ARGS > $out/log 2>&1 )
    printf "%-32s " "$setting"; python $ROOT/tools/pmc_summary.py $(find $out -name 'p_counter_collection.csv') | tr '\n' ' '; echo
  done
done
