#!/bin/bash
# Collects kernel stats and PMC means for a list of bench configurations on the GPU box and writes
# ready-to-commit summaries.
# usage (under gpurun, repo root): tools/profile_set.sh <outdir> <workload>:<W>x<H>[:u8] ...
# (frames are stored as f64 unless the configuration ends in :u8, which is bench.py's default store)
# Output per configuration: <outdir>/kernel_stats_<workload>_<W>x<H>.csv and pmc_<workload>_<W>x<H>.txt
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUTDIR=$ROOT/$1; shift
mkdir -p $OUTDIR
for cfg in "$@"; do
  store=f64; case $cfg in *:u8) store=u8; cfg=${cfg%:u8};; esac
  w=${cfg%%:*}; size=${cfg##*:}; W=${size%%x*}; H=${size##*x}
  tag=prof_${w}_${size}
  bash $ROOT/tools/profile_gpu.sh $tag --workload $w --width $W --height $H --store $store
  cp $ROOT/gpurun_out/$tag/trace/t_kernel_stats.csv $OUTDIR/kernel_stats_${w}_${size}.csv
  : > $OUTDIR/pmc_${w}_${size}.txt
  for i in 1 2 3 4 5 6 7; do
    python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/$tag/pmc_$i/p_counter_collection.csv >> $OUTDIR/pmc_${w}_${size}.txt
  done
  echo "profiled $cfg"
done
