#!/bin/bash
# usage (under gpurun, repo root): tools/pmc_valu.sh lib.so [bench args]  -- vector/scalar instruction counts of the render kernel
ROOT=${GRAFT_REPO_ROOT:-$PWD}
export RT_AMD_LIB=$ROOT/$1; shift
export PYTHONPATH=$ROOT TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_valu
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-extras "$@" > $OUT/log 2>&1
python $ROOT/tools/pmc_summary.py $OUT/p_counter_collection.csv
