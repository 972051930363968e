#!/bin/bash
# usage (under gpurun, repo root): tools/issue_calib.sh
# What the SQ counters read for kernels whose vector-ALU load is known: tools/ubench/valu_rates runs long
# streams of ONE instruction that fill every SIMD, so the issue fraction computed from the counters
# must come out at 1 for them.  Writes gpurun_out/issue_calib/{avail.txt,calib.txt}.
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/issue_calib
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 60 rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
timeout -k 10 120 $ROOT/tools/ubench/valu_rates > $OUT/rates.txt 2>&1
: > $OUT/calib.txt
i=0
for counters in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc_$i -o p -- $ROOT/tools/ubench/valu_rates > $OUT/pmc_$i.log 2>&1
  python $ROOT/tools/pmc_summary_all.py $OUT/pmc_$i/p_counter_collection.csv >> $OUT/calib.txt
done
# the same for a kernel that is nothing but a chain of dependent scalar fetches (the scene walk's record fetch)
timeout -k 10 120 $ROOT/tools/ubench/smem_latency > $OUT/smem_latency.txt 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM_NORM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_smem -o p -- $ROOT/tools/ubench/smem_latency > $OUT/pmc_smem.log 2>&1
python $ROOT/tools/pmc_summary_all.py $OUT/pmc_smem/p_counter_collection.csv > $OUT/calib_smem.txt
echo calib done
