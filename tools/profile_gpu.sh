#!/bin/bash
# Collects the rocprofv3 evidence for one bench configuration on the GPU box.
# usage: tools/profile_gpu.sh <tag> [bench args...]      (run from the repo root under gpurun)
# Output: gpurun_out/<tag>/{trace,pmc_*}/...csv ; copy the summaries into profiles/ afterwards.
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONPATH=$ROOT TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --steps 50 --warmup 3 --no-extras $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
PMC_BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-extras $*"
i=0
for counters in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SMEM" \
  "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_SMEM_NORM SQ_ACTIVE_INST_VMEM" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $counters --output-format csv -d $OUT/pmc_$i -o p -- $PMC_BENCH > $OUT/pmc_$i.log 2>&1
done
echo profile done
