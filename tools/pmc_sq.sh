#!/bin/bash
# SQ issue / stall / LDS counters of the hot kernels (run on the GPU box through gpurun):   bash tools/pmc_sq.sh [worlds] [tag]
# Passes of <= 8 SQ counters each, --kernel-trace only (no other trace domains next to --pmc).
set -e
export RL_WORLDS=${1:-256}
TAG=${2:-sq}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$RL_WORLDS
rm -rf $OUT && mkdir -p $OUT
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -- python $GRAFT_REPO_ROOT/tools/pmc_driver.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; }
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
pass b SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA
pass c SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_FLAT SQ_INST_LEVEL_LDS
pass d SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES_EQ_64 SQ_THREAD_CYCLES_VALU
pass g GRBM_GUI_ACTIVE GRBM_COUNT
cd $GRAFT_REPO_ROOT
python tools/pmc_sq_report.py $OUT | tee $OUT/report.txt
