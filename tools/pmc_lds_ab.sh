#!/bin/bash
# LDS counters of k_run for one build of the library (tuning):   bash tools/pmc_lds_ab.sh <lib.so> <tag>
export REINLIFE_HIP_LIB=$(realpath $1)
export RL_WORLDS=256
TAG=$2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $OUT/b -- python $GRAFT_REPO_ROOT/tools/pmc_driver.py > $OUT/b.log 2>&1 || tail -5 $OUT/b.log
cd $GRAFT_REPO_ROOT
python tools/pmc_sq_report.py $OUT | grep -A 14 "k_run"
