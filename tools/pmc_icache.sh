#!/bin/bash
# Instruction-cache and instruction-fetch counters of the multi-tick launch (GPU box, through gpurun):   bash tools/pmc_icache.sh
set -e
export RL_WORLDS=${1:-256}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_icache_${RL_PMC_WORKLOAD:-c4}
rm -rf $OUT && mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*" | sort -u > $OUT/available.txt || true
head -60 $OUT/available.txt
pass() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -- python $GRAFT_REPO_ROOT/tools/pmc_driver.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; } }
pass a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
pass b SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_INSTS_SMEM SQC_TC_INST_REQ SQC_TC_REQ
cd $GRAFT_REPO_ROOT
python tools/pmc_sq_report.py $OUT 2>&1 | tail -40
