#!/bin/bash
# HBM traffic of the hot kernels from PMC counters (run on the GPU box through gpurun).  Separate passes per counter
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/$C -- python $GRAFT_REPO_ROOT/tools/pmc_driver.py > $OUT/$C.log 2>&1 || { tail -5 $OUT/$C.log; }
done
cd $GRAFT_REPO_ROOT
python tools/pmc_report.py $OUT
