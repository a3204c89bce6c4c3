#!/bin/bash
# SQ occupancy / stall counters of the hot kernels (tuning aid; run on the GPU box through gpurun).
#   bash tools/pmc_sq.sh [worlds]
set -e
export RL_WORLDS=${1:-256}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$RL_WORLDS
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/a -- python $GRAFT_REPO_ROOT/tools/pmc_driver.py > $OUT/a.log 2>&1 || tail -5 $OUT/a.log
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES -d $OUT/b -- python $GRAFT_REPO_ROOT/tools/pmc_driver.py > $OUT/b.log 2>&1 || tail -5 $OUT/b.log
cd $GRAFT_REPO_ROOT
python - <<PY
import glob, sqlite3
for run in ("a", "b"):
    for db in glob.glob("$OUT/%s/*/*_results.db" % run):
        cur = sqlite3.connect(db).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
        namecol = "kernel_name" if "kernel_name" in cols else "name"
        valcol = "value" if "value" in cols else "counter_value"
        rows = cur.execute("select %s, counter_name, avg(%s), count(*) from counters_collection group by %s, counter_name" % (namecol, valcol, namecol)).fetchall()
        for name, c, v, n in sorted(rows):
            if "k_policy" in name or "k_world" in name:
                print("%-40s %-32s %16.0f  n=%d" % (name[26:66], c, v, n))
PY
