#!/bin/bash
# Vector-memory / L2 counters of a dense policy launch (tuning aid; run on the GPU box through gpurun):  bash tools/pmc_policy_dense.sh [variant]
set -e
export RL_POLICY_VARIANT=${1:-pair}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_dense_$RL_POLICY_VARIANT
rm -rf $OUT && mkdir -p $OUT
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
pass() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -- python $GRAFT_REPO_ROOT/tools/policy_ab.py > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; }; }
pass a TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pass b TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass c SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES
pass g GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python - <<PY
import glob, sqlite3
for db in sorted(glob.glob("$OUT/*/*/*_results.db")):
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
    if "counters_collection" not in tabs: continue
    cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"; valcol = "value" if "value" in cols else "counter_value"
    # the LAST dispatches are the largest launches (10880 tiles): take the max per counter
    for name, c, mx, n in cur.execute("select %s, counter_name, max(%s), count(*) from counters_collection group by %s, counter_name" % (namecol, valcol, namecol)):
        if "k_policy" in name: print("%-36s %-34s max %18.0f  n=%d" % (name[26:62], c, mx, n))
PY
