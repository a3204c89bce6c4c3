#!/bin/bash
# One GPU visit: tests, bench line, rocprofv3 kernel stats, optional SQ counters.   bash tools/gpu_round.sh <tag> [notest] [nopmc]
TAG=${1:-x}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
if [[ "$*" != *notest* ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/$TAG/pytest.log
fi
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/$TAG/bench.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_driver_window.json 2>> gpurun_out/$TAG/bench.err; echo
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --no-api-trainer > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof.log 2>&1)
DB=$(ls -t gpurun_out/$TAG/prof/*/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/$TAG/kernel_stats.txt && head -8 gpurun_out/$TAG/kernel_stats.txt
timeout 300 python bench.py --path two-launch --no-cpu-baseline > gpurun_out/$TAG/bench_two_launch.json 2>> gpurun_out/$TAG/bench.err
timeout 300 python bench.py --workload c5 --no-cpu-baseline > gpurun_out/$TAG/bench_c5.json 2>> gpurun_out/$TAG/bench.err
for f in bench_two_launch bench_c5; do python -c "import json,sys; d=json.loads(open('gpurun_out/$TAG/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['config']['loop'])"; done
if [[ "$*" != *nopmc* ]]; then
  timeout 900 bash tools/pmc_sq.sh 256 $TAG > gpurun_out/$TAG/pmc_sq.txt 2>&1; tail -60 gpurun_out/$TAG/pmc_sq.txt
  timeout 600 bash tools/pmc_traffic.sh > gpurun_out/$TAG/pmc_traffic.txt 2>&1; tail -8 gpurun_out/$TAG/pmc_traffic.txt; cp gpurun_out/pmc/tick_traffic.json gpurun_out/$TAG/traffic.json
fi
