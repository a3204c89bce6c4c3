#!/bin/bash
# round 6, eighth GPU visit: lane_id() from v_mbcnt + fresh lane in the finish: p1 (previous product) vs product, TRAIN 1 and TRAIN 0; then the suite
TAG=${1:-r6h}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
L=reinlife_amd/lib
for wl in c4 c5; do
  echo "== $wl TRAIN 1: p1 vs product"; RL_AB_TRAIN=1 RL_AB_WORKLOAD=$wl timeout 900 python tools/run_ab.py $L/libreinlife_hip_p1.so $L/libreinlife_hip.so 4 2>&1
  echo "== $wl TRAIN 0: p1 vs product"; RL_AB_WORKLOAD=$wl timeout 900 python tools/run_ab.py $L/libreinlife_hip_p1.so $L/libreinlife_hip.so 4 2>&1
done > gpurun_out/$TAG/ab.txt 2>&1; cat gpurun_out/$TAG/ab.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/$TAG/pytest.log
timeout 300 python bench.py --path two-launch --no-cpu-baseline --no-api-trainer --no-c5 --no-single-world > gpurun_out/$TAG/bench_two_launch.json 2> gpurun_out/$TAG/bench.err
python -c "
import json; d=json.loads(open('gpurun_out/$TAG/bench_two_launch.json').read().strip().splitlines()[-1]); print('two-launch', d['value'], d['ms_per_step'], d['roofline_tick']['avg_launch_us'], d['roofline_policy']['avg_launch_us'])"
