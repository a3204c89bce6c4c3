#!/bin/bash
# round 6, sixth GPU visit: product (seam open, likely hint, hoist-spill fixes) vs the round-5 kernel sources, then the suite
TAG=${1:-r6f}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
L=reinlife_amd/lib
timeout 900 python tools/run_ab.py $L/libreinlife_hip_r05.so $L/libreinlife_hip.so 5 > gpurun_out/$TAG/ab_c4.txt 2>&1; cat gpurun_out/$TAG/ab_c4.txt
RL_AB_WORKLOAD=c5 timeout 900 python tools/run_ab.py $L/libreinlife_hip_r05.so $L/libreinlife_hip.so 5 > gpurun_out/$TAG/ab_c5.txt 2>&1; cat gpurun_out/$TAG/ab_c5.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/$TAG/pytest.log
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_driver_window.json 2>> gpurun_out/$TAG/bench.err
python - <<PY
import json
for f in ("bench","bench_driver_window"):
    d=json.loads(open("gpurun_out/$TAG/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "tick_us", d["roofline"].get("avg_tick_us"), "api", d["api_trainer"]["value"], d["api_trainer"]["us_per_tick"], d["api_trainer"]["value_at_steps"], "c5", d["c5"]["value"], d["c5"]["kernel_us_per_tick"], "two", d.get("two_launch_step_us"))
PY
