#!/bin/bash
# The four bench lines kept under profiles/ (GPU box):   bash tools/bench_lines.sh <tag>
TAG=${1:-lines}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_driver_window.json 2>> gpurun_out/$TAG/bench.err
timeout 300 python bench.py --path two-launch --no-cpu-baseline > gpurun_out/$TAG/bench_two_launch.json 2>> gpurun_out/$TAG/bench.err
timeout 300 python bench.py --workload c5 --no-cpu-baseline > gpurun_out/$TAG/bench_c5.json 2>> gpurun_out/$TAG/bench.err
for f in bench bench_driver_window bench_two_launch bench_c5; do python -c "import json,sys; d=json.loads(open('gpurun_out/$TAG/$f.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), (d.get('api_trainer') or {}).get('value'))"; done
