#!/bin/bash
# Everything profiles/ keeps for a final kernel (GPU box):   bash tools/final_round.sh <tag>
TAG=${1:-final}
cd $GRAFT_REPO_ROOT
bash tools/gpu_round.sh $TAG
mkdir -p gpurun_out/$TAG/w
for w in 512 768 1024 2048 4096; do timeout 400 python bench.py --worlds $w --no-cpu-baseline --no-api-trainer > gpurun_out/$TAG/w/bench_${w}worlds.json 2>/dev/null; done
