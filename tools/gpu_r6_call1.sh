#!/bin/bash
# round 6, first GPU visit: the suite (incl. the 8-rank dry run), the default line, the driver's window
TAG=${1:-r6a}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/$TAG/pytest.log
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/$TAG/bench.json
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench_driver_window.json 2>> gpurun_out/$TAG/bench.err; echo
timeout 600 python bench.py --gpus 2 --dist-backend gloo --share-gpu --worlds 128 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_2rank_dry.json 2>> gpurun_out/$TAG/bench.err; echo "2rank rc=$?"
