#!/bin/bash
# round 6, second GPU visit: RL_SEAM_OPEN A/B (product = open, seam0 = closed), parity of the open seam
TAG=${1:-r6b}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 900 python tools/run_ab.py reinlife_amd/lib/libreinlife_hip_seam0.so reinlife_amd/lib/libreinlife_hip.so 5 > gpurun_out/$TAG/ab_c4.txt 2>&1; cat gpurun_out/$TAG/ab_c4.txt
RL_AB_WORKLOAD=c5 timeout 900 python tools/run_ab.py reinlife_amd/lib/libreinlife_hip_seam0.so reinlife_amd/lib/libreinlife_hip.so 5 > gpurun_out/$TAG/ab_c5.txt 2>&1; cat gpurun_out/$TAG/ab_c5.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/$TAG/pytest.log
