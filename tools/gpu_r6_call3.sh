#!/bin/bash
# round 6, third GPU visit: RL_SEAM_OPEN v2 A/B, fence A/B digests, unit costs
TAG=${1:-r6c}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 900 python tools/run_ab.py reinlife_amd/lib/libreinlife_hip_seam0.so reinlife_amd/lib/libreinlife_hip.so 5 > gpurun_out/$TAG/ab_c4.txt 2>&1; cat gpurun_out/$TAG/ab_c4.txt
RL_AB_WORKLOAD=c5 timeout 900 python tools/run_ab.py reinlife_amd/lib/libreinlife_hip_seam0.so reinlife_amd/lib/libreinlife_hip.so 5 > gpurun_out/$TAG/ab_c5.txt 2>&1; cat gpurun_out/$TAG/ab_c5.txt
timeout 1200 python tools/fence_ab.py > gpurun_out/$TAG/fence_ab.txt 2>&1; echo "fence_ab rc=$?"; cat gpurun_out/$TAG/fence_ab.txt | grep -v amdgpu.ids
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/unit_costs tools/ubench/unit_costs.hip 2>/dev/null && /tmp/unit_costs > gpurun_out/$TAG/unit_costs.txt; cat gpurun_out/$TAG/unit_costs.txt
rocprofv3 -L 2>/dev/null | grep -i -E "barrier|SQ_INSTS_" | head -40 > gpurun_out/$TAG/counters_avail.txt; head -40 gpurun_out/$TAG/counters_avail.txt
