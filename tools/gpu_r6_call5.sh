#!/bin/bash
# round 6, fifth GPU visit: RL_XM_LIKELY A/B on configs[4] (and configs[3]: unchanged kernel, control), fence A/B, unit costs
TAG=${1:-r6e}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
L=reinlife_amd/lib
RL_AB_WORKLOAD=c5 timeout 900 python tools/run_ab.py $L/libreinlife_hip.so $L/libreinlife_hip_lk.so 5 > gpurun_out/$TAG/ab_c5_lk.txt 2>&1; cat gpurun_out/$TAG/ab_c5_lk.txt
RL_AB_WORKLOAD=c5 timeout 900 python tools/run_ab.py $L/libreinlife_hip_f3.so $L/libreinlife_hip_lk.so 3 > gpurun_out/$TAG/ab_c5_f3_lk.txt 2>&1; cat gpurun_out/$TAG/ab_c5_f3_lk.txt
timeout 1500 python tools/fence_ab.py > gpurun_out/$TAG/fence_ab.txt 2>&1; echo "fence_ab rc=$?"; grep -v amdgpu.ids gpurun_out/$TAG/fence_ab.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/unit_costs tools/ubench/unit_costs.hip 2>/dev/null && /tmp/unit_costs > gpurun_out/$TAG/unit_costs.txt; cat gpurun_out/$TAG/unit_costs.txt
