#!/bin/bash
# round 6, fourth GPU visit: configs[4] attribution (f3 / cold-call A/B, tile stamps in prof builds), c4 / c5 stamps for the latency model, fence A/B, unit costs
TAG=${1:-r6d}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
L=reinlife_amd/lib
RL_AB_WORKLOAD=c5 timeout 900 python tools/run_ab.py $L/libreinlife_hip.so $L/libreinlife_hip_f3.so 4 > gpurun_out/$TAG/ab_c5_f3.txt 2>&1; cat gpurun_out/$TAG/ab_c5_f3.txt
RL_AB_WORKLOAD=c5 timeout 900 python tools/run_ab.py $L/libreinlife_hip.so $L/libreinlife_hip_cold.so 4 > gpurun_out/$TAG/ab_c5_cold.txt 2>&1; cat gpurun_out/$TAG/ab_c5_cold.txt
for w in 0 1 4 5; do
  echo "== shipped (prof build), wave $w"; REINLIFE_HIP_LIB=$L/libreinlife_hip_prof.so timeout 300 python tools/run_pair2_profile.py $w 2>&1 | grep -v amdgpu.ids
  echo "== f3 forced (prof build + RL_XM_ASSUME_ALWAYS), wave $w"; REINLIFE_HIP_LIB=$L/libreinlife_hip_proff3.so timeout 300 python tools/run_pair2_profile.py $w 2>&1 | grep -v amdgpu.ids
done > gpurun_out/$TAG/c5_tile_stamps.txt 2>&1; tail -45 gpurun_out/$TAG/c5_tile_stamps.txt
{ echo "== c4 tick half"; REINLIFE_HIP_LIB=$L/libreinlife_hip_prof.so timeout 300 python tools/run_tick_profile.py; echo "== c4 policy half"; REINLIFE_HIP_LIB=$L/libreinlife_hip_prof.so timeout 300 python tools/run_phase_profile.py
  echo "== c5 tick half"; RL_AB_WORKLOAD=c5 REINLIFE_HIP_LIB=$L/libreinlife_hip_prof.so timeout 300 python tools/run_tick_profile.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/stamps.txt; cat gpurun_out/$TAG/stamps.txt
timeout 1500 python tools/fence_ab.py > gpurun_out/$TAG/fence_ab.txt 2>&1; echo "fence_ab rc=$?"; grep -v amdgpu.ids gpurun_out/$TAG/fence_ab.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/unit_costs tools/ubench/unit_costs.hip 2>/dev/null && /tmp/unit_costs > gpurun_out/$TAG/unit_costs.txt; cat gpurun_out/$TAG/unit_costs.txt
