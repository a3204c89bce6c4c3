#!/bin/bash
# round 6, seventh GPU visit: the TRAIN 1 instantiations (what trainer() launches): r05 sources vs product vs RL_TIDX_VIA_WAVE
TAG=${1:-r6g}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
L=reinlife_amd/lib
for wl in c4 c5; do
  for pair in "r05 " "tw "; do
    t=${pair% }; echo "== $wl TRAIN 1: $t vs product"
    RL_AB_TRAIN=1 RL_AB_WORKLOAD=$wl timeout 900 python tools/run_ab.py $L/libreinlife_hip_$t.so $L/libreinlife_hip.so 3 2>&1
  done
  echo "== $wl TRAIN 0: tw vs product"; RL_AB_WORKLOAD=$wl timeout 900 python tools/run_ab.py $L/libreinlife_hip_tw.so $L/libreinlife_hip.so 3 2>&1
done > gpurun_out/$TAG/ab_train.txt 2>&1; cat gpurun_out/$TAG/ab_train.txt
