# A/B of the input-layer knowledge variants: bash tools/xf_ab.sh "base f3 p3" "c4 c5"   (GPU box; libraries lib/libreinlife_hip_<tag>.so)
cd $GRAFT_REPO_ROOT
L=reinlife_amd/lib
for wl in ${2:-c4 c5}; do
  echo "== workload $wl"
  for r in 1 2 3; do
    for v in ${1:-base}; do
      RL_AB_WORKLOAD=$wl python tools/run_ab.py $L/libreinlife_hip_$v.so $L/libreinlife_hip_$v.so 1 2>&1 | head -1
    done
  done
done
