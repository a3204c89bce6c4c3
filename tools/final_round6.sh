#!/bin/bash
# Everything profiles/ keeps for the round-6 kernel sources (GPU box; ~30 min):   bash tools/final_round6.sh <tag>
# Order matters: the PMC traffic of BOTH paths is taken first and put where bench.py looks for it (profiles/run_traffic.json, profiles/tick_traffic.json,
# stamped with the kernel sources' hash), the stamps + unit costs give profiles/latency_model.json, THEN the bench lines are taken -- so that
# `traffic`, `roofline_tick.traffic`, `roofline_policy.traffic` and `latency_bound_us` in them are this run's own.
TAG=${1:-r6final}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O/w
L=reinlife_amd/lib
H=$(python -c "from reinlife_amd import build; print(build.source_hash())")
echo "kernel sources $H"
# ---- 1. the suite
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
# ---- 2. HBM traffic by PMC counters, the multi-tick launch and the two stand-alone kernels (separate FETCH_SIZE / WRITE_SIZE passes)
timeout 900 bash tools/pmc_traffic.sh > $O/pmc_traffic.txt 2>&1; tail -4 $O/pmc_traffic.txt; cp gpurun_out/pmc/tick_traffic.json $O/run_traffic.json
RL_PMC_PATH=two-launch timeout 900 bash tools/pmc_traffic.sh > $O/pmc_traffic_two_launch.txt 2>&1; tail -4 $O/pmc_traffic_two_launch.txt; cp gpurun_out/pmc/tick_traffic.json $O/tick_traffic.json
python - <<PY
import json
for src, dst, note in (("$O/run_traffic.json", "profiles/run_traffic.json", "r06_pmc_hbm_traffic.txt"), ("$O/tick_traffic.json", "profiles/tick_traffic.json", "r06_pmc_hbm_traffic_two_launch.txt")):
    t = json.load(open(src)); assert t["kernel_src_sha16"] == "$H"
    t["source"] = "profiles/%s (tools/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes, calibrated with a device copy)" % note
    json.dump(t, open(dst, "w"), indent=1); json.dump(t, open(src, "w"), indent=1)
PY
# ---- 3. SQ counters: the multi-tick launch (configs[3], configs[4]) and the two stand-alone kernels
timeout 900 bash tools/pmc_sq.sh 256 ${TAG}_run > $O/sq_run.log 2>&1; cp gpurun_out/pmc_${TAG}_run_256/report.txt $O/run_sq_counters.txt
RL_PMC_WORKLOAD=c5 timeout 900 bash tools/pmc_sq.sh 256 ${TAG}_c5 > $O/sq_c5.log 2>&1; cp gpurun_out/pmc_${TAG}_c5_256/report.txt $O/run_sq_counters_c5.txt
RL_PMC_PATH=two-launch timeout 900 bash tools/pmc_sq.sh 256 ${TAG}_two > $O/sq_two.log 2>&1; cp gpurun_out/pmc_${TAG}_two_256/report.txt $O/two_launch_sq_counters.txt
grep -A3 "^== " $O/run_sq_counters.txt | head -8
# ---- 4. unit costs + stamps -> the latency-bound model
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/unit_costs tools/ubench/unit_costs.hip 2>/dev/null && /tmp/unit_costs > $O/ubench.txt; cat $O/ubench.txt
{ echo "# shader-clock stamps of the stamped build (RL_PHASE_PROFILE=1 python reinlife_amd/build.py) on kernel sources $H: tools/run_tick_profile.py, tools/run_phase_profile.py, tools/run_pair2_profile.py"
  echo "== c4 tick half"; REINLIFE_HIP_LIB=$L/libreinlife_hip_prof.so timeout 300 python tools/run_tick_profile.py
  echo "== c4 policy half"; REINLIFE_HIP_LIB=$L/libreinlife_hip_prof.so timeout 300 python tools/run_phase_profile.py
  echo "== c5 tick half"; RL_AB_WORKLOAD=c5 REINLIFE_HIP_LIB=$L/libreinlife_hip_prof.so timeout 300 python tools/run_tick_profile.py
  echo "== c5 policy half"; for w in 0 1 2 3; do REINLIFE_HIP_LIB=$L/libreinlife_hip_prof.so timeout 300 python tools/run_pair2_profile.py $w; done; } 2>&1 | grep -v amdgpu.ids > $O/stamps.txt
python tools/latency_bound.py $O/ubench.txt $O/stamps.txt $O/latency_model.json && python - <<PY
import json
m = json.load(open("$O/latency_model.json")); m["inputs"] = {"ubench": "profiles/r06_ubench.txt", "stamps": "profiles/r06_stamps.txt"}
json.dump(m, open("profiles/latency_model.json", "w"), indent=1); json.dump(m, open("$O/latency_model.json", "w"), indent=1)
PY
# ---- 5. the bench lines (traffic + latency model of THIS run in them), rocprofv3 kernel stats of the default command
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_window.json 2>> $O/bench.err
timeout 300 python bench.py --path two-launch --no-cpu-baseline > $O/bench_two_launch.json 2>> $O/bench.err
timeout 300 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err
timeout 900 python bench.py --gpus 2 --dist-backend gloo --share-gpu --worlds 128 --steps 20 --warmup 5 > $O/bench_2rank_dry.json 2>> $O/bench.err
for w in 512 768 1024 2048 4096; do timeout 400 python bench.py --worlds $w --no-cpu-baseline --no-api-trainer --no-single-world > $O/w/bench_${w}worlds.json 2>/dev/null; done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --no-api-trainer --no-single-world > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
DB=$(ls -t $O/prof/*/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --no-api-trainer --no-single-world  (r06, kernel sources $H)"
  echo "# k_run<512, true, 2, 0> is launched three times: burn-in (2000 ticks), warm-up (300 ticks), TIMED REGION (2000 ticks): see the per-call line; k_run<512, true, 4, 0> = the c5 leg (1000 + 1000 ticks)"
  python tools/rocpd_summary.py $DB; } > $O/kernel_stats.txt; head -12 $O/kernel_stats.txt
rm -rf $O/prof
python - <<PY
import json
for f in ("bench", "bench_driver_window", "bench_two_launch", "bench_c5", "bench_2rank_dry"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r.get("traffic"), "lat", r.get("latency_bound_us"), r.get("frac_of_latency_bound"),
              "tickT", (d.get("roofline_tick") or {}).get("traffic"), "polT", (d.get("roofline_policy") or {}).get("traffic"), "api", (d.get("api_trainer") or {}).get("value"), "c5", (d.get("c5") or {}).get("value"))
    except Exception as e: print(f, "unreadable", e)
PY
# ---- 6. race evidence: fence A/B digests, soak + fuzz against the oracle on the product AND on the full-fence build
timeout 1500 python tools/fence_ab.py 2>&1 | grep -v amdgpu.ids > $O/fence_ab.txt; tail -3 $O/fence_ab.txt
for lib in "" "$L/libreinlife_hip_fence.so"; do
  { echo "# library: ${lib:-$L/libreinlife_hip.so (product)}  kernel sources $H"
    for a in "256 600 static fused" "48 500 nonstatic fused PPO,PERD3QN" "32 400 static fused DQN,PPO,D3QN train" "64 400 static"; do REINLIFE_HIP_LIB=$lib timeout 600 python tools/soak_parity.py $a 2>&1 | grep -v amdgpu.ids | tail -1; done
    REINLIFE_HIP_LIB=$lib timeout 900 python tools/fuzz_parity.py 120 2031 2>&1 | grep -v amdgpu.ids | tail -2; } >> $O/full_fence_soak.txt
done; cat $O/full_fence_soak.txt
# ---- 7. against the round-5 kernel sources, same box
{ echo "# tools/run_ab.py: lib built from commit 3ed9574 (round 5) vs the product (kernel sources $H), alternating 2000-tick launches at 256 worlds"
  echo "== configs[3], TRAIN 0"; timeout 600 python tools/run_ab.py $L/libreinlife_hip_r05.so $L/libreinlife_hip.so 4
  echo "== configs[4], TRAIN 0"; RL_AB_WORKLOAD=c5 timeout 600 python tools/run_ab.py $L/libreinlife_hip_r05.so $L/libreinlife_hip.so 4
  echo "== configs[3], TRAIN 1 (Tracker + epsilon schedule: what trainer() launches)"; RL_AB_TRAIN=1 timeout 600 python tools/run_ab.py $L/libreinlife_hip_r05.so $L/libreinlife_hip.so 3
  echo "== configs[4], TRAIN 1"; RL_AB_TRAIN=1 RL_AB_WORKLOAD=c5 timeout 600 python tools/run_ab.py $L/libreinlife_hip_r05.so $L/libreinlife_hip.so 3; } > $O/ab_vs_r05.txt 2>&1; cat $O/ab_vs_r05.txt
# ---- 8. only the summaries travel back (gpurun merges at most 64 MiB): the rocpd databases stay on the box
rm -rf gpurun_out/pmc gpurun_out/pmc_* $O/prof; du -sh gpurun_out | tail -1
