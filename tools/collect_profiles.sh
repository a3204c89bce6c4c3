#!/bin/bash
# Copy what tools/final_round.sh <tag> left under gpurun_out/ into profiles/ (build container):   [R=r04] bash tools/collect_profiles.sh <tag>
T=$1
H=$(python -c "from reinlife_amd import build; print(build.source_hash())")
cp gpurun_out/$T/pmc_traffic.txt profiles/${R:-r04}_pmc_hbm_traffic.txt
cp gpurun_out/pmc_${T}_256/report.txt profiles/${R:-r04}_run_sq_counters.txt
DB=$(ls -t gpurun_out/$T/prof/*/*_results.db | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-kernel-timing --no-api-trainer  (${R:-r04}, final kernel sources $H)"
  echo "# k_run is launched three times: burn-in (2000 ticks, with the start-up transient of the synthetic worlds), warm-up (300 ticks), TIMED REGION (2000 ticks): see the per-call line below"
  python tools/rocpd_summary.py $DB; } > profiles/${R:-r04}_kernel_stats.txt
python - <<PY
import json
t=json.load(open('gpurun_out/pmc/tick_traffic.json'))
assert t["kernel_src_sha16"] == "$H", (t["kernel_src_sha16"], "$H")
t["source"]="profiles/${R:-r04}_pmc_hbm_traffic.txt (tools/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes, calibrated with a device copy)"
json.dump(t,open('profiles/run_traffic.json','w'),indent=1)
print(t["kernel_src_sha16"], t["hbm_bytes_per_tick"])
PY
for w in 512 768 1024 2048 4096; do cp gpurun_out/$T/w/bench_${w}worlds.json profiles/${R:-r04}_bench_line_${w}worlds.json; done
grep "calls of" profiles/${R:-r04}_kernel_stats.txt | head -1 | cut -c1-200
