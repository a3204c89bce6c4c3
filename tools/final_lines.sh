cd $GRAFT_REPO_ROOT
bash tools/bench_lines.sh r04c_lines
{ echo "# tools/soak_parity.py on the kernel sources $(python -c 'from reinlife_amd import build; print(build.source_hash())') (round 4, final): policy-driven worlds with refills against the oracle fed the launch's actions"
  python tools/soak_parity.py 256 1500 static fused; python tools/soak_parity.py 64 1500 static; python tools/soak_parity.py 48 1000 nonstatic fused PPO,PERD3QN; python tools/soak_parity.py 32 800 static fused DQN,PPO,D3QN train; python tools/soak_parity.py 48 1000 nonstatic fused PPO,PERD3QN train
  echo "# the same with RL_WORLD_BLOCK=1024 (k_run<1024>, four waves per tile)"; RL_WORLD_BLOCK=1024 python tools/soak_parity.py 128 1000 static fused; RL_WORLD_BLOCK=1024 python tools/soak_parity.py 48 800 static fused PERD3QN,D3QN train; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_soak_parity.txt
{ echo "# tools/fuzz_parity.py 500 2028 on the kernel sources $(python -c 'from reinlife_amd import build; print(build.source_hash())') (round 4, final): last lines"; python tools/fuzz_parity.py 500 2028 2>&1 | grep -v amdgpu.ids | tail -4
  echo "# RL_WORLD_BLOCK=1024 python tools/fuzz_parity.py 200 2029: last lines"; RL_WORLD_BLOCK=1024 python tools/fuzz_parity.py 200 2029 2>&1 | grep -v amdgpu.ids | tail -2; } > gpurun_out/r04_fuzz_parity.txt
RL_WORLD_BLOCK=1024 timeout 300 python bench.py --no-cpu-baseline --no-api-trainer --no-c5 > gpurun_out/r04c_lines/bench_block1024.json 2>/dev/null
tail -3 gpurun_out/r04_soak_parity.txt; tail -3 gpurun_out/r04_fuzz_parity.txt
