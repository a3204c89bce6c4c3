"""Workload run under rocprofv3 --pmc: a calibration copy of known size, then a few bench steps."""
import argparse
import os
if os.environ.get("RL_PMC_RUN_MASK", "0") != "0":
    os.environ.setdefault("RL_TUNE", "1")   # counters of one half alone: the tuning library carries rl_debug_set_run_mask (the product does not)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = argparse.Namespace(worlds=int(os.environ.get("RL_WORLDS", "256")), workload=os.environ.get("RL_PMC_WORKLOAD", "c4"), seed=1)
x = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda:0").normal_()   # 512 MiB > L3
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)            # known traffic: 512 MiB read + 512 MiB written per call
torch.cuda.synchronize()
dw = bench.make_worlds(args, 0, "cuda:0")
fused = os.environ.get("RL_PMC_PATH", "fused") == "fused" and dw.run_supported() and args.worlds <= 768
TICKS, LAUNCHES = 40, 5
if fused:   # the multi-tick launch: LAUNCHES dispatches of TICKS ticks each
    mask = int(os.environ.get("RL_PMC_RUN_MASK", "0"))   # tuning: 1 = skip the policy half, 2 = skip the tick half (counters of one half alone)
    if mask:
        from reinlife_amd import _lib
        dw.run(300, 70, 100)
        _lib.lib().rl_debug_set_run_mask(mask)
    for _ in range(LAUNCHES):
        dw.run(TICKS, 70, 100)
    print("ticks_per_launch", TICKS)
else:
    for _ in range(TICKS * LAUNCHES):
        bench.one_step(dw)
    print("ticks_per_launch", 1)
torch.cuda.synchronize()
print("agent_steps_per_tick", float(dw.acted_total.item()) / (TICKS * LAUNCHES))
