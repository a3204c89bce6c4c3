"""Workload run under rocprofv3 --pmc: a calibration copy of known size, then a few bench steps."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = argparse.Namespace(worlds=int(os.environ.get("RL_WORLDS", "256")), workload="c4", seed=1)
x = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda:0").normal_()   # 512 MiB > L3
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)            # known traffic: 512 MiB read + 512 MiB written per call
torch.cuda.synchronize()
dw = bench.make_worlds(args, 0, "cuda:0")
for _ in range(40):
    bench.one_step(dw)
torch.cuda.synchronize()
print("agent_steps_per_tick", float(dw.acted_total.item()) / 40)
