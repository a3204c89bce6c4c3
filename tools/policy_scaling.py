"""Policy kernel duration vs number of 32-row tiles (tuning aid, GPU only): separates the per-workgroup dependency chain
(intercept) from MFMA contention per extra tile on a CU (slope)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reinlife_amd import _lib  # noqa: E402
from reinlife_amd.worlds import pack_brain_weights, policy_forward  # noqa: E402

for name in ("PERD3QN", "DQN", "PPO"):
    kind = _lib.KIND_BY_METHOD[name]
    packed = pack_brain_weights(kind, bench.brain_weights(name, 1))
    for tiles in (64, 256, 512, 768, 1024, 2048, 4096, 16384):
        n = tiles * 32
        obs = torch.randn(n + 1, 153, device="cuda:0")[:n]
        out = torch.empty(n, 8, device="cuda:0")
        for _ in range(5):
            policy_forward(kind, packed, obs, out)
        evs = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); policy_forward(kind, packed, obs, out); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        t = float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
        flop = bench.POLICY_FLOP_PER_AGENT[name] * n
        print("%-8s tiles %6d rows %7d : %8.2f us  %6.1f TFLOP/s (%.0f%% of fp32 MFMA peak)" % (name, tiles, n, t, flop / t / 1e6, 100 * flop / t / 1e6 / 157.3))
