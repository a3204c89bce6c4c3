"""Tick duration with and without the fused refill (tuning aid, GPU only): every tick has a few refilling worlds, and a
launch lasts as long as its slowest world."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = argparse.Namespace(worlds=256, workload="c4", seed=1)
for thr in (70, 0, 70, 0):
    dw = bench.make_worlds(args, 0, "cuda:0")
    for _ in range(40):
        dw.act(); dw.tick_refill(70, 100)
    evs = []
    for _ in range(60):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); dw.act(); e[1].record(); dw.tick_refill(thr, 100); e[2].record(); evs.append(e)
    torch.cuda.synchronize()
    print("refill threshold %3d: policy %.2f us, tick %.2f us, refills so far %d, mean agents %.1f" % (
        thr, np.median([a.elapsed_time(b) for a, b, _ in evs]) * 1e3, np.median([b.elapsed_time(c) for _, b, c in evs]) * 1e3,
        int(dw.refill_count.item()), float(dw.s["n_agents"].float().mean().item())))
