#!/usr/bin/env python
"""Where a launch's fixed cost goes: shader-clock stamps of k_run's stages for launches of 1 / 2 / 20 ticks (prof build; tuning; GPU)."""
import ctypes as C, os, sys
os.environ["RL_PHASE_PROFILE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reinlife_amd import _lib
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
stamps = torch.zeros(128, dtype=torch.int64, device="cuda:0")
lib = _lib.lib()
dw.run(300, 70, 100)
NAMES = ["entry -> world loaded, mirror / lists / constants ready", "stagger wait", "policy half of tick 0", "tick half of tick 0", "tick 1 (policy + tick)", "ticks 2 .. n-1", "store_world + drain"]
for n in (1, 2, 20, 100, 500):
    acc, lacc = [], []
    for t in range(48):
        _lib.check(lib.rl_bind_phase_profile(dw.handle, C.c_void_p(stamps.data_ptr()), (37 * t + 5) % 256), "bind")
        stamps.zero_()
        dw.run(n, 70, 100)
        torch.cuda.synchronize()
        st = stamps.cpu().numpy()[70:78].astype(np.float64)
        if n == 1:
            st[5] = st[4]
        acc.append(np.diff(st))
        raw = stamps.cpu().numpy().astype(np.float64)
        if n == 1:
            lacc.append([raw[90] - raw[70], raw[91] - raw[90], raw[92] - raw[91], raw[93] - raw[92], raw[94] - raw[93], raw[95] - raw[94],
                         raw[32] - raw[30], raw[33] - raw[32], raw[91] - raw[33]])
    m = np.mean(acc, axis=0); mx = np.max(acc, axis=0)
    if n == 1:
        ld = np.mean(lacc, axis=0)
        print("   inside the load: kernel entry -> params/carve %.0f | load_world (HBM trip, LDS init, 2 barriers) %.0f [of it: loads issued -> LDS init done %.0f, consume %.0f, barrier+occ %.0f] | mirror preload %.0f | lists (wave 0) %.0f | constants %.0f | barrier %.0f" % (
              ld[0], ld[1], ld[6], ld[7], ld[8], ld[2], ld[3], ld[4], ld[5]))
    print("launch of %d tick(s): %d samples (different worlds), mean total %.0f counts" % (n, len(acc), m.sum()))
    if n > 2:
        print("   ticks 2 .. n-1: %.0f counts per tick" % (m[5] / (n - 2)))
    for name, v, x in zip(NAMES, m, mx):
        print("   %-60s mean %8.0f   max %8.0f" % (name, v, x))
