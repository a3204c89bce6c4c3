#!/usr/bin/env python
"""Shader-clock stamps of the four-wave policy tile (policy_quad) in k_run<1024>: wave 0 of the sampled world (tile 0, wave q = 0), last tick
of a launch (tuning; GPU; prof build)."""
import ctypes as C, os, sys
os.environ["RL_PHASE_PROFILE"] = "1"
os.environ["RL_WORLD_BLOCK"] = "1024"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reinlife_amd import _lib
NAMES = ["policy half entry -> tile known", "tile entry -> own chunks read, partial maximum", "barrier 1 (row maximum)", "input layer (30 MFMA)",
         "epilogue, barrier 2 (row maximum)", "split, barrier 3", "hidden layers (48 MFMA)", "epilogues, barrier 4", "splits, barrier 5", "advantage head (24 MFMA)",
         "barrier behind the tiles", "finish + last barrier"]
IDX = [100, 110, 101, 102, 103, 104, 105, 106, 107, 108, 109, 111, 112]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args = __import__("argparse").Namespace(worlds=R, workload="c4", seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
stamps = torch.zeros(128, dtype=torch.int64, device="cuda:0")
lib = _lib.lib()
dw.run(50, 70, 100)
acc = []
arr = []
for t in range(30):
    _lib.check(lib.rl_bind_phase_profile(dw.handle, C.c_void_p(stamps.data_ptr()), (7 * t) % R), "bind")
    stamps.zero_()
    dw.run(20, 70, 100)
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy()
    st = raw[IDX]
    if t == 0: print("raw", raw[98:114])
    if st.all():
        acc.append(np.diff(st))
        arr.append(raw[116:128] - raw[110])
m = np.mean(acc, axis=0)
print("k_run<1024> policy half, wave 0 of the sampled world, mean of %d launches, total %.0f counts" % (len(acc), m.sum()))
for n, v in zip(NAMES, m):
    print("   %-52s %8.0f  %5.1f%%" % (n, v, 100 * v / m.sum()))
print("tile 0's waves q = 0 .. 3, counts from tile entry: before barrier 1", np.mean(arr, axis=0)[0:4].round(0), "after the input layer", np.mean(arr, axis=0)[4:8].round(0),
      "after the hidden layers", np.mean(arr, axis=0)[8:12].round(0))
