#!/usr/bin/env python
"""Shader-clock stamps of the policy half of the multi-tick launch (rl_run, two waves per tile: role 0 of the first tile), workgroup `world`, wave 0, last tick of a launch
(tuning; GPU; prof build)."""
import ctypes as C, os, sys
os.environ["RL_PHASE_PROFILE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reinlife_amd import _lib
NAMES = ["entry -> tile known (lists were built during the previous tick)", "tile entry", "observation row: 20 reads, max, scale, first split", "input layer, own tile pair (60 MFMA) + epilogue, row-max exchange, split, exchange (2 barriers)",
         "hidden layer of the role's branch (96 MFMA), epilogue of tiles 0,1 in its shadow", "epilogue of tiles 2,3, row max, scale", "head (24 MFMA) with the split of its input",
         "barrier (partner wave, other tiles)", "dueling combine, argmax, stores, barrier"]
IDX = [100, 110, 101, 102, 104, 105, 106, 109, 111, 112]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args = __import__("argparse").Namespace(worlds=R, workload=os.environ.get("RL_AB_WORKLOAD", "c4"), seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
stamps = torch.zeros(128, dtype=torch.int64, device="cuda:0")
lib = _lib.lib()
dw.run(50, 70, 100)
acc = []
wv = []
for t in range(30):
    _lib.check(lib.rl_bind_phase_profile(dw.handle, C.c_void_p(stamps.data_ptr()), (7 * t) % R), "bind")
    stamps.zero_()
    dw.run(int(os.environ.get("RL_PROFILE_TICKS", "20")), 70, 100)   # the stamps of the LAST tick remain
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy()
    st = raw[IDX]
    if t == 0: print("raw", raw[98:114])
    if st.all():
        acc.append(np.diff(st))
        if raw[116:124].all(): wv.append(raw[116:124] - raw[100])
m = np.mean(acc, axis=0)
print("rl_run policy half, wave 0 of the sampled world, mean of %d launches, total %.0f cycles" % (len(acc), m.sum()))
for n, v in zip(NAMES, m):
    print("   %-52s %8.0f  %5.1f%%" % (n, v, 100 * v / m.sum()))
if wv:
    print("cycles from entry to each wave's arrival at the barrier behind the tiles (waves 0-3: advantage role of tiles 0-3, waves 4-7: value role):")
    print("   ", np.mean(wv, axis=0).round(0))
