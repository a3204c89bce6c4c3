#!/usr/bin/env python
"""Shader-clock stamps of the policy half of the multi-tick launch (rl_run, one-wave tile), workgroup `world`, wave 0, last tick of a launch
(tuning; GPU; prof build)."""
import ctypes as C, os, sys
os.environ["RL_PHASE_PROFILE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reinlife_amd import _lib
NAMES = ["entry -> lists built, barrier, tile known", "tile entry", "observation row: 20 reads, max, scale, split", "input layer (120 MFMA)",
         "epilogue + row max + split (-> B2), next ring", "hidden adv (96 MFMA)", "epilogue + split (-> B3)", "head adv (24 MFMA)",
         "hidden val (96 MFMA)", "epilogue + split + head val", "dueling, argmax, stores", "barrier (the other tile waves)"]
IDX = [100, 110, 101, 102, 103, 104, 105, 106, 107, 108, 109, 111, 112]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args = __import__("argparse").Namespace(worlds=R, workload="c4", seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
stamps = torch.zeros(128, dtype=torch.int64, device="cuda:0")
lib = _lib.lib()
dw.run(50, 70, 100)
acc = []
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
m = np.mean(acc, axis=0)
print("rl_run policy half, wave 0 of the sampled world, mean of %d launches, total %.0f cycles" % (len(acc), m.sum()))
for n, v in zip(NAMES, m):
    print("   %-52s %8.0f  %5.1f%%" % (n, v, 100 * v / m.sum()))
