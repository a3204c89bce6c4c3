#!/usr/bin/env python
"""Shader-clock stamps of the policy half of the multi-tick launch (rl_run), workgroup `world`, wave 0 (tuning; GPU; prof build)."""
import ctypes as C, os, sys
os.environ["RL_PHASE_PROFILE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reinlife_amd import _lib
NAMES = ["entry -> row lists built, tile known", "observation rows staged", "input layer (30 MFMA)", "relu + row maxima + publish + 2 barriers",
         "hidden adv (24 MFMA)", "epilogue + head adv", "hidden val (24 MFMA)", "epilogue + head val", "partials + barrier", "epilogue (wave 0)"]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args = __import__("argparse").Namespace(worlds=R, workload="c4", seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
stamps = torch.zeros(128, dtype=torch.int64, device="cuda:0")
lib = _lib.lib()
dw.run(50, 70, 100)
acc = []
fine = []
for t in range(30):
    _lib.check(lib.rl_bind_phase_profile(dw.handle, C.c_void_p(stamps.data_ptr()), (7 * t) % R), "bind")
    stamps.zero_()
    dw.run(int(os.environ.get('RL_PROFILE_TICKS', '20')), 70, 100)   # the stamps of the LAST tick remain
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy()
    st = np.concatenate([raw[100:102], raw[110:111], raw[102:110]])
    if st.all():
        acc.append(np.diff(st))
        fine.append([raw[120] - raw[100], raw[121] - raw[120], raw[101] - raw[121], raw[122] - raw[101], raw[123] - raw[122], raw[110] - raw[123]])
m = np.mean(acc, axis=0)
print("rl_run policy half, wave 0 of the sampled world, mean of %d launches, total %.0f cycles" % (len(acc), m.sum()))
for n, v in zip(NAMES, m):
    print("   %-52s %8.0f  %5.1f%%" % (n, v, 100 * v / m.sum()))
print("   finer: lists built by wave 0 %d | barrier %d | tile lookup + TileIO + spills %d | call -> tile entry %d | ring start + stage_x (own rows) %d | barrier after staging %d"
      % tuple(np.mean(fine, axis=0)))
