#!/usr/bin/env python
"""Shader-clock stamps inside policy_pair2 / policy_tile1s of the mixed-kind kernel (c5: PPO + PERD3QN), one wave of the sampled world,
last tick of a launch (prof build; tuning; GPU).   python tools/run_pair2_profile.py [wave]"""
import ctypes as C, os, sys
os.environ["RL_PHASE_PROFILE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reinlife_amd import _lib
wave = int(sys.argv[1]) if len(sys.argv) > 1 else 0
args = __import__("argparse").Namespace(worlds=256, workload="c5", seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
stamps = torch.zeros(128, dtype=torch.int64, device="cuda:0")
lib = _lib.lib()
lib.rl_debug_set_ablate.argtypes = [C.c_int]
lib.rl_debug_set_ablate(wave << 20)
dw.run(300, 70, 100)
NAMES = ["policy half entry -> tile entry", "weights requested, row reads, max, scale, first split", "input layer pass 0 (60 MFMA)",
         "input layer pass 1 (60 MFMA), epilogue of pass 0 in its shadow", "exposed epilogue of pass 1 (32 steps), row max", "barrier 1",
         "row scale, split of 8 own chunks -> LDS", "barrier 2", "hidden layer pass 0 (96 MFMA, B from LDS)", "hidden layer pass 1 (96 MFMA), epilogue of pass 0 in its shadow",
         "exposed epilogue of pass 1, row max, scale", "head (24 MFMA)", "-> arrival behind the tiles"]
IDX = [100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 113, 114, 115]
acc = []
for t in range(40):
    w = (37 * t + 5) % 256
    _lib.check(lib.rl_bind_phase_profile(dw.handle, C.c_void_p(stamps.data_ptr()), w), "bind")
    stamps.zero_()
    dw.run(20, 70, 100)
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy().astype(np.float64)
    st = list(raw[IDX]) + [raw[116 + wave]]
    if all(st) and all(b >= a for a, b in zip(st, st[1:])):
        acc.append(np.diff(st))
lib.rl_debug_set_ablate(0)
if not acc:
    print("no complete samples (the stamped wave had no PPO tile?)"); sys.exit(0)
m = np.mean(acc, axis=0)
print("wave %d, %d samples, policy half entry -> arrival: %.0f counts" % (wave, len(acc), m.sum()))
for n, v in zip(NAMES, m):
    print("   %-75s %7.0f  %5.1f%%" % (n, v, 100 * v / m.sum()))
