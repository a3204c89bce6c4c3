#!/usr/bin/env python
"""kKindAll kernel (c5: PPO + PERD3QN): per-wave arrival at the barrier behind the policy tiles and the length of the policy half, last
tick of a launch (prof build; tuning; GPU)."""
import ctypes as C, os, sys
os.environ["RL_PHASE_PROFILE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reinlife_amd import _lib
args = __import__("argparse").Namespace(worlds=256, workload=os.environ.get("RL_AB_WORKLOAD", "c5"), seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
stamps = torch.zeros(128, dtype=torch.int64, device="cuda:0")
lib = _lib.lib()
dw.run(300, 70, 100)
tot, wv, brains = [], [], []
for t in range(48):
    w = (37 * t + 5) % 256
    _lib.check(lib.rl_bind_phase_profile(dw.handle, C.c_void_p(stamps.data_ptr()), w), "bind")
    stamps.zero_()
    dw.run(20, 70, 100)
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy().astype(np.float64)
    if raw[100] and raw[112] > raw[100] and raw[116:124].all():
        tot.append(raw[112] - raw[100]); wv.append(raw[116:124] - raw[100])
        n = int(dw.s["n_agents"][w].item())
        br = dw.s["a_brain"][w, :n].cpu().numpy()
        brains.append((int((br == 0).sum()), int((br == 1).sum())))
print("policy half: mean %.0f cycles (min %.0f, max %.0f) over %d samples" % (np.mean(tot), np.min(tot), np.max(tot), len(tot)))
print("per-wave arrival behind the tiles (mean):", np.mean(wv, axis=0).round(0))
print("agents per brain (PPO, PERD3QN) of the sampled worlds AFTER the launch:", brains[:12])
