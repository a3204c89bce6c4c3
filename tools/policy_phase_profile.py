#!/usr/bin/env python
"""Shader-clock stamps of one policy workgroup (wave 0) per launch (tuning aid; GPU only; uses libreinlife_hip_prof.so).

    python tools/policy_phase_profile.py [--worlds 256]
"""
import argparse
import ctypes as C
import os
import sys

os.environ["RL_PHASE_PROFILE"] = "1"

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reinlife_amd import _lib  # noqa: E402

NAMES = ["entry -> rows known (counts, row list)", "observation rows staged (HBM reads, scale, split, barrier)",
         "input layer (30 MFMA)", "relu + row maxima + publish + 2 barriers", "hidden adv (24 MFMA)", "epilogue + head adv (6 MFMA)",
         "hidden val (24 MFMA)", "epilogue + head val (6 MFMA)", "partials + barrier",
         "epilogue (wave 0: dueling, argmax, Philox, store)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, default=256)
    a = ap.parse_args()
    args = argparse.Namespace(worlds=a.worlds, workload="c4", seed=1)
    dw = bench.make_worlds(args, 0, "cuda:0")
    stamps = torch.zeros(128, dtype=torch.int64, device="cuda:0")
    lib = _lib.lib()
    for blocks in (0, 150, 330):
        acc, ends = [], []
        for t in range(40):
            _lib.check(lib.rl_bind_phase_profile(dw.handle, C.c_void_p(stamps.data_ptr()), blocks), "bind")
            stamps.zero_()
            dw.act()
            dw.tick_refill(70, 100)
            torch.cuda.synchronize()
            raw = stamps.cpu().numpy()
            st = np.concatenate([raw[100:102], raw[110:111], raw[102:110]])  # entry, rows known, staged (mark 10), marks 2..9
            if t >= 10 and st.all():
                acc.append(np.diff(st))
        m = np.mean(acc, axis=0)
        print("policy workgroup %d (wave 0), mean of %d launches, total %.0f cycles" % (blocks, len(acc), m.sum()))
        for n, v in zip(NAMES, m):
            print("   %-52s %8.0f  %5.1f%%" % (n, v, 100 * v / m.sum()))


if __name__ == "__main__":
    main()
