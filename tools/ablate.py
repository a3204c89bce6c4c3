"""Ablation timing of the tick kernel (tuning aid, GPU only).  Sections are skipped with a runtime bit mask
(rl_debug_set_ablate; results of such launches are WRONG, only their duration is used).  Every timed launch starts from
the same recorded mid-life world states, so all variants see identical inputs."""
import argparse
import os
import sys

os.environ["RL_PHASE_PROFILE"] = "1"  # the tuning library (libreinlife_hip_prof.so) carries the ablation switches

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reinlife_amd import _lib  # noqa: E402

MASKS = [(0, "full"), (1, "-obs writes"), (2, "-planes"), (3, "-obs writes-planes"), (4, "-add_food"), (8, "-repro/births"),
         (16, "-conflict loop"), (64, "-store"), (128, "-emit lists"), (256, "-update phase"), (512, "-step phase"),
         (768, "-step-update"), (771, "-step-update-obs-planes"), (835, "launch + load + orders + outputs"), (65536, "launch + load_world only"), (32768, "empty kernel")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, default=256)
    a = ap.parse_args()
    args = argparse.Namespace(worlds=a.worlds, workload="c4", seed=1)
    dw = bench.make_worlds(args, 0, "cuda:0")
    lib = _lib.lib()
    lib.rl_debug_set_ablate.argtypes = [__import__("ctypes").c_int]
    snaps = []
    for t in range(60):
        bench.one_step(dw)
        if t >= 20 and t % 4 == 0:
            dw.act()
            torch.cuda.synchronize()
            snaps.append(({k: v.clone() for k, v in dw.s.items()}, dw.actions.clone()))
    # keep the GPU busy (no host syncs inside a variant): restore copies + timed tick are enqueued back to back, the
    # event pairs are read afterwards -- isolated launches would run at idle clocks and measure launch latency instead
    for mask, name in MASKS:
        evs = []
        for rep in range(6):
            for st, acts in snaps:
                for k, v in st.items():
                    dw.s[k].copy_(v)
                dw.actions.copy_(acts)
                lib.rl_debug_set_ablate(mask)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); dw.tick_refill(70, 100); e1.record()
                lib.rl_debug_set_ablate(0)
                evs.append((e0, e1))
        torch.cuda.synchronize()
        times = [a.elapsed_time(b) * 1e3 for a, b in evs][len(snaps):]
        print("%-36s median %6.2f us   min %6.2f" % (name, float(np.median(times)), min(times)))


if __name__ == "__main__":
    main()
