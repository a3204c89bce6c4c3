#!/usr/bin/env python
"""Per-phase shader-clock breakdown of the tick kernel for a few worlds (tuning aid; GPU only).

    python tools/phase_profile.py [--worlds 256] [--ticks 60]
"""
import argparse
import ctypes as C
import os
import sys

os.environ["RL_PHASE_PROFILE"] = "1"  # selects libreinlife_hip_prof.so (built with -DRL_PHASE_PROFILE)

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reinlife_amd import _lib  # noqa: E402

NAMES = {1: "load", 2: "act+attack+prep", 3: "conflict loop", 4: "eat/move/death/hash", 5: "rewards", 6: "food count+bitmap",
         8: "food placement", 9: "order1", 10: "planes1", 11: "obs1 write", 12: "step outputs", 13: "best agents",
         14: "repro gates+parents+bitmap", 15: "births+produce", 17: "remove dead", 18: "order2", 19: "refill (if any)",
         20: "genes+planes2", 21: "obs2 write", 22: "store (before the last observation pass)", 30: "load: all global loads issued", 31: "load: LDS clear done", 32: "load: scalars + grid consumed", 33: "load: agent arrays issued", 34: "load: barrier", 35: "eat+vanish flags (+bar)", 36: "clear old cells (+bar)", 37: "place+death+hash", 38: "elig bitmap+scan", 39: "gates+parents bitmap", 40: "parents compaction", 41: "birth placements", 42: "produce", 100: "  (of load: kernarg + n_agents fetch)", 143: "  writer wave: obs1 rows", 144: "  writer wave: step outputs", 146: "  writer wave: obs2 rows", 147: "  wave 1 starts after mark 10", 148: "  wave 8 obs1 rows", 149: "  wave 8 step outputs", 150: "  wave 8 starts after mark 10", 151: "  wave 14 obs1 rows", 152: "  wave 14 step outputs", 153: "  wave 14 starts after mark 10", 154: "  wave 1 done -> mark 12", 155: "  wave 8 done -> mark 12", 156: "  wave 14 done -> mark 12"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, default=256)
    ap.add_argument("--ticks", type=int, default=60)
    a = ap.parse_args()
    args = argparse.Namespace(worlds=a.worlds, workload="c4", seed=1)
    dw = bench.make_worlds(args, 0, "cuda:0")
    stamps = torch.zeros(128, dtype=torch.int64, device="cuda:0")
    lib = _lib.lib()
    acc = {}
    totals = []
    pops = []
    n_before = []
    samples = []
    refilled = []
    order = []
    for t in range(a.ticks):
        world = t % a.worlds
        _lib.check(lib.rl_bind_phase_profile(dw.handle, C.c_void_p(stamps.data_ptr()), world), "bind")
        epoch_before = int(dw.s["epoch"][world].item())
        n_before.append(int(dw.s["n_agents"][world].item()))
        stamps.zero_()
        dw.act()
        dw.tick_refill(70, 100)
        torch.cuda.synchronize()
        if t < 10:
            continue
        st = stamps.cpu().numpy()
        if st[23]:
            acc.setdefault(100, []).append(int(st[23] - st[0]))
            st[23] = 0
        order = [0, 30, 31, 32, 33, 34, 1, 2, 3, 35, 36, 37, 4, 5, 6, 8, 9, 10, 11, 12, 13, 38, 39, 40, 14, 41, 42, 15, 17, 18, 19, 20, 22, 21]
        for a_, b_, nm in ((43, 44, 143), (44, 45, 144), (46, 47, 146), (10, 43, 147), (48, 49, 148), (49, 50, 149), (10, 48, 150), (51, 52, 151), (52, 53, 152), (10, 51, 153), (45, 12, 154), (50, 12, 155), (53, 12, 156)):  # stamped by thread 64 (a writer wave)
            if st[a_] and st[b_]:
                acc.setdefault(nm, []).append(int(st[b_] - st[a_]))
        for k_ in (43, 44, 45, 48, 49, 50, 51, 52, 53, 13, 38, 39, 40, 14, 41, 42, 15, 12, 17):
            if st[k_] and st[10]:
                acc.setdefault(1000 + k_, []).append(int(st[k_] - st[10]))
        if st[64]:
            acc.setdefault(2000, []).append([int(st[64 + v] - st[10]) for v in range(16)])
            acc.setdefault(2001, []).append([int(st[80 + v] - st[10]) for v in range(16)])
        keys = [k for k in order if st[k] != 0]
        for prev, k in zip(keys[:-1], keys[1:]):
            acc.setdefault(k, []).append(int(st[k] - st[prev]))
        totals.append(int(st[keys[-1]] - st[keys[0]]))
        samples.append({k: int(st[k] - st[prev]) for prev, k in zip(keys[:-1], keys[1:])})
        pops.append(int(dw.s["n_agents"][world].item()))
        refilled.append(int(dw.s["epoch"][world].item()) != epoch_before)
    print("phase cycles (shader clock, thread 0 of the sampled world), mean over %d ticks" % len(totals))
    tot = np.mean(totals)
    for k in [k for k in order if k in acc] + [100, 143, 144, 146, 147, 148, 149, 150, 151, 152, 153, 154, 155, 156]:
        if k not in acc:
            continue
        m = np.mean(acc[k])
        print("  %2d %-34s %9.0f  %5.1f%%" % (k, NAMES.get(k, "?"), m, 100 * m / tot))
    print("  total %.0f cycles" % tot)
    tt, pp = np.asarray(totals), np.asarray(pops)
    rf = np.asarray(refilled)
    if rf.any():
        print("  sampled worlds that were refilled in their tick: %d of %d, total cycles mean %d (others: mean %d, max %d)"
              % (rf.sum(), len(rf), tt[rf].mean(), tt[~rf].mean(), tt[~rf].max()))
    nb = np.asarray(n_before[-len(tt):])
    lo, hi = (nb < 90) & ~rf, (nb >= 90) & ~rf
    if lo.any() and hi.any():
        print("  not refilled, by population before the tick (< 90 prepares a refill on its idle waves): %d worlds mean %d cycles | %d worlds mean %d cycles"
              % (lo.sum(), tt[lo].mean(), hi.sum(), tt[hi].mean()))
        for k in [k for k in order if k in acc]:
            a_ = np.mean([samples[i].get(k, 0) for i in np.nonzero(lo)[0]]); b_ = np.mean([samples[i].get(k, 0) for i in np.nonzero(hi)[0]])
            if abs(a_ - b_) > 150:
                print("    %2d %-34s %6.0f | %6.0f" % (k, NAMES.get(k, "?"), a_, b_))
    slow = np.argsort(tt)[-8:]
    print("  the 8 slowest samples against the mean, by phase (cycles):")
    for k in [k for k in order if k in acc]:
        print("    %2d %-34s mean %6.0f | slowest: %s" % (k, NAMES.get(k, "?"), np.mean(acc[k]), " ".join("%6d" % samples[i].get(k, 0) for i in slow)))
    print("       agents after the tick / refilled: %s" % " ".join("%d%s" % (pp[i], "R" if rf[i] else "") for i in slow))
    print("  per sampled world: total cycles min %d / median %d / p90 %d / max %d; agents after the tick min %d / median %d / max %d; corr(total, agents) %.2f"
          % (tt.min(), np.median(tt), np.percentile(tt, 90), tt.max(), pp.min(), np.median(pp), pp.max(), np.corrcoef(tt, pp)[0, 1]))
    if 2000 in acc:
        print("  per-wave arrival at the barrier ending the overlapped interval (from mark 10):", np.mean(acc.pop(2000), axis=0).astype(int).tolist())
        print("  per-wave release:", np.mean(acc.pop(2001), axis=0).astype(int).tolist())
    print("  offsets from mark 10 (start of the overlapped interval):", {k - 1000: int(np.mean(v)) for k, v in sorted(acc.items()) if k >= 1000})


if __name__ == "__main__":
    main()
