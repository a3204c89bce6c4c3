"""When do the bench worlds refill (tuning aid, GPU only): world tick and population before the refilling tick.
Measured: 90 % at world tick >= 19 (the reset cohort starves together), 98 % from fewer than 85 agents -- which is why every
world below threshold + 20 prepares its refill (rl_world.hip, kSpecMargin): a launch lasts as long as its slowest world, so a
predictor that misses 3 % of the refills loses more than it saves."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
args = argparse.Namespace(worlds=256, workload="c4", seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
hist = {}; nb_hist = {}
for t in range(400):
    tick0 = dw.s["tick"].cpu().numpy().copy(); ep0 = dw.s["epoch"].cpu().numpy().copy(); n0 = dw.s["n_agents"].cpu().numpy().copy()
    dw.act(); dw.tick_refill(70, 100)
    ep1 = dw.s["epoch"].cpu().numpy()
    for w in np.nonzero(ep1 != ep0)[0]:
        hist[int(tick0[w])] = hist.get(int(tick0[w]), 0) + 1
        nb_hist[int(n0[w]) // 5 * 5] = nb_hist.get(int(n0[w]) // 5 * 5, 0) + 1
print("world tick (before the refilling tick) -> refills:", sorted(hist.items()))
print("population before the refilling tick (bins of 5) -> refills:", sorted(nb_hist.items()))
