"""Which part of the Tracker pass costs what in the TRAIN launch (tuning build with measurement masks; results of masked launches are wrong)."""
import os, sys, time
os.environ.setdefault("RL_TUNE", "1")   # the tuning library (libreinlife_hip_tune.so) carries rl_debug_set_run_mask; the product does not
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from reinlife_amd import _lib
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=1, steps=20)
N = 1000
def rate(tag, tracking, mask):
    a = bench.make_worlds(args, 0, "cuda:0")
    if tracking:
        a.enable_tracking(True)
    sched = torch.zeros((N, 2), device="cuda:0")
    a.run(600, 70, 100); torch.cuda.synchronize()
    _lib.lib().rl_debug_set_run_mask(mask)
    a.run(N, 70, 100, eps_schedule=sched); torch.cuda.synchronize()
    t0 = time.perf_counter(); a.run(N, 70, 100, eps_schedule=sched); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    _lib.lib().rl_debug_set_run_mask(0)
    print("%-60s %.2f us/tick" % (tag, dt / N * 1e6), flush=True)
for rep in range(2):
    rate("TRAIN, no Tracker", False, 0)
    rate("TRAIN + Tracker", True, 0)
    rate("TRAIN + Tracker, pass skipped entirely (32)", True, 32)
    rate("  without the statistics pass (128)", True, 128)
    rate("  without the grouped rewards (64)", True, 64)
    rate("  without the pairwise sums (256)", True, 256)
    rate("  without statistics, rewards and sums (128+64+256)", True, 128 + 64 + 256)
