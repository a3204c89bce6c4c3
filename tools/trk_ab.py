"""TRAIN launches with / without the Tracker, per-tick time at 256 worlds, for A/B of builds (REINLIFE_HIP_LIB=...; tuning; GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
args = __import__("argparse").Namespace(worlds=256, workload=os.environ.get("RL_AB_WORKLOAD", "c4"), seed=1, steps=20)
N = 1000
def rate(tracking):
    a = bench.make_worlds(args, 0, "cuda:0")
    if tracking:
        a.enable_tracking(True)
    sched = torch.zeros((N, a.n_brains), device="cuda:0")
    a.run(600, 70, 100); torch.cuda.synchronize()
    a.run(N, 70, 100, eps_schedule=sched); torch.cuda.synchronize()
    t0 = time.perf_counter(); a.run(N, 70, 100, eps_schedule=sched); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6
print("%s: TRAIN %.2f us/tick, TRAIN + Tracker %.2f us/tick" % (os.path.basename(os.environ.get("REINLIFE_HIP_LIB", "product")), rate(False), rate(True)), flush=True)
