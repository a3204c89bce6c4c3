"""What the TRAIN instantiation of k_run costs: per-tick time of rl_run launches at 256 worlds with (a) the plain kernel, (b) the TRAIN
kernel with a constant zero epsilon schedule, (c) + the Tracker, (d) + exploring brains (epsilon 0.05: the Philox draw per row), (e)
trainer() itself; and the host cost of one Environment.run call (tuning; GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=1, steps=20)
N = 1000
def rate(tag, tracking, eps):
    a = bench.make_worlds(args, 0, "cuda:0")
    if tracking:
        a.enable_tracking(True)
    sched = None if eps is None else torch.full((N, 2), float(eps), device="cuda:0")
    a.run(600, 70, 100); torch.cuda.synchronize()
    a.run(N, 70, 100, eps_schedule=sched)
    before = int(a.acted_total.item()); torch.cuda.synchronize()
    t0 = time.perf_counter(); a.run(N, 70, 100, eps_schedule=sched); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-46s %.2f us/tick  %.3e agent-steps/s" % (tag, dt / N * 1e6, (int(a.acted_total.item()) - before) / dt), flush=True)
for rep in range(2):
    rate("plain kernel", False, None)
    rate("TRAIN kernel, eps schedule == 0", False, 0.0)
    rate("TRAIN kernel, eps == 0, Tracker", True, 0.0)
    rate("TRAIN kernel, eps == 0.05, Tracker", True, 0.05)
    rate("TRAIN kernel, eps == 0.05", False, 0.05)
api = bench.api_trainer(args, "cuda:0")
print({k: api[k] for k in ("value", "us_per_tick", "value_at_steps", "window_ms")})
# host cost of Environment.run(k): the call returns after the launch is queued
import warnings
from reinlife_amd import Models
from reinlife_amd.World.environment import Environment
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    env = Environment(brains=[Models.PERD3QN(), Models.PERD3QN()], max_agents=100, print_results=False, n_worlds=256, seed=1, rng="philox", synthetic_agents=100, refill_below=70)
env.reset(); env.run(0, 30); torch.cuda.synchronize()
for k in (1, 20, 200):
    ts = []
    for i in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); env.run(31 + i * k, k); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts.sort(); print("Environment.run(%d): host %.0f us, with sync %.0f us" % (k, ts[3][0] * 1e6, ts[3][1] * 1e6), flush=True)
