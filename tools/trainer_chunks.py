"""trainer()'s loop chunk by chunk: wall time, agent-steps and refills of each Environment.run chunk of a training=True run on the
benchmark's synthetic worlds (exploring brains: epsilon 0.9 -> 0.05), next to a greedy run (tuning; GPU)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from reinlife_amd import Models
from reinlife_amd.World.environment import Environment
for training in (True, False):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        brains = [Models.PERD3QN(training=training), Models.PERD3QN(training=training)]
        env = Environment(brains=brains, max_agents=100, print_results=False, n_worlds=256, seed=1, rng="philox", synthetic_agents=100, refill_below=70, training=training)
    env.reset(); env._sync()
    n_epi = 0
    for k in (1, 100, 100, 100, 200, 500, 500, 500):
        a0, r0 = int(env.worlds.acted_total.item()), int(env.worlds.refill_count.item())
        torch.cuda.synchronize(); t0 = time.perf_counter(); env.run(n_epi, k); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        a1, r1 = int(env.worlds.acted_total.item()), int(env.worlds.refill_count.item())
        print("training=%s episodes %4d..%4d: %.2f us/tick, %.1f agents/world, %.3f refills/tick, eps %.3f, %.3e agent-steps/s" % (
            training, n_epi, n_epi + k - 1, dt / k * 1e6, (a1 - a0) / k / 256, (r1 - r0) / k, brains[0].epsilon, (a1 - a0) / dt), flush=True)
        n_epi += k
