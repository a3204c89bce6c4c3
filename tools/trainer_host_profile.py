"""Where a SHORT trainer() call spends its host time (tuning; GPU).   python tools/trainer_host_profile.py [episodes]
cProfile over three 20-episode trainer() calls behind a warm-up call (the driver's 20-step window through the API: bench.py api_trainer.value_at_steps)."""
import cProfile
import os
import pstats
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from reinlife_amd import Models  # noqa: E402
from reinlife_amd.Helpers.trainer import trainer  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warnings.simplefilter("ignore")


def call(n):
    env = trainer([Models.PERD3QN(), Models.PERD3QN()], n_episodes=n, n_worlds=256, save=False, print_results=False, synthetic_agents=100, refill_below=70)
    return env


call(2000)
for _ in range(3):
    t0 = time.perf_counter(); env = call(k); t1 = time.perf_counter()
    print("trainer(%d): loop %.1f us, whole call %.1f us, %.3e agent-steps/s in the loop" % (k, env.loop_seconds * 1e6, (t1 - t0) * 1e6, int(env.worlds.acted_total.item()) / env.loop_seconds))
pr = cProfile.Profile()
env = trainer([Models.PERD3QN(), Models.PERD3QN()], n_episodes=0, n_worlds=256, save=False, print_results=False, synthetic_agents=100, refill_below=70)
torch.cuda.synchronize()
pr.enable()
for r in range(20):
    env.run(0, k + 1)
    torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
