"""Where the literal drop-in default -- trainer([DQN(max_epi=300), DQN(max_epi=300)], n_episodes=300, save=False, print_results=False): one
world, rng="reference", the host makes the reference's draws tick by tick -- spends its host time (tuning; GPU).
    python tools/trainer_default_profile.py [episodes]"""
import cProfile
import os
import pstats
import random
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from reinlife_amd import Models  # noqa: E402
from reinlife_amd.Helpers.trainer import trainer  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 300
warnings.simplefilter("ignore")


def call(n):
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    return trainer([Models.DQN(max_epi=300), Models.DQN(max_epi=300)], n_episodes=n, save=False, print_results=False)


call(30)
env = call(k)
print("trainer(%d): %.1f us per tick, %d agent-steps" % (k, env.loop_seconds / (k + 1) * 1e6, int(env.worlds.acted_total.item())))
pr = cProfile.Profile()
pr.enable()
env = call(k)
pr.disable()
print("profiled: %.1f us per tick" % (env.loop_seconds / (k + 1) * 1e6))
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
