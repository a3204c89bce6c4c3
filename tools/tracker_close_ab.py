"""The Tracker's two-half interval close against the synchronous one: trainer(n_episodes=2000) at 256 worlds (four closed intervals),
loop seconds alternating between the two (tuning; GPU).   python tools/tracker_close_ab.py [update_interval]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reinlife_amd import Models
from reinlife_amd.Helpers import tracker as trk
from reinlife_amd.Helpers.trainer import trainer
warnings.simplefilter("ignore")
interval = int(sys.argv[1]) if len(sys.argv) > 1 else 500
orig = trk.Tracker.update_results


def call(defer):
    trk.Tracker.update_results = orig if defer else (lambda self, agents=None, n_epi=0, defer=False: orig(self, agents, n_epi, False))
    env = trainer([Models.PERD3QN(), Models.PERD3QN()], n_episodes=2000, n_worlds=256, save=False, print_results=False, synthetic_agents=100,
                  refill_below=70, update_interval=interval)
    steps = int(env.worlds.acted_total.item())
    return env.loop_seconds, steps, env.tracker.results["Avg Population Size"][0]


call(True); call(False)
for rep in range(4):
    for defer in (True, False):
        t, n, res = call(defer)
        print("%-12s loop %.3f ms  %.2f us/tick  %.4e agent-steps/s  (%d intervals, first %.6f)" % ("two halves" if defer else "synchronous", t * 1e3, t / 2001 * 1e6, n / t, len(res), res[0]), flush=True)
