"""Saver: the on-disk layout of ReinLife/Helpers/saver.py:58-194 --

    <main_folder>/<date>_V<n>/<METHOD>/brain_gene_<g>.pt | brain_<k>.pt      torch.save(state_dict)  (entities.py:224-242)
                                       parameters_gene_<g>.json | parameters_<k>.json   scalar brain attributes (saver.py:170-194)
                              results.json   Tracker.results            settings.json   run settings (environment.py:246-250)

so that brains saved here load into the reference (`load_model=`) and vice versa: the state-dict key names and shapes are
the reference's (tests/golden/state_dict_keys.json).  Differences: paths are joined with os.sep (the reference hard-codes a
backslash outside colab, saver.py:55), and no matplotlib figure is written."""
import inspect
import json
import os
from datetime import date

import torch


class SavedAgent:
    """What Saver.save needs of an agent: .gene and .brain (environment.py:253 builds `Agent(gene=gene, brain=brain)`)."""

    def __init__(self, gene, brain):
        self.gene, self.brain = gene, brain

    def save_brain(self, path):  # entities.py:224-242
        net = {"DQN": "agent", "D3QN": "eval_net", "PERD3QN": "eval_net", "PPO": "model"}[self.brain.method]
        torch.save(getattr(self.brain, net).state_dict(), path + ".pt")


class Saver:
    def __init__(self, main_folder, google_colab=False):
        self.google_colab = google_colab
        self.separator = os.sep
        self.main_folder = os.path.join(os.getcwd(), main_folder)

    def _get_paths(self, agents, family):
        today = str(date.today())
        version = 1
        if os.path.isdir(self.main_folder):
            prev = [int(p.split("V")[-1]) for p in os.listdir(self.main_folder) if today in p and p.split("V")[-1].isdigit()]
            version = max(prev) + 1 if prev else 1
        experiment = os.path.join(self.main_folder, "%s_V%d" % (today, version))
        paths = {}
        per_method = {}
        for a in agents:
            m = a.brain.method
            if family:
                paths[a] = os.path.join(experiment, m, "brain_gene_%d" % a.gene)
            else:
                per_method[m] = per_method.get(m, 0) + 1
                paths[a] = os.path.join(experiment, m, "brain_%d" % per_method[m])
        return experiment, paths

    def save(self, agents, family, results, settings, fig=None):
        experiment, paths = self._get_paths(agents, family)
        for a in agents:
            os.makedirs(os.path.dirname(paths[a]), exist_ok=True)
            a.save_brain(paths[a])
            params = {n: v for n, v in inspect.getmembers(a.brain, lambda x: not inspect.isroutine(x))
                      if type(v) in (float, int, bool, str) and not n.startswith("__")}
            params.setdefault("one", 1)   # a constant every reference brain carries (Models/utils.py:7) and therefore every parameters file
            d, f = os.path.split(paths[a])
            with open(os.path.join(d, f.replace("brain", "parameters") + ".json"), "w") as fh:
                json.dump(params, fh, indent=4)
        with open(os.path.join(experiment, "results.json"), "w") as fh:
            json.dump(results, fh, indent=4)
        with open(os.path.join(experiment, "settings.json"), "w") as fh:
            json.dump(settings, fh, indent=4)
        return experiment
