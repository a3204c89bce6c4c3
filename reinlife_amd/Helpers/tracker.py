"""Tracker: the numeric part of ReinLife/Helpers/tracker.py (per-gene statistics averaged every `update_interval`
episodes, tracker.py:107-121,166-176,279-282) on top of the accumulators the HIP step kernel maintains.

`results` has the reference's layout: {variable: {gene: [aggregate per interval, ...]}} plus the gene-less
"Avg Number of Populations".  With several replicas (and several GPUs) the valid per-tick values of ALL worlds are pooled:
aggregate = sum over worlds of trk_sum / sum over worlds of trk_cnt -- for one world this is exactly the reference's
np.mean over the interval (values <= -1 are dropped, as Tracker._aggregate does); across GPUs the same two arrays are
summed with one RCCL all-reduce per interval.  Plots and the colab widgets are out of scope."""
import numpy as np
import torch

VARIABLES = ["Avg Population Size", "Avg Population Age", "Avg Population Fitness", "Best Population Age",
             "Avg Number of Attacks", "Avg Number of Kills", "Avg Number of Intra Kills", "Avg Number of Populations"]


class Tracker:
    def __init__(self, update_interval, interactive=False, print_results=True, google_colab=False, nr_genes=None,
                 static_families=True, brains=None, *, worlds=None, dist=None):
        self.update_interval = update_interval
        self.print_results = print_results
        self.families = static_families
        self.nr_genes = nr_genes if static_families else 1
        self.variables = list(VARIABLES)
        self.results = {v: {g: [] for g in range(self.nr_genes)} for v in VARIABLES[:-1]}
        self.results["Avg Number of Populations"] = []
        self.worlds = worlds
        self.dist = dist
        self.fig = None
        if worlds is not None:
            worlds.enable_tracking(True)

    def update_results(self, agents=None, n_epi=0):
        """Called once per tick after step() (environment.py:206-207).  The per-tick statistics were accumulated inside
        the step kernel; this only closes an interval.  Episode 0 never reaches an aggregate in the reference
        (tracker.py:279-282 keeps the last `update_interval` entries of update_interval+1), so its contribution is dropped."""
        if n_epi == 0:
            self.worlds.reset_tracking()
            return
        if n_epi % self.update_interval == 0:
            self._average_results()
            if self.print_results:
                self._print_results()

    def _average_results(self):
        w = self.worlds
        s = w.trk_sum.sum(0)
        c = w.trk_cnt.sum(0).to(torch.float64)
        pop = w.trk_pop[:, 1:].sum(0)
        if self.dist is not None and self.dist.is_initialized():   # (also at world size 1: the collective is the same code path)
            buf = torch.cat([s.reshape(-1), c.reshape(-1), pop])
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)  # one small fused buffer per interval
            n = s.numel()
            s, c, pop = buf[:n].view_as(s), buf[n:2 * n].view_as(c), buf[2 * n:]
        s, c, pop = s.cpu().numpy(), c.cpu().numpy(), pop.cpu().numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            agg = s / c  # nan when a variable had no valid tick, like np.mean([])
            pagg = pop[0] / pop[1]
        for i, v in enumerate(VARIABLES[:-1]):
            for g in range(self.nr_genes):
                self.results[v][g].append(float(agg[g, i]))
        self.results["Avg Number of Populations"].append(float(pagg))
        w.reset_tracking()

    def _print_results(self):
        cols = ["Gene"] + [v.replace(" Population", "").replace("Number", "Nr").replace("of ", "") for v in VARIABLES[:-1]]
        line = "+" + "+".join("-" * (len(c) + 2) for c in cols) + "+"
        print("\n" + line + "\n|" + "|".join(" %s " % c for c in cols) + "|\n" + line)
        for g in range(self.nr_genes):
            vals = [str(g)] + [str(round(self.results[v][g][-1], 2)) for v in VARIABLES[:-1]]
            print("|" + "|".join(x.center(len(c) + 2) for x, c in zip(vals, cols)) + "|\n" + line)
        print()
