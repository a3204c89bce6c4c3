"""Tracker: the numeric part of ReinLife/Helpers/tracker.py (per-gene statistics averaged every `update_interval`
episodes, tracker.py:107-121,166-176,279-282) on top of the accumulators the HIP step kernel maintains.

`results` has the reference's layout: {variable: {gene: [aggregate per interval, ...]}} plus the gene-less
"Avg Number of Populations".  With several replicas (and several GPUs) the valid per-tick values of ALL worlds are pooled:
aggregate = sum over worlds of trk_sum / sum over worlds of trk_cnt -- for one world this is exactly the reference's
np.mean over the interval (values <= -1 are dropped, as Tracker._aggregate does); across GPUs the same two arrays are
pooled with ONE RCCL collective per interval (an all-gather of the ranks' per-world rows, summed in global replica order on every rank:
the aggregates do not depend on the number of ranks).  Plots and the colab widgets are out of scope."""
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
        self.collectives_executed = 0   # collectives issued so far: exactly one per closed interval when a process group exists
        self.fig = None
        if worlds is not None:
            worlds.enable_tracking(True)

    def update_results(self, agents=None, n_epi=0):
        """Called once per tick after step() (environment.py:206-207).  The per-tick statistics were accumulated inside
        the step kernel; this only closes an interval.  Episode 0 never reaches an aggregate in the reference
        (tracker.py:279-282 keeps the last `update_interval` entries of update_interval+1), so its contribution is dropped."""
        if n_epi == 0:
            self.worlds.reset_tracking()
            return False
        if n_epi % self.update_interval == 0:
            self._average_results()
            if self.print_results and (self.dist is None or not self.dist.is_initialized() or self.dist.get_rank() == 0):
                self._print_results()
            return True   # an interval was closed (and the device synchronised)
        return False

    def _average_results(self):
        w = self.worlds
        rows = torch.cat([w.trk_sum.reshape(w.trk_sum.shape[0], -1), w.trk_cnt.reshape(w.trk_cnt.shape[0], -1).to(torch.float64),
                          w.trk_pop[:, 1:]], dim=1)   # one row per world of this rank: [sums | counts | population sums]
        if self.dist is not None and self.dist.is_initialized():   # (also at world size 1: the collective is the same code path)
            # ONE collective per closed interval.  The ranks' PER-WORLD rows are gathered (a few hundred bytes per world and interval)
            # and every rank then sums the job's worlds in global replica order with the same reduction a single-rank job uses: the
            # aggregates are bit-identical whatever the number of ranks (a sum of per-rank partial sums -- an all-reduce(SUM) -- would
            # re-associate the float64 additions: equal to 1e-16 relative, not exactly)
            backend = getattr(self.dist, "get_backend", lambda: "")()
            if rows.is_cuda and str(backend) == "gloo":   # (a CPU-only backend under GPU worlds: the few hundred bytes go through the host)
                rows = rows.cpu()
            flat = torch.empty(self.dist.get_world_size() * rows.numel(), dtype=torch.float64, device=rows.device)
            self.dist.all_gather_into_tensor(flat, rows.reshape(-1).contiguous())
            self.collectives_executed += 1
            rows = flat.view(-1, rows.shape[1])
        tot = rows.sum(0).cpu().numpy()
        n = w.trk_sum[0].numel()
        s, c, pop = tot[:n].reshape(tuple(w.trk_sum.shape[1:])), tot[n:2 * n].reshape(tuple(w.trk_sum.shape[1:])), tot[2 * n:]
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
