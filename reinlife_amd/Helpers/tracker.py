"""Tracker: the numeric part of ReinLife/Helpers/tracker.py (per-gene statistics averaged every `update_interval`
episodes, tracker.py:107-121,166-176,279-282) on top of the accumulators the HIP step kernel maintains.

`results` has the reference's layout: {variable: {gene: [aggregate per interval, ...]}} plus the gene-less
"Avg Number of Populations".  With several replicas (and several GPUs) the valid per-tick values of ALL worlds are pooled:
aggregate = sum over worlds of trk_sum / sum over worlds of trk_cnt -- for one world this is exactly the reference's
np.mean over the interval (values <= -1 are dropped, as Tracker._aggregate does); across GPUs the same two arrays are
pooled with ONE RCCL collective per interval (an all-gather of the ranks' per-world rows, summed in global replica order on every rank:
the aggregates do not depend on the number of ranks).  Plots and the colab widgets are out of scope.

Closing an interval has two halves.  The first is queued on the device behind the launch that ends the interval: the rows, the
collective, the sum over worlds, a copy of it (and of the error flag) to pinned memory on a side stream, the zeroing of the running
sums.  The second runs on the host once the copy has arrived: the divisions, `results`, the printed table.  Environment.run() queues
the NEXT chunk of episodes between the two (`update_results(..., defer=True)` ... `resolve()`), so the device works through episode
501 ... while the host reads back and prints interval 1 ... 500; every other caller gets both halves at once.  `results` resolves
what is pending before it is read."""
import numpy as np
import torch

VARIABLES = ["Avg Population Size", "Avg Population Age", "Avg Population Fitness", "Best Population Age",
             "Avg Number of Attacks", "Avg Number of Kills", "Avg Number of Intra Kills", "Avg Number of Populations"]


class _Here:
    """Sums that are on the host already (CPU worlds, a CPU-only collective): the same protocol as DeviceWorlds.readback()."""

    def __init__(self, arrays):
        self.arrays = arrays

    def wait(self):
        return self.arrays


class Tracker:
    def __init__(self, update_interval, interactive=False, print_results=True, google_colab=False, nr_genes=None,
                 static_families=True, brains=None, *, worlds=None, dist=None):
        self.update_interval = update_interval
        self.print_results = print_results
        self.families = static_families
        self.nr_genes = nr_genes if static_families else 1
        self.variables = list(VARIABLES)
        self._results = {v: {g: [] for g in range(self.nr_genes)} for v in VARIABLES[:-1]}
        self._results["Avg Number of Populations"] = []
        self._pending = None            # a closed interval whose sums are still on their way to the host
        self.worlds = worlds
        self.dist = dist
        self.collectives_executed = 0   # collectives issued so far: exactly one per closed interval when a process group exists
        self.setup_collectives_executed = 0   # (+ the layout check at construction, below)
        self.fig = None
        if worlds is not None:
            worlds.enable_tracking(True)
            if dist is not None and dist.is_initialized():
                self._check_layout()

    @property
    def results(self):
        self.resolve()
        return self._results

    @results.setter
    def results(self, value):
        self._results = value

    def update_results(self, agents=None, n_epi=0, defer=False):
        """Called once per tick after step() (environment.py:206-207).  The per-tick statistics were accumulated inside
        the step kernel; this only closes an interval.  Episode 0 never reaches an aggregate in the reference
        (tracker.py:279-282 keeps the last `update_interval` entries of update_interval+1), so its contribution is dropped.
        defer=True leaves the host half of the close to resolve() (module docstring)."""
        if n_epi == 0:
            self.worlds.reset_tracking()
            return False
        if n_epi % self.update_interval == 0:
            self.resolve()
            self._average_results()
            if not defer:
                self.resolve()
            return True   # an interval was closed (defer=False: and the device synchronised)
        return False

    def resolve(self):
        """The host half of a closed interval, if one is pending: waits for ITS sums (not for the device), appends the aggregates to
        `results`, prints them, and raises if the error flag that travelled with them is set."""
        if self._pending is None:
            return
        (copies, shape), self._pending = self._pending, None
        arrays = copies.wait()
        self._append(arrays[0], shape)
        if self.print_results and (self.dist is None or not self.dist.is_initialized() or self.dist.get_rank() == 0):
            self._print_results()
        if len(arrays) > 1:
            self.worlds.raise_on_error_flag(arrays[1])

    def _check_layout(self):
        """Set-up, once per Tracker under a process group: every rank must hold the SAME number of worlds (the interval collective is an
        all-gather of equal rows) at world_base = rank x n_worlds (the pooled sums are taken in global replica order, which is what
        makes them independent of the number of ranks).  One tiny all-gather of (world_base, n_worlds); a layout that would hang or
        mis-order the interval collective is refused here, on every rank, with the table in the message."""
        from ..distributed import gather_rows
        w = self.worlds
        mine = torch.tensor([float(getattr(w, "world_base", 0)), float(w.trk_sum.shape[0])], dtype=torch.float64, device=w.trk_sum.device)
        table = gather_rows(mine, self.dist).cpu().numpy()
        self.setup_collectives_executed += 1
        n = table[0, 1]
        if not (table[:, 1] == n).all() or not (table[:, 0] == table[0, 0] + n * np.arange(len(table))).all():
            raise ValueError("Tracker under a process group needs equal shards in rank order (world_base = first + rank * n_worlds); "
                             "got (world_base, n_worlds) per rank: %s" % table.astype(np.int64).tolist())

    def _average_results(self):
        """The device half: queued, not waited for."""
        w = self.worlds
        rows = torch.cat([w.trk_sum.reshape(w.trk_sum.shape[0], -1), w.trk_cnt.reshape(w.trk_cnt.shape[0], -1).to(torch.float64),
                          w.trk_pop[:, 1:]], dim=1)   # one row per world of this rank: [sums | counts | population sums]
        err = getattr(w, "err", None)
        if self.dist is not None and self.dist.is_initialized():   # (also at world size 1: the collective is the same code path)
            # ONE collective per closed interval.  The ranks' PER-WORLD rows are gathered (a few hundred bytes per world and interval)
            # and every rank then sums the job's worlds in global replica order with the same reduction a single-rank job uses: the
            # aggregates are bit-identical whatever the number of ranks (a sum of per-rank partial sums -- an all-reduce(SUM) -- would
            # re-associate the float64 additions: equal to 1e-16 relative, not exactly).  The rank's device error flag rides in the same
            # row, so a corrupted world on ANY rank stops EVERY rank at this interval (no rank is left waiting in the next collective).
            from ..distributed import gather_rows
            ncol = rows.shape[1]
            payload = rows.reshape(-1) if err is None else torch.cat([rows.reshape(-1), err.to(torch.float64)])
            table = gather_rows(payload, self.dist)   # (under a CPU-only backend the few hundred bytes go through the host)
            self.collectives_executed += 1
            if err is not None:
                errs = table[:, -err.numel():]
                err = errs[(errs[:, 0] != 0).to(torch.int32).argmax()].to(torch.int32)   # the first rank's flag that is set, else zeros
                table = table[:, :-err.numel()]
            rows = table.reshape(-1, ncol)
        tot = rows.sum(0)
        shape = tuple(w.trk_sum.shape[1:])
        if hasattr(w, "readback") and tot.device.type == torch.device(w.device).type:   # (not after a hop through a CPU-only collective)
            self._pending = (w.readback([tot, err]), shape)
        else:   # sums that are on the host already; the error flag comes along (one small read-back when it is still on the device)
            self._pending = (_Here([tot.cpu().numpy()] + ([err.cpu().numpy()] if err is not None else [])), shape)
        w.reset_tracking()

    def _append(self, tot, shape):
        n = int(np.prod(shape))
        s, c, pop = tot[:n].reshape(shape), tot[n:2 * n].reshape(shape), tot[2 * n:]
        with np.errstate(invalid="ignore", divide="ignore"):
            agg = s / c  # nan when a variable had no valid tick, like np.mean([])
            pagg = pop[0] / pop[1]
        for i, v in enumerate(VARIABLES[:-1]):
            for g in range(self.nr_genes):
                self._results[v][g].append(float(agg[g, i]))
        self._results["Avg Number of Populations"].append(float(pagg))

    def _print_results(self):
        cols = ["Gene"] + [v.replace(" Population", "").replace("Number", "Nr").replace("of ", "") for v in VARIABLES[:-1]]
        line = "+" + "+".join("-" * (len(c) + 2) for c in cols) + "+"
        print("\n" + line + "\n|" + "|".join(" %s " % c for c in cols) + "|\n" + line)
        for g in range(self.nr_genes):
            vals = [str(g)] + [str(round(self._results[v][g][-1], 2)) for v in VARIABLES[:-1]]
            print("|" + "|".join(x.center(len(c) + 2) for x, c in zip(vals, cols)) + "|\n" + line)
        print()
