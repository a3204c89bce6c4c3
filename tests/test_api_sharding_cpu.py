"""CPU, 2 processes over gloo: replica sharding THROUGH THE API (trainer() / Environment / Tracker), SURVEY.md 8e + 8f-1.

The device layer is stood in for by the oracle (tests may use it; the product never does): real worlds, real per-tick Tracker statistics,
actions a pure function of (global replica id, tick, slot).  What is under test is the product's host side: Environment takes its rank
from the process group, keys its worlds by global replica id (world_base = rank * n_worlds), and the Tracker pools every closed interval
over ALL ranks' worlds with ONE collective -- so trainer() on 2 ranks x 3 worlds returns, on every rank, exactly the Tracker.results of
trainer() on 1 rank x 6 worlds."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_EPISODES, INTERVAL, PER_RANK, SEED = 45, 15, 3, 4242


class OracleBackedWorlds:
    """What Environment / Tracker need of DeviceWorlds, computed by the CPU oracle."""

    def __init__(self, n_worlds=1, width=30, height=30, max_agents=100, n_brains=2, static_families=True, limit_reproduction=False,
                 incentivize_killing=True, seed=0, device="cpu", world_base=0, **kw):
        from oracle import oracle as orc
        self.ow = orc.OracleWorlds(n_worlds=n_worlds, width=width, height=height, max_agents=max_agents, n_brains=n_brains,
                                   static_families=static_families, limit_reproduction=limit_reproduction,
                                   incentivize_killing=incentivize_killing, seed=seed, world_base=world_base)
        self.R, self.cap, self.n_brains, self.world_base = n_worlds, self.ow.cap, n_brains, world_base
        self.device = torch.device("cpu")
        self.trk_sum, self.trk_cnt, self.trk_pop = (torch.from_numpy(a) for a in (self.ow.trk_sum, self.ow.trk_cnt, self.ow.trk_pop))
        self.s = {k: torch.from_numpy(v) for k, v in self.ow.s.items()}
        self.tracking, self.launches, self.acted = False, 0, 0
        self.cfg = self.ow.cfg

    def enable_tracking(self, on=True):
        self.tracking = on

    def reset_tracking(self):
        self.trk_sum.zero_(); self.trk_cnt.zero_(); self.trk_pop[:, 1:].zero_()

    def set_brains(self, brains):
        self.eps = [e for _, e, _ in brains]

    def _set_epsilons(self, eps):
        self.eps = list(eps)

    def reset_synthetic(self, n):
        self.ow.reset_synthetic(n)

    def reset_families(self):
        self.ow.reset_families()

    def load_world(self, w, snap):
        self.ow.load_world(w, snap)

    def observe(self):
        self.ow.observe()

    def check_error_flag(self):
        pass

    def run(self, n_ticks, threshold=-1, n_agents=0, eps_schedule=None, trk_skip=0):
        keep = None
        if trk_skip > 0:
            keep = (self.trk_sum.clone(), self.trk_cnt.clone(), self.trk_pop[:, 1:].clone())
        for t in range(n_ticks):
            acts = np.zeros((self.R, self.cap), np.int8)
            for w in range(self.R):   # a pure function of (global replica, the world's own tick and epoch): layout-independent
                tick, epoch = int(self.ow.s["tick"][w]), int(self.ow.s["epoch"][w])
                acts[w] = np.random.RandomState((self.world_base + w) * 1_000_003 + epoch * 7919 + tick).randint(0, 8, self.cap)
            self.acted += int(self.ow.s["n_agents"].sum())
            self.ow.step(acts)
            self.ow.update()
            if threshold >= 0:
                self.ow.refill(threshold, n_agents)
            if keep is not None and t + 1 == trk_skip:
                self.trk_sum.copy_(keep[0]); self.trk_cnt.copy_(keep[1]); self.trk_pop[:, 1:].copy_(keep[2])
        self.launches += 1


def _patch(monkeypatch=None):
    """The stand-in device layer: for good in a worker process, through pytest's monkeypatch (undone afterwards) in the test process."""
    sys.path.insert(0, ROOT)
    from reinlife_amd.World import environment as envmod
    from reinlife_amd import Models
    put = monkeypatch.setattr if monkeypatch is not None else setattr
    put(envmod, "DeviceWorlds", OracleBackedWorlds)
    put(envmod.torch.cuda, "synchronize", lambda *a, **k: None)
    put(Models.brains._HipBrain, "packed_weights", lambda self, device="cuda:0": None)
    return envmod, Models


def _train(n_worlds, dist=None, static_families=True, monkeypatch=None):
    import warnings
    envmod, Models = _patch(monkeypatch)
    from reinlife_amd.Helpers.trainer import trainer
    np.random.seed(5)   # (replica 0 of the job is built from the process-global generator, environment.py:147-154)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env = trainer([Models.PERD3QN(), Models.D3QN()], n_episodes=N_EPISODES, update_interval=INTERVAL, n_worlds=n_worlds,
                      save=False, print_results=False, rng="philox", seed=SEED, static_families=static_families, dist=dist)
    return env


def _worker(rank, world_size, port, q, pass_dist):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    env = _train(PER_RANK, dist if pass_dist else None)   # (None: Environment finds the initialised default group by itself)
    q.put((rank, env.world_base, env.rank, env.world_size, env.device, env.tracker.results, env.tracker.collectives_executed,
           env.worlds.ow.s["cell_type"].copy(), env.worlds.acted))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pass_dist", [True, False])
def test_trainer_on_two_ranks_equals_one_rank_with_all_the_worlds(pass_dist, monkeypatch):
    ref = _train(2 * PER_RANK, monkeypatch=monkeypatch)
    assert ref.dist is None and ref.world_base == 0 and ref.tracker.collectives_executed == 0
    n_int = N_EPISODES // INTERVAL
    assert len(ref.tracker.results["Avg Number of Populations"]) == n_int
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + (7 if pass_dist else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, pass_dist)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    acted = 0
    for rank, base, erank, wsize, device, results, n_coll, cells, a in res:
        assert (base, erank, wsize) == (rank * PER_RANK, rank, 2)
        assert device == "cuda:%d" % rank                      # cuda:LOCAL_RANK unless the caller names a device
        assert n_coll == n_int                                  # ONE collective per closed interval, nothing else
        assert np.array_equal(cells, ref.worlds.ow.s["cell_type"][base:base + PER_RANK])   # the same replicas, whatever the layout
        assert results.keys() == ref.tracker.results.keys()
        for var, want in ref.tracker.results.items():          # exactly: ==, not allclose (nan == nan where no tick was valid)
            if isinstance(want, dict):
                for g in want:
                    assert np.array_equal(np.array(results[var][g]), np.array(want[g]), equal_nan=True), (var, g)
            else:
                assert np.array_equal(np.array(results[var]), np.array(want), equal_nan=True), var
        acted += a
    assert acted == ref.worlds.acted
    assert any(np.isfinite(v).all() and len(v) == n_int for v in ref.tracker.results["Avg Population Size"].values())


def test_environment_world_base_and_device_resolution_without_a_process_group(monkeypatch):
    envmod, Models = _patch(monkeypatch)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env = envmod.Environment(brains=[Models.PERD3QN(), Models.PPO()], n_worlds=2, rng="philox")
        assert (env.dist, env.rank, env.world_size, env.world_base, env.device) == (None, 0, 1, 0, "cuda:0")
        env = envmod.Environment(brains=[Models.PERD3QN()], n_worlds=4, rng="philox", world_base=12, device="cuda:3")
        assert env.world_base == 12 and env.worlds.world_base == 12 and env.device == "cuda:3"
        env.reset()   # a rank that does not hold the job's replica 0 builds every world on the device (no np.random draw)
        st = np.random.get_state()[1].copy()
        env.reset()
        assert np.array_equal(st, np.random.get_state()[1])


# ---------------------------------------------------------------------------------------------------------------------
# round 5: what a sharded job refuses, and how a device error on ONE rank reaches EVERY rank
# ---------------------------------------------------------------------------------------------------------------------
class FlaggedWorlds(OracleBackedWorlds):
    """The stand-in with a device error flag (DeviceWorlds.err / raise_on_error_flag)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.err = torch.zeros(4, dtype=torch.int32)

    @staticmethod
    def raise_on_error_flag(e):
        from reinlife_amd import _lib
        if e[0] != 0:
            raise _lib.ReinLifeHipError("device error flag: code %d world %d detail (%d, %d)" % tuple(int(x) for x in e))


def _edge_worker(rank, world_size, port, q):
    import warnings
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    envmod, Models = _patch()
    envmod.DeviceWorlds = FlaggedWorlds
    from reinlife_amd import _lib
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # (1) the trainer()/tester() default n_worlds=1 under a multi-rank job: keyed Philox streams, one distinct replica per rank
        env = envmod.Environment(brains=[Models.PERD3QN(), Models.PPO()])
        out["default"] = (env.rng, env.world_base, env.worlds.world_base, env.tracker.setup_collectives_executed)
        try:   # (every rank raises before the Tracker's collective: nobody is left waiting)
            envmod.Environment(brains=[Models.PERD3QN()], rng="reference")
            out["reference"] = "accepted"
        except ValueError as e:
            out["reference"] = str(e)
        # (2) unequal shards: refused on EVERY rank at construction, not a hang in the first interval's collective
        try:
            envmod.Environment(brains=[Models.PERD3QN()], n_worlds=2 + rank)
            out["unequal"] = "accepted"
        except ValueError as e:
            out["unequal"] = str(e)
        # (3) a device error on rank 1 only: both ranks raise when the interval that carried it is resolved
        env = envmod.Environment(brains=[Models.PERD3QN(), Models.D3QN()], n_worlds=2, update_interval=5, print_results=False, seed=3)
        env.reset()
        env.run(0, 5)            # episodes 0..4: no interval closes
        if rank == 1:
            env.worlds.err[:] = torch.tensor([7, 1, 2, 3], dtype=torch.int32)
        try:
            env.run(5, 6)        # episode 5 closes interval 1; its flag travels with the gathered rows
            out["flag"] = "no error"
        except _lib.ReinLifeHipError as e:
            out["flag"] = str(e)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_a_sharded_job_refuses_what_it_cannot_shard_and_shares_its_error_flag():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_edge_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(2):
        out = res[rank]
        assert out["default"] == ("philox", rank, rank, 1)
        assert "cannot be one shard" in out["reference"]
        assert "equal shards" in out["unequal"] and "[[0, 2], [3, 3]]" in out["unequal"]
        assert out["flag"] == "device error flag: code 7 world 1 detail (2, 3)"   # rank 1's flag, on both ranks
