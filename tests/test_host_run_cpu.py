"""CPU: the host side of the multi-tick loop -- Environment.run / trainer() chunking, epsilon schedules, Tracker boundaries, lazy host
mirrors -- against a stand-in for the device layer that records what would be launched (no GPU here; the launches themselves are covered by
tests/test_hip_round3.py)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeWorlds:
    """What Environment needs of DeviceWorlds, on the host: records run() calls, keeps trivial state tensors."""

    def __init__(self, n_worlds=1, width=30, height=30, max_agents=100, n_brains=2, static_families=True, **kw):
        self.R, self.cap, self.n_brains = n_worlds, 256, n_brains
        self.device = torch.device("cpu")
        self.G = n_brains if static_families else 1
        self.calls, self.resets, self.tracking = [], [], False
        self.s = {"n_agents": torch.zeros(n_worlds, dtype=torch.int32), "max_gene": torch.full((n_worlds,), n_brains, dtype=torch.int32),
                  "cell_type": torch.zeros((n_worlds, width * height), dtype=torch.uint8)}
        self.trk_sum = torch.zeros((n_worlds, self.G, 7), dtype=torch.float64)
        self.trk_cnt = torch.zeros((n_worlds, self.G, 7), dtype=torch.int32)
        self.trk_pop = torch.zeros((n_worlds, 3), dtype=torch.float64)
        self.cfg = type("cfg", (), {"seed": 0, "world_base": 0})()

    def enable_tracking(self, on=True):
        self.tracking = on

    def reset_tracking(self):
        self.calls.append(("reset_tracking",))
        self.trk_sum.zero_(); self.trk_cnt.zero_(); self.trk_pop.zero_()

    def set_brains(self, brains):
        self.brains = [(k, e) for k, e, _ in brains]
        self.calls.append(("set_brains",))

    def _set_epsilons(self, eps):
        self.brains = [(k, e) for (k, _), e in zip(self.brains, eps)]

    def run(self, n_ticks, threshold=-1, n_agents=0, eps_schedule=None, trk_skip=0):
        self.calls.append(("run", n_ticks, threshold, n_agents, None if eps_schedule is None else np.array(eps_schedule), trk_skip,
                           [e for _, e in self.brains]))
        self.trk_sum += n_ticks - trk_skip; self.trk_cnt += n_ticks - trk_skip     # one valid "value 1" per counted tick
        self.trk_pop[:, 1:] += n_ticks - trk_skip

    def reset_synthetic(self, n):
        self.resets.append(("synthetic", n))

    def reset_families(self):
        self.resets.append(("families",))

    def load_world(self, w, snap):
        self.resets.append(("load", w))

    def observe(self):
        pass

    def check_error_flag(self):
        pass


@pytest.fixture
def fake_env(monkeypatch):
    from reinlife_amd.World import environment as envmod
    from reinlife_amd import Models
    monkeypatch.setattr(envmod, "DeviceWorlds", _FakeWorlds)
    monkeypatch.setattr(envmod.torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(Models.brains._HipBrain, "packed_weights", lambda self, device="cuda:0": None)

    def make(training=True, update_interval=25, n_worlds=4, **kw):
        brains = [Models.PERD3QN(training=training), Models.D3QN(training=training)]
        with pytest.warns(UserWarning) if training else _nullcontext():
            env = envmod.Environment(brains=brains, max_agents=100, update_interval=update_interval, print_results=False, training=training,
                                     n_worlds=n_worlds, rng="philox", **kw)
        return env
    return make


class _nullcontext:
    def __enter__(self): return self
    def __exit__(self, *a): return False


def test_run_cuts_chunks_at_tracker_boundaries_and_builds_the_brains_own_schedule(fake_env):
    env = fake_env(update_interval=25)
    env.reset()
    assert env.worlds.resets[0] == ("families",) and env.worlds.resets[1] == ("load", 0)   # replicas on the device, world 0 from np.random
    env.run(0, 131)                                                                          # trainer(n_episodes=130)
    runs = [c for c in env.worlds.calls if c[0] == "run"]
    assert [c[1] for c in runs] == [26, 25, 25, 25, 25, 5]                                   # ... ending at episodes 25, 50, 75, 100, 125, 130
    assert [c[5] for c in runs] == [1, 0, 0, 0, 0, 0]                                        # episode 0 stays out of the running sums
    assert all(c[2] == -1 for c in runs)                                                     # no refill rule in trainer()
    # the schedule is what the reference's brains do: epsilon x 0.99 per new episode (D3QN.py:84-89), float32 rows, one per tick
    want, e = [], 0.9
    for n_epi in range(131):
        if n_epi > 0 and e > 0.05:
            e *= 0.99
        want.append(e)
    got = np.concatenate([c[4][:, 0] for c in runs])
    assert np.array_equal(got, np.array(want, np.float32)) and np.array_equal(got, np.concatenate([c[4][:, 1] for c in runs]))
    assert env.brains[0].epsilon == want[-1] and runs[-1][6] == [want[-1]] * 2                # brains bound with their current epsilon
    # five closed intervals; the stand-in counted one valid value per counted tick and world, so every aggregate is exactly 1
    res = env.tracker.results
    assert len(res["Avg Number of Populations"]) == 5 and res["Avg Population Size"][0] == [1.0] * 5
    # ... each closing zeroes the sums, and so does episode 0 (Tracker.update_results(n_epi=0), tracker.py:279-282) before its launch
    assert sum(1 for c in env.worlds.calls if c[0] == "reset_tracking") == 6 and env.worlds.calls.index(("reset_tracking",)) < env.worlds.calls.index(runs[0])
    # the weights go to the device ONCE; later chunks only move the exploration rates
    assert sum(1 for c in env.worlds.calls if c[0] == "set_brains") == 1


def test_the_next_chunk_is_queued_before_the_host_reads_a_closed_interval(fake_env, capsys):
    """Tracker intervals close in two halves (Helpers/tracker.py): sums copied out behind the chunk that ends the interval, read on the
    host only after the NEXT chunk has been queued; `results` and the end of run() resolve what is pending; the error flag that
    travelled with the sums raises where they are read."""
    from reinlife_amd import _lib
    env = fake_env(update_interval=10)
    w = env.worlds
    w.err = torch.zeros(4, dtype=torch.int32)

    class _Copies:
        def __init__(self, arrays): self.arrays = arrays
        def wait(self):
            w.calls.append(("wait",))
            return self.arrays

    def readback(tensors):
        w.calls.append(("readback",))
        return _Copies([t.clone().numpy() for t in tensors])

    def raise_on_error_flag(e):
        if e[0]:
            raise _lib.ReinLifeHipError("device error flag: code %d" % e[0])
    w.readback, w.raise_on_error_flag = readback, raise_on_error_flag
    env.tracker.print_results = True
    env.reset()
    env.run(0, 31)
    seq = [c[0] for c in w.calls if c[0] in ("run", "readback", "wait")]
    assert seq == ["run", "readback", "run", "wait", "readback", "run", "wait", "readback", "wait"]   # the last close: resolved at the end of run()
    assert capsys.readouterr().out.count("| Gene |") == 3 and len(env.tracker.results["Avg Number of Populations"]) == 3
    # a piece that ends on a boundary leaves nothing pending either; reading `results` never sees a half-closed interval
    env.run(31, 10)
    assert env.tracker._pending is None and len(env.tracker._results["Avg Number of Populations"]) == 4
    env.tracker.update_results(None, 50, defer=True)
    assert env.tracker._pending is not None and len(env.tracker.results["Avg Number of Populations"]) == 5 and env.tracker._pending is None
    # the tick-by-tick path closes both halves at once
    env.tracker.update_results(None, 60)
    assert env.tracker._pending is None and len(env.tracker._results["Avg Number of Populations"]) == 6
    # a device error surfaces with the interval's read-back
    w.err[0] = 3
    env.tracker.update_results(None, 70, defer=True)
    with pytest.raises(_lib.ReinLifeHipError):
        env.tracker.resolve()


def test_run_in_arbitrary_pieces_and_constant_epsilon(fake_env):
    env = fake_env(update_interval=20)
    env.reset()
    for n_epi, k in ((0, 1), (1, 7), (8, 30), (38, 23)):
        env.run(n_epi, k)
    assert [c[1] for c in env.worlds.calls if c[0] == "run"] == [1, 7, 13, 17, 3, 20]
    assert len(env.tracker.results["Avg Number of Populations"]) == 3
    # inference (training=False): epsilon 0 throughout -> no schedule is uploaded, no Tracker, chunks only bounded by max_chunk
    env = fake_env(training=False)
    env.reset()
    env.run(0, 10_000, max_chunk=4096)
    runs = [c for c in env.worlds.calls if c[0] == "run"]
    assert [c[1] for c in runs] == [4096, 4096, 1808] and all(c[4] is None and c[5] == 0 for c in runs)
    assert not env.worlds.tracking


def test_benchmark_worlds_through_the_api(fake_env):
    env = fake_env(synthetic_agents=100, refill_below=70)
    env.reset()
    assert env.worlds.resets == [("synthetic", 100)]
    env.run(0, 3)
    assert env.worlds.calls[-1][2:4] == (70, 100)
    from reinlife_amd.World import environment as envmod
    with pytest.raises(ValueError):
        envmod.Environment(brains=env.brains, n_worlds=1, refill_below=70, synthetic_agents=100)    # rng="reference" draws on the host
    with pytest.raises(ValueError):
        envmod.Environment(brains=env.brains, n_worlds=2, rng="philox", refill_below=70)


def test_host_mirrors_are_built_on_first_read_only(fake_env):
    env = fake_env()
    assert env.agents == [] and env.grid is None          # before reset(): nothing to show, nothing touched
    env.reset()
    env.run(0, 5)
    assert env._mirrors == {} and env._grid is None
    assert env.grid.shape == (30, 30) and env.max_gene == 2 and env._mirrors == {}


def test_run_opts_struct_matches_the_header():
    from reinlife_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "reinlife_hip.h")).read()
    end = hdr.index("} rl_run_opts;")
    body = re.sub(r"/\*.*?\*/", "", hdr[hdr.rindex("typedef struct {", 0, end) + len("typedef struct {"):end], flags=re.S)
    names = [re.findall(r"[A-Za-z_][A-Za-z_0-9]*", part)[-1] for decl in body.split(";") if decl.strip() for part in decl.split(",")]
    assert [n for n, _ in _lib.RunOpts._fields_] == names
    assert C.sizeof(_lib.RunOpts) == 48 and _lib.RunOpts.eps_schedule.offset == 16 and _lib.RunOpts.trk_skip_ticks.offset == 24 and _lib.RunOpts.eps_schedule_on_host.offset == 28
    assert _lib.RunOpts.replays.offset == 32 and _lib.RunOpts.policy_out.offset == 40


def test_epsilon_schedules_equal_the_brains_own_update_rule():
    """epsilon_schedule(n_epi, k) (one call per brain and chunk: Environment.run) == k calls of update_epsilon (D3QN.py:84-89, DQN.py:67-69 of the
    reference): the same float64 values bit for bit, the brain left in the same state -- for training and inference brains, chunks of any
    shape, repeated and overlapping episode ranges."""
    from reinlife_amd import Models
    makers = (lambda: Models.PERD3QN(), lambda: Models.D3QN(training=False), lambda: Models.DQN(max_epi=1000),
              lambda: Models.DQN(max_epi=200, training=False), lambda: Models.PPO())
    for mk in makers:
        for pieces in ([(0, 700)], [(0, 1), (1, 7), (8, 30), (38, 400), (438, 100)], [(5, 20), (25, 500)], [(0, 21)], [(3, 3), (3, 5), (2, 9)]):
            a, b = mk(), mk()
            for n0, k in pieces:
                want = []
                for t in range(k):
                    a.update_epsilon(n0 + t)
                    want.append(getattr(a, "epsilon", 0.0))
                got = b.epsilon_schedule(n0, k)
                assert np.array_equal(np.array(want, np.float64), got), (type(a).__name__, pieces, n0)
                assert getattr(a, "epsilon", 0.0) == getattr(b, "epsilon", 0.0) and getattr(a, "n_epi", 0) == getattr(b, "n_epi", 0)
