"""GPU, round 3: the multi-tick launch as a TRAINING loop (rl_run_ex: per-tick epsilon schedule + Tracker statistics inside the
launch), trainer() / tester() / Environment.run on top of it, parity at the benched size (256 worlds), SURVEY C2 (step-only, 20
seeds x 200 ticks against the oracle), rl_reset_families, and the host mirrors' action semantics."""
import copy
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from test_hip_round2 import _cmp_rows, _cmp_state, _need_run, _run_pair, _same_device_state, _weights  # noqa: E402


def _same_tracker(a, b, tag):
    for name in ("trk_tick", "trk_sum", "trk_cnt", "trk_pop"):
        x, y = getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy()
        assert np.array_equal(x, y, equal_nan=True), (tag, name, np.argwhere(x != y)[:4])


# ---------------------------------------------------------------------------------------------------------------------
# rl_run_ex: epsilon schedule + Tracker accumulators in the launch
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("static", [True, False], ids=["static", "nonstatic"])
@pytest.mark.parametrize("block", [None, 256, 1024], ids=["T512", "T256", "T1024"])
def test_training_launch_equals_the_two_launch_loop(static, block, hip_option):
    """rl_run_ex with a per-tick epsilon schedule and the Tracker accumulators == n x (set epsilons, rl_policy_act, rl_tick_refill
    with the Tracker outputs): worlds, observations, actions and trk_tick / trk_sum / trk_cnt / trk_pop, bit for bit, over launches
    of 1 / 3 / 20 / 37 ticks, with the caller zeroing the running sums between some of them (interval boundaries)."""
    if block:
        hip_option("world_block", block)
    (fused, loop), *_ = _run_pair(14, static, 4242)
    for dw in (fused, loop):
        dw.enable_tracking(True)
    _need_run(fused, block)
    rng = np.random.RandomState(7)
    done = 0
    for chunk in (1, 3, 20, 37, 1, 20):
        eps = rng.uniform(0.0, 0.6, size=(chunk, 2)).astype(np.float32)
        eps[:, 0] *= (rng.uniform(size=chunk) < 0.5)          # brain 0 is greedy in about half of the ticks
        fused.run(chunk, 70, 100, eps_schedule=eps)
        for t in range(chunk):
            loop._set_epsilons(eps[t].tolist())
            loop.act(); loop.tick_refill(70, 100)
        done += chunk
        fused.check_error_flag(); loop.check_error_flag()
        tag = "after %d ticks" % done
        _same_device_state(fused, loop, tag)
        acted = fused.n_acted.cpu().numpy()
        _cmp_rows(fused.actions.cpu().numpy(), loop.actions.cpu().numpy(), acted, tag + " actions")
        _same_tracker(fused, loop, tag)
        assert int(fused.acted_total.item()) == int(loop.acted_total.item())
        assert int(fused.trk_cnt.sum().item()) > 0
        if chunk == 20 and done < 60:
            fused.reset_tracking(); loop.reset_tracking()
    assert int(fused.refill_count.item()) > 0


def test_training_launch_tracker_matches_the_oracle():
    """The Tracker values of rl_run_ex against the ORACLE's (tracker.py:178-266 restated sequentially), tick by tick."""
    from oracle import oracle as orc
    (fused, _), wts, names, eps, cfg = _run_pair(10, True, 99)
    fused.enable_tracking(True)
    ow = orc.OracleWorlds(n_worlds=10, seed=99, world_base=3, **cfg)
    ow.reset_synthetic(100)
    for t in range(30):
        fused.run(1, 70, 100)
        ow.step(fused.actions.cpu().numpy().copy())
        assert np.array_equal(fused.trk_tick.cpu().numpy(), ow.trk_tick), t
        assert np.array_equal(fused.trk_sum.cpu().numpy(), ow.trk_sum) and np.array_equal(fused.trk_cnt.cpu().numpy(), ow.trk_cnt), t
        assert np.array_equal(fused.trk_pop.cpu().numpy(), ow.trk_pop), t
        ow.update(); ow.refill(70, 100)
        _cmp_state(fused, ow, "tick %d" % t)


# ---------------------------------------------------------------------------------------------------------------------
# trainer() / tester() / Environment.run on the multi-tick launch
# ---------------------------------------------------------------------------------------------------------------------
def _brains(seed, training=True):
    import torch
    from reinlife_amd import Models
    torch.manual_seed(seed)
    return [Models.PERD3QN(training=training), Models.D3QN(training=training)]


@pytest.mark.parametrize("static", [True, False], ids=["static", "nonstatic"])
def test_trainer_fused_loop_equals_the_tick_by_tick_loop(static):
    """trainer(..., fused=True) -- whole Tracker intervals per launch -- against fused=False (act / step / update_env launches and a
    host round trip per tick, the round-2 loop): identical Tracker.results, identical final worlds, identical brain epsilons.
    training=True: the brains' epsilon decays per episode (x 0.99), i.e. the launch runs on a schedule."""
    from reinlife_amd.Helpers.trainer import trainer
    proto = _brains(5)
    envs = []
    for fused in (True, False):
        brains = copy.deepcopy(proto)
        np.random.seed(1234)   # world 0 is built by Environment.reset from the process-global np.random, like the reference's
        with pytest.warns(UserWarning):
            env = trainer(brains, n_episodes=130, update_interval=25, print_results=False, static_families=static, save=False,
                          n_worlds=6, seed=77, fused=fused)
        envs.append(env)
    a, b = envs
    assert a.tracker.results == b.tracker.results or _results_equal(a.tracker.results, b.tracker.results)
    assert len(a.tracker.results["Avg Number of Populations"]) == 5            # 130 // 25 closed intervals
    _same_device_state(a.worlds, b.worlds, "final worlds")
    assert [br.epsilon for br in a.brains] == [br.epsilon for br in b.brains] and a.brains[0].epsilon < 0.9 * 0.99 ** 129
    assert a.loop_seconds > 0 and b.loop_seconds > 0
    # the host view is built on demand and agrees
    assert [(v.i, v.j, v.gene, v.health, v.age) for v in a.agents] == [(v.i, v.j, v.gene, v.health, v.age) for v in b.agents]
    assert np.array_equal(a.grid, b.grid) and a.max_gene == b.max_gene


def _results_equal(x, y):
    """Tracker.results with nan-aware float comparison (an interval without a valid tick aggregates to nan, like np.mean([]))."""
    if isinstance(x, dict):
        return set(x) == set(y) and all(_results_equal(x[k], y[k]) for k in x)
    return np.array_equal(np.asarray(x, np.float64), np.asarray(y, np.float64), equal_nan=True)


def test_environment_run_is_lazy_and_splits_at_tracker_boundaries():
    """Environment.run: arbitrary (n_epi, n_ticks) calls close the same Tracker intervals as the per-tick loop; nothing is copied to the
    host until env.agents / env.grid are read."""
    from reinlife_amd.World.environment import Environment
    proto = _brains(9)
    res = []
    for pieces in ([(0, 61)], [(0, 1), (1, 7), (8, 30), (38, 23)]):
        np.random.seed(99)
        with pytest.warns(UserWarning):
            env = Environment(width=30, height=30, brains=copy.deepcopy(proto), max_agents=100, update_interval=20, print_results=False,
                              n_worlds=4, seed=3, rng="philox")
        env.reset()
        for n_epi, k in pieces:
            env.run(n_epi, k)
            assert env._mirrors == {} and env._grid is None
        res.append(env)
    assert _results_equal(res[0].tracker.results, res[1].tracker.results)
    assert len(res[0].tracker.results["Avg Population Size"][0]) == 3
    _same_device_state(res[0].worlds, res[1].worlds, "pieces")
    assert len(res[0].agents) == int(res[0].worlds.s["n_agents"][0].item()) and res[0]._mirrors != {}


def test_tester_runs_one_launch_per_frame():
    from reinlife_amd.Helpers.tester import tester
    frames = []
    env = tester(_brains(2, training=False), n_steps=5, n_worlds=3, seed=1, on_frame=lambda e: frames.append(e.frame.copy()))
    assert len(frames) == 5 and frames[0].shape == (30 * 24, 30 * 24, 3) and env.loop_seconds is None
    assert int(env.worlds.s["tick"][0].item()) == 5


def test_replica_mirror_holds_the_policys_actions_after_act():
    """env.act(); env.agents_of(w) must show the actions the policy chose for this tick, and overriding ONE of them
    must leave the others as chosen."""
    from reinlife_amd.World.environment import Environment
    with pytest.warns(UserWarning):
        env = Environment(width=30, height=30, brains=_brains(4), max_agents=100, print_results=False, n_worlds=3, seed=8, rng="philox")
    env.reset()
    env.run(0, 12)                      # a few agents per world by now
    early = env.agents_of(2)            # built BEFORE act(): must be refreshed by it
    env.act(12)
    chosen = env.worlds.actions.cpu().numpy().copy()
    for w in (1, 2):
        views = env.agents_of(w)
        assert [v.action for v in views] == chosen[w, : len(views)].tolist(), w
    assert early is env.agents_of(2)
    views = env.agents_of(1)
    other = (chosen[1, 0] + 3) % 8
    views[0].action = int(other)
    env.step()
    taken = env.worlds.s["a_action"].cpu().numpy()
    src = env.worlds.src1.cpu().numpy()
    n1 = int(env.worlds.s["n_agents"][1].item())
    want = chosen[1].copy(); want[0] = other
    assert np.array_equal(taken[1, :n1], want[src[1, :n1]])          # every other agent acted as the policy chose
    n2 = int(env.worlds.s["n_agents"][2].item())
    assert np.array_equal(taken[2, :n2], chosen[2][src[2, :n2]])


# ---------------------------------------------------------------------------------------------------------------------
# Environment.reset on the device
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_brains,shape", [(2, (30, 30)), (5, (12, 9)), (8, (30, 20))])
def test_reset_families_matches_the_oracle(n_brains, shape):
    from oracle import oracle as orc
    from reinlife_amd.worlds import DeviceWorlds
    cfg = dict(width=shape[0], height=shape[1], max_agents=100, n_brains=n_brains, static_families=True)
    dw = DeviceWorlds(n_worlds=40, seed=12, world_base=9, **cfg)
    ow = orc.OracleWorlds(n_worlds=40, seed=12, world_base=9, **cfg)
    dw.reset_families(); ow.reset_families()
    dw.check_error_flag()
    _cmp_state(dw, ow, "reset_families")
    _cmp_rows(dw.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "obs")
    genes = dw.s["a_gene"].cpu().numpy()[:, :n_brains]
    assert (dw.s["n_agents"].cpu().numpy() == n_brains).all() and (np.sort(genes, axis=1) == np.arange(n_brains)).all()
    assert len({tuple(g) for g in genes}) > 1          # which brain lands where differs from world to world


# ---------------------------------------------------------------------------------------------------------------------
# parity at the benched size (BASELINE configs[3]: 256 worlds, two greedy PERD3QN brains, refill below 70) and SURVEY C2
# ---------------------------------------------------------------------------------------------------------------------
def _bench_worlds(seed=20260928):
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    return bench.make_worlds(argparse.Namespace(worlds=256, workload="c4", seed=seed), 0, "cuda:0")


def test_benched_size_multi_tick_launch_tracks_the_oracle_for_300_ticks():
    """The instantiation bench.py times (k_run<512, fixed> on 256 workgroups) against the oracle fed the chosen actions: 300 launches
    of one tick; integer state every tick, both observation passes and every per-agent output every 10 ticks.  Round 5: the POLICY half
    at that size too (tests/policy_check.py) -- every 10th tick the chosen actions of all ~22k agents against the selection rule on an
    f32 forward of the ORACLE's rows, and every other 10th tick a launch that also writes its Q values (policy_out): Q within 1e-5,
    actions exactly the rule on them."""
    import bench
    from oracle import oracle as orc
    from policy_check import PolicyCheck
    dw = _bench_worlds()
    ow = orc.OracleWorlds(n_worlds=256, seed=20260928, width=30, height=30, max_agents=100, n_brains=2, static_families=True)
    ow.reset_synthetic(100)
    assert dw.run_supported()
    pc = PolicyCheck(["PERD3QN", "PERD3QN"], [bench.brain_weights("PERD3QN", 100 + k) for k in range(2)], [0.0, 0.0])
    steps = 0
    for t in range(300):
        n0 = ow.s["n_agents"].copy()
        with_q, with_a = t % 10 == 4, t % 10 == 9
        if with_q or with_a:
            pc.before(ow)
        dw.run(1, 70, 100, want_q=with_q)
        acts = dw.actions.cpu().numpy()
        if with_q or with_a:
            pc.after(acts, dw.out_q.cpu().numpy() if with_q else None, "tick %d" % t)
        ow.step(acts)
        full = t % 10 == 9
        if full:
            n1 = ow.s["n_agents"].copy()
            _cmp_rows(dw.obs_state_prime().cpu().numpy(), ow.obs1, n1, "tick %d obs1" % t)
            _cmp_rows(dw.reward.cpu().numpy(), ow.reward, n1, "tick %d reward" % t)
            _cmp_rows(dw.done.cpu().numpy(), ow.done, n1, "tick %d done" % t)
            _cmp_rows(dw.src1.cpu().numpy(), ow.src1, n1, "tick %d src1" % t)
        ow.update(); ow.refill(70, 100)
        assert np.array_equal(dw.n_acted.cpu().numpy(), n0)
        steps += int(n0.sum())
        for key in ("cell_type", "n_agents", "tick", "epoch", "next_uid"):
            assert np.array_equal(dw.s[key].cpu().numpy().reshape(ow.s[key].shape), ow.s[key]), (t, key)
        if full:
            dw.check_error_flag()
            _cmp_state(dw, ow, "tick %d" % t)
            _cmp_rows(dw.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)
    assert steps > 6_000_000 and int(dw.acted_total.item()) == steps and int(dw.refill_count.item()) > 500
    assert pc.rows > 1_200_000 and pc.q_rows > 600_000 and pc.max_dq < 1e-5


def test_benched_size_one_launch_of_300_ticks_equals_300_launches_of_one():
    a, b = _bench_worlds(), _bench_worlds()
    a.run(300, 70, 100)
    for _ in range(300):
        b.run(1, 70, 100)
    a.check_error_flag(); b.check_error_flag()
    _same_device_state(a, b, "300 ticks")
    assert int(a.acted_total.item()) == int(b.acted_total.item()) and int(a.refill_count.item()) == int(b.refill_count.item()) > 500
    acted = a.n_acted.cpu().numpy()
    _cmp_rows(a.actions.cpu().numpy(), b.actions.cpu().numpy(), acted, "actions")
    for name in ("reward", "done", "src1", "src2"):
        assert np.array_equal(getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy()), name
    assert np.array_equal(a.obs_state_prime().cpu().numpy(), b.obs_state_prime().cpu().numpy())


def test_c2_step_only_twenty_seeds_two_hundred_ticks_against_the_oracle():
    """30x30, 100 agents, uniformly random actions, Environment.step (and update_env, so that the world goes on) against
    the CPU restatement: 20 seeds x 200 ticks, integer state bit-exact after EVERY step and update, rewards / observations (float32
    identical) every 20 ticks and at the end."""
    from oracle import oracle as orc
    from reinlife_amd.worlds import DeviceWorlds
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=True, limit_reproduction=False, incentivize_killing=True)
    agent_steps = 0
    for seed in range(20):
        dw = DeviceWorlds(n_worlds=1, seed=1000 + seed, **cfg)
        ow = orc.OracleWorlds(n_worlds=1, seed=1000 + seed, **cfg)
        dw.reset_synthetic(100); ow.reset_synthetic(100)
        rng = np.random.RandomState(seed)
        for t in range(200):
            acts = rng.randint(0, 8, size=(1, dw.cap)).astype(np.int8)
            agent_steps += int(ow.s["n_agents"][0])
            dw.step(acts); ow.step(acts)
            full = t % 20 == 19 or t == 199
            keys = list(dw.s) if full else ("cell_type", "n_agents", "a_health", "a_age", "a_flags", "a_i", "a_j")
            n1 = ow.s["n_agents"]
            for key in keys:
                got, want = dw.s[key].cpu().numpy(), ow.s[key]
                if key.startswith("a_"):
                    assert np.array_equal(got[0, : n1[0]], want[0, : n1[0]]), (seed, t, "step", key)
                else:
                    assert np.array_equal(got.reshape(want.shape), want), (seed, t, "step", key)
            if full:
                _cmp_rows(dw.obs_state_prime().cpu().numpy(), ow.obs1, n1, "seed %d tick %d obs1" % (seed, t))
                _cmp_rows(dw.reward.cpu().numpy(), ow.reward, n1, "reward")
                _cmp_rows(dw.done.cpu().numpy(), ow.done, n1, "done")
            dw.update(); ow.update()
            n2 = ow.s["n_agents"]
            for key in keys:
                got, want = dw.s[key].cpu().numpy(), ow.s[key]
                if key.startswith("a_"):
                    assert np.array_equal(got[0, : n2[0]], want[0, : n2[0]]), (seed, t, "update", key)
                else:
                    assert np.array_equal(got.reshape(want.shape), want), (seed, t, "update", key)
            if full:
                _cmp_rows(dw.obs_state().cpu().numpy(), ow.obs2, n2, "seed %d tick %d obs2" % (seed, t))
        dw.check_error_flag()
    assert agent_steps > 20 * 200 * 10


# ---------------------------------------------------------------------------------------------------------------------
# the multi-tick launch for brains of ANY kind (DQN, PPO, mixed): two waves per tile, the tile code picked per tile
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind_name", ["DQN", "PPO", "D3QN", "PERD3QN"])
def test_pair_tiles_against_the_oracle_forward(kind_name, hip_option):
    """policy_variant "pair" (k_policy_pair: the two-waves-per-tile code rl_run's policy half runs, for every kind) against the oracle's
    f32 forward (1e-5) for ragged row counts; for the dueling kinds the pair IS the one-wave tile bit for bit."""
    import torch
    from oracle import oracle as orc
    from reinlife_amd import _lib
    from reinlife_amd.worlds import pack_brain_weights, policy_forward
    kind = _lib.KIND_BY_METHOD[kind_name]
    w = _weights(kind_name, 31)
    packed = pack_brain_weights(kind, w)
    g = torch.Generator(device="cuda:0").manual_seed(17)
    for n in (1, 31, 33, 257, 4099):
        obs = (torch.randn(n + 1, 153, device="cuda:0", generator=g) * torch.rand(n + 1, 1, device="cuda:0", generator=g) * 3)[:n].contiguous()
        hip_option("policy_variant", "pair")
        out = torch.full((n, 8), float("nan"), device="cuda:0")
        policy_forward(kind, packed, obs, out)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        want = orc.policy_forward(orc.KIND_BY_NAME[kind_name], w, obs.cpu().numpy())
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-5, err_msg="%s n=%d" % (kind_name, n))
        if kind_name in ("D3QN", "PERD3QN"):
            hip_option("policy_variant", "wave")
            ref = torch.full((n, 8), float("nan"), device="cuda:0")
            policy_forward(kind, packed, obs, ref)
            torch.cuda.synchronize()
            assert np.array_equal(got, ref.cpu().numpy()), n
        if kind_name == "PPO":
            np.testing.assert_allclose(got.sum(1), 1.0, rtol=0, atol=1e-5)


def _kind_pair(names, eps, R, static, seed, **extra):
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    cfg = dict(width=30, height=30, max_agents=100, n_brains=len(names), static_families=static, limit_reproduction=False, incentivize_killing=True)
    cfg.update(extra)
    wts = [_weights(n, 300 + k) for k, n in enumerate(names)]
    out = []
    for _ in range(2):
        dw = DeviceWorlds(n_worlds=R, seed=seed, world_base=11, **cfg)
        dw.set_brains([(_lib.KIND_BY_METHOD[n], e, pack_brain_weights(_lib.KIND_BY_METHOD[n], w)) for n, e, w in zip(names, eps, wts)])
        dw.reset_synthetic(cfg["max_agents"])
        out.append(dw)
    return out, wts, cfg


KIND_SETS = [(("DQN",), (0.0,), True), (("DQN", "DQN"), (0.0, 0.2), True), (("PPO", "PERD3QN"), (0.0, 0.05), False), (("PPO", "DQN", "D3QN"), (0.0, 0.1, 0.0), True),
             (("PPO", "PPO"), (0.0, 0.0), True), (("DQN", "PERD3QN", "PPO", "D3QN"), (0.3, 0.0, 0.0, 0.2), False)]


@pytest.mark.parametrize("names,eps,static", KIND_SETS, ids=["+".join(k[0]) for k in KIND_SETS])
def test_multi_tick_launch_any_brain_kinds_equals_the_two_launch_loop(names, eps, static):
    """rl_run with DQN / PPO / mixed-kind brains (the kKindAll kernel: per tile two waves running its kind's code, the waves dealt over
    the SIMDs by cost) == n x (rl_policy_act + rl_tick_refill) with the stand-alone launch (which runs the same tiles by default):
    worlds, observations, actions (PPO: the sampled ones), outputs, counters, for launches of 1 / 2 / 7 / 30 / 45 ticks with refills."""
    (fused, loop), wts, cfg = _kind_pair(names, eps, 14, static, 2026)
    assert fused.run_supported()
    done = 0
    for chunk in (1, 2, 7, 30, 45):
        fused.run(chunk, 70, 100)
        for _ in range(chunk):
            loop.act(); loop.tick_refill(70, 100)
        done += chunk
        fused.check_error_flag(); loop.check_error_flag()
        tag = "%s after %d ticks" % ("+".join(names), done)
        _same_device_state(fused, loop, tag)
        acted = fused.n_acted.cpu().numpy()
        _cmp_rows(fused.actions.cpu().numpy(), loop.actions.cpu().numpy(), acted, tag + " actions")
        assert int(fused.acted_total.item()) == int(loop.acted_total.item()) and int(fused.refill_count.item()) == int(loop.refill_count.item())
        for name in ("reward", "done", "src1", "src2"):
            assert np.array_equal(getattr(fused, name).cpu().numpy(), getattr(loop, name).cpu().numpy()), (tag, name)
        assert np.array_equal(fused.obs_state_prime().cpu().numpy(), loop.obs_state_prime().cpu().numpy()), tag + " obs1"
    assert int(fused.refill_count.item()) > 0


def test_multi_tick_launch_mixed_kinds_tracks_the_oracle_with_tracker_and_schedule():
    """PPO + PERD3QN, non-static families (BASELINE configs[4] per GPU), as a TRAINING launch (epsilon schedule + Tracker): run(1) per
    tick against the oracle fed the chosen actions; every 8th tick the actions themselves against the oracle's selection rule applied
    to the oracle's f32 forward (PPO: where the inverse-CDF sample is not within 1e-5 of a bin edge; PERD3QN: where the top two Q
    values are 1e-5 apart)."""
    from oracle import oracle as orc
    names, eps0 = ("PPO", "PERD3QN"), (0.0, 0.0)
    (fused, _), wts, cfg = _kind_pair(names, eps0, 12, False, 515)
    fused.enable_tracking(True)
    ow = orc.OracleWorlds(n_worlds=12, seed=515, world_base=11, **cfg)
    ow.reset_synthetic(100)
    checked = 0
    for t in range(48):
        n = ow.s["n_agents"].copy()
        e1 = 0.0 if t % 8 == 0 else 0.3
        fused.run(1, 70, 100, eps_schedule=np.array([[0.0, e1]], np.float32))
        acts = fused.actions.cpu().numpy().copy()
        if t % 8 == 0:
            for b, name in enumerate(names):
                ws, ks = np.nonzero((np.arange(fused.cap)[None, :] < n[:, None]) & (ow.s["a_brain"] == b))
                q = orc.policy_forward(orc.KIND_BY_NAME[name], wts[b], ow.obs2[ws, ks])
                want = orc.select_actions(ow.cfg, orc.KIND_BY_NAME[name], q, ws, ks, ow.s["tick"], ow.s["epoch"], 0.0)
                if name == "PPO":   # a sample that lies within 1e-5 of a CDF edge may fall either way
                    u = np.array([(orc.philox(515, int(ow.s["epoch"][w]), 11 + int(w), int(ow.s["tick"][w]), 5, int(k))[0] >> 8) / 16777216.0
                                  for w, k in zip(ws, ks)])
                    clear = np.abs(np.cumsum(q, axis=1) - u[:, None]).min(1) > 1e-5
                else:
                    srt = np.sort(q, axis=1)
                    clear = srt[:, -1] - srt[:, -2] > 1e-5
                assert np.array_equal(acts[ws, ks][clear], want[clear]), (t, name)
                checked += int(clear.sum())
        ow.step(acts)
        assert np.array_equal(fused.trk_tick.cpu().numpy(), ow.trk_tick), t
        ow.update(); ow.refill(70, 100)
        fused.check_error_flag()
        _cmp_state(fused, ow, "tick %d" % t)
        _cmp_rows(fused.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)
    assert checked > 1000


def test_multi_tick_launch_many_tiles_take_several_rounds():
    """Eight brains of all four kinds: every world has more than four tiles, so the kKindAll kernel runs the tiles in rounds (fixed wave
    pairs, 32 KB exchange buffer per slot, rows from memory) -- the same tiles, the same bits as the stand-alone pair launch; plus a
    crowded configuration (200 agents: slot capacity 448, a smaller mirror)."""
    names = ("DQN", "PPO", "D3QN", "PERD3QN", "PPO", "DQN", "PERD3QN", "D3QN")
    eps = (0.1, 0.0, 0.0, 0.2, 0.0, 0.0, 0.0, 0.0)
    for extra, thr in ((dict(), 70), (dict(max_agents=200), 120)):
        (fused, loop), wts, cfg = _kind_pair(names, eps, 10, True, 77, **extra)
        assert fused.run_supported()
        for chunk in (1, 5, 24):
            fused.run(chunk, thr, cfg["max_agents"])
            for _ in range(chunk):
                loop.act(); loop.tick_refill(thr, cfg["max_agents"])
            fused.check_error_flag(); loop.check_error_flag()
            _same_device_state(fused, loop, "8 brains, chunk %d" % chunk)
            _cmp_rows(fused.actions.cpu().numpy(), loop.actions.cpu().numpy(), fused.n_acted.cpu().numpy(), "actions")


# ---------------------------------------------------------------------------------------------------------------------
# transition capture inside the multi-tick launch
# ---------------------------------------------------------------------------------------------------------------------
def _ring_rows(dw, b, lo, hi):
    """Transitions lo..hi-1 of brain b's ring as comparable byte rows (state | state_prime | reward | prob | action | done | age)."""
    r = dw.replays[b]
    cap = r["state"].shape[0]
    idx = np.arange(lo, hi) % cap
    cols = [r["state"].cpu().numpy()[idx], r["state_prime"].cpu().numpy()[idx], r["reward"].cpu().numpy()[idx, None]]
    if r["prob"] is not None:
        cols.append(r["prob"].cpu().numpy()[idx, None])
    cols += [r[k].cpu().numpy()[idx, None].astype(np.float32) for k in ("action", "done", "age")]
    rows = np.ascontiguousarray(np.concatenate(cols, axis=1))
    return rows[np.lexsort(rows.T[::-1])]


@pytest.mark.parametrize("names,eps,static,block", [(("PERD3QN", "D3QN"), (0.0, 0.15), True, None), (("PPO", "PERD3QN"), (0.0, 0.05), False, None),
                                                    (("PERD3QN", "D3QN"), (0.0, 0.15), False, 256), (("D3QN", "PERD3QN"), (0.1, 0.0), True, 1024)],
                         ids=["dueling", "PPO+PERD3QN", "dueling-T256", "dueling-T1024"])
def test_capture_inside_the_multi_tick_launch(names, eps, static, block, hip_option):
    """DeviceWorlds.enable_capture + run(k): the launch appends every tick's transitions (state the policy read, action, reward,
    state_prime, done, age, the taken action's policy output) to the brains' rings -- the same transitions rl_capture_transitions stores
    after each stand-alone tick (trainer.py:95-96 / entities.py:194-208: agents of the post-step list with age > 1).  Within a tick the
    worlds' transitions interleave by atomics in both paths, so a tick's transitions are compared as sets; counts and worlds exactly."""
    if block:
        hip_option("world_block", block)
    (fused, loop), wts, cfg = _kind_pair(names, eps, 9, static, 31)
    for dw in (fused, loop):
        dw.enable_capture(capacity=40_000, with_prob=True)
    _need_run(fused, block)
    seen = [0] * len(names)
    for chunk in (1, 1, 3, 12):
        fused.run(chunk, 70, 100)
        for _ in range(chunk):
            loop.act(want_q=True); loop.tick_refill(70, 100); loop.capture_transitions(with_policy_out=True)
        fused.check_error_flag(); loop.check_error_flag()
        _same_device_state(fused, loop, "capture chunk %d" % chunk)
        for b in range(len(names)):
            tot_f, tot_l = int(fused.replays[b]["count"].item()), int(loop.replays[b]["count"].item())
            assert tot_f == tot_l >= seen[b], (b, tot_f, tot_l)   # (nothing in the first tick: Agent.learn needs age > 1)
            assert np.array_equal(_ring_rows(fused, b, seen[b], tot_f), _ring_rows(loop, b, seen[b], tot_l)), (names[b], chunk)
            seen[b] = tot_f
    assert sum(seen) > 5000


# (the random-configuration draw of tools/fuzz_parity.py runs in tests/test_hip_round4.py: 64 cases)
