"""GPU: the Python mirror of the reference API (Environment / trainer / tester / Models) drives the HIP path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_environment_protocol_matches_oracle_with_per_agent_actions():
    """reset() under a numpy seed, per-agent `agent.action = ...`, step(), update_env(): world 0 tracks the oracle."""
    from oracle import oracle as orc
    from reinlife_amd import Environment, Models
    from reinlife_amd.World.environment import host_reset
    brains = [Models.PERD3QN(training=False), Models.DQN(training=False), Models.PPO()]
    env = Environment(width=30, height=30, brains=brains, max_agents=100, static_families=True, training=False, seed=5)
    assert env.action_space == 8 and env.observation_space == 153
    np.random.seed(11)
    env.reset()
    np.random.seed(11)
    snap = host_reset(30, 30, 3)
    ow = orc.OracleWorlds(1, 30, 30, 100, 3, True, False, True, seed=5)
    ow.load_world(0, snap)
    ow.observe()
    assert len(env.agents) == 3 and [a.gene for a in env.agents] == list(snap["gene"])
    assert np.array_equal(np.stack([a.state for a in env.agents]).astype(np.float32), ow.obs2[0, :3])
    rng = np.random.RandomState(0)
    for t in range(60):
        acts = np.zeros((1, ow.cap), np.int8)
        for k, agent in enumerate(env.agents):
            agent.action = int(rng.randint(0, 8))
            acts[0, k] = agent.action
        env.step()
        ow.step(acts)
        n1 = int(ow.s["n_agents"][0])
        assert len(env.agents) == n1
        assert [a.health for a in env.agents] == list(ow.s["a_health"][0, :n1])
        assert [a.dead for a in env.agents] == [bool(f & 1) for f in ow.s["a_flags"][0, :n1]]
        assert np.allclose([a.reward for a in env.agents], ow.reward[0, :n1], atol=0)
        assert [a.action for a in env.agents] == list(ow.s["a_action"][0, :n1])
        if n1:
            assert np.array_equal(np.stack([a.state_prime for a in env.agents]).astype(np.float32), ow.obs1[0, :n1])
        env.update_env(t)
        ow.update()
        n2 = int(ow.s["n_agents"][0])
        assert len(env.agents) == n2 and env.max_gene == int(ow.s["max_gene"][0])
        assert np.array_equal(env.grid.reshape(-1), ow.s["cell_type"][0])
        if n2:
            assert np.array_equal(np.stack([a.state for a in env.agents]).astype(np.float32), ow.obs2[0, :n2])


def test_trainer_and_tester_run_the_batched_loop():
    from reinlife_amd import Models, tester, trainer
    np.random.seed(3)
    env = trainer([Models.PERD3QN(), Models.PERD3QN()], n_episodes=40, width=30, height=30, max_agents=100,
                  static_families=True, save=False, print_results=False, n_worlds=4)
    assert env.worlds.s["tick"].cpu().numpy().tolist() == [41] * 4
    for a in env.agents:
        assert a.state.shape == (153,) and 0 <= a.gene < 2 and a.brain.method == "PERD3QN"
    env2 = tester([Models.PPO(), Models.DQN(training=False), Models.D3QN(training=False)], width=30, height=20, max_agents=150,
                  n_steps=25)
    assert env2.grid.shape == (20, 30) and int(env2.worlds.s["tick"][0].item()) == 25


def test_single_state_get_action_matches_batched_forward():
    from oracle import oracle as orc
    from reinlife_amd import Models
    b = Models.D3QN(training=False)
    rng = np.random.RandomState(1)
    states = rng.uniform(-1, 1, size=(5, 153))
    q = orc.policy_forward(orc.D3QN, b.state_dict_flat(), states.astype(np.float32))
    for s, qq in zip(states, q):
        srt = np.sort(qq)
        if srt[-1] - srt[-2] > 1e-4:
            assert b.get_action(s, 0) == int(qq.argmax())


def test_tracker_interval_aggregates_match_reference_rule():
    """Tracker.results from the in-kernel accumulators == mean over the interval of the reference's per-tick values that
    are > -1 (Helpers/tracker.py:279-282), episode 0 excluded; per-tick values come from the golden trace."""
    import golden_io
    from hip_backend import HipBackend
    from reinlife_amd.Helpers.tracker import Tracker, VARIABLES
    path = [p for p in golden_io.trace_files("trace_") if "natural_static" in p][0]
    tr = np.load(path)
    cfg, ticks = golden_io.trace_cfg(tr)
    hb = HipBackend(1, **cfg)
    hb.load_world(0, golden_io.initial_snapshot(tr))
    interval = 25
    trk = Tracker(update_interval=interval, print_results=False, nr_genes=cfg["n_brains"], static_families=True, worlds=hb.dw)
    for t in range(ticks):
        n0 = int(tr["n0"][t])
        acts = np.zeros((1, hb.cap), np.int8)
        acts[0, :n0] = tr["actions"][t][:n0]
        tape = hb.make_tape([golden_io.tick_tape(tr, t)])
        hb.step(acts, tape)
        trk.update_results(None, t)
        hb.update(tape)
    n_int = (ticks - 1) // interval
    assert len(trk.results["Avg Number of Populations"]) == n_int >= 3
    for k in range(n_int):
        win = slice(k * interval + 1, (k + 1) * interval + 1)
        for i, v in enumerate(VARIABLES[:-1]):
            for g in range(cfg["n_brains"]):
                vals = tr["trk_tick"][win, g, i]
                vals = vals[vals > -1]
                want = vals.mean() if len(vals) else float("nan")
                got = trk.results[v][g][k]
                assert (np.isnan(want) and np.isnan(got)) or abs(got - want) <= 1e-12 * max(1.0, abs(want)), (k, v, g, got, want)
        pv = tr["trk_pop"][win]
        assert abs(trk.results["Avg Number of Populations"][k] - pv[pv > -1].mean()) <= 1e-12
