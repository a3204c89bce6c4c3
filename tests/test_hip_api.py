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
    env = Environment(width=30, height=30, brains=brains, max_agents=100, static_families=True, training=False, seed=5,
                      rng="philox")
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
    import torch
    from reinlife_amd import Models, tester, trainer
    np.random.seed(3)
    torch.manual_seed(3)   # the brains' initial weights: tester()'s three agents starve at tick 20 unless they eat, and with unseeded weights
                           # they all did in 1 of 40 runs (the assertion on the last frame needs a living agent)
    env = trainer([Models.PERD3QN(), Models.PERD3QN()], n_episodes=40, width=30, height=30, max_agents=100,
                  static_families=True, save=False, print_results=False, n_worlds=4)
    assert env.worlds.s["tick"].cpu().numpy().tolist() == [41] * 4
    for a in env.agents:
        assert a.state.shape == (153,) and 0 <= a.gene < 2 and a.brain.method == "PERD3QN"
    frames = []
    env2 = tester([Models.PPO(), Models.DQN(training=False), Models.D3QN(training=False)], width=30, height=20, max_agents=150,
                  n_steps=25, on_frame=lambda e: frames.append(e.frame))
    assert env2.grid.shape == (20, 30) and int(env2.worlds.s["tick"][0].item()) == 25
    # tester.py:55,72: one frame after reset and one per tick; the last one shows the device world as it stands
    assert len(frames) == 25 and frames[-1].shape == (20 * 24, 30 * 24, 3) and frames[-1].dtype == np.uint8
    feed = env2.render_feed()
    assert len(feed.i) == len(env2.agents) > 0
    for a, i, j, g in zip(env2.agents, feed.i, feed.j, feed.gene):
        assert (a.i, a.j, a.gene) == (i, j, g)
        assert tuple(frames[-1][i * 24 + 12, j * 24 + 6]) == env2.viz.colors[g % 8]   # body colour, inside the border, off the eyes
    ii, jj = feed.cells(env2.entities.food)
    assert len(ii) and all(tuple(frames[-1][i * 24 + 12, j * 24 + 12]) == (255, 255, 255) for i, j in zip(ii, jj))


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


def _check_rings(dw, expected, seen):
    """expected: per brain list of transition dicts appended this tick (call order); seen: counts before."""
    for b, exp in enumerate(expected):
        r = dw.replays[b]
        total = int(r["count"].item())
        assert total == seen[b] + len(exp["action"]), (b, total, seen[b], len(exp["action"]))
        cap = r["state"].shape[0]
        idx = (np.arange(seen[b], total) % cap)
        for key in ("action", "reward", "done", "age"):
            assert np.array_equal(r[key].cpu().numpy()[idx], exp[key]), (b, key)
        assert np.array_equal(r["state"].cpu().numpy()[idx], exp["state"].astype(np.float32)), (b, "state")
        assert np.array_equal(r["state_prime"].cpu().numpy()[idx], exp["state_prime"].astype(np.float32)), (b, "state_prime")
        seen[b] = total


def test_transition_capture_matches_the_reference_learn_calls():
    """rl_capture_transitions == what trainer.py:95-96 / entities.py:194-208 hand to brain.learn, per brain, in call order
    (golden trace: the recorded learn() call list; ring capacity small enough to wrap)."""
    import golden_io
    from hip_backend import HipBackend
    from oracle import capture_np
    path = [p for p in golden_io.trace_files("trace_") if "dense100.npz" in p][0]
    tr = np.load(path)
    cfg, ticks = golden_io.trace_cfg(tr)
    hb = HipBackend(1, **cfg)
    hb.load_world(0, golden_io.initial_snapshot(tr))
    hb.observe()
    hb.dw.enable_capture(capacity=700)
    seen = [0] * cfg["n_brains"]
    prev = tr["init_obs"]
    for t in range(ticks):
        n0, n1 = int(tr["n0"][t]), int(tr["step_n"][t])
        acts = np.zeros((1, hb.cap), np.int8)
        acts[0, :n0] = tr["actions"][t][:n0]
        tape = hb.make_tape([golden_io.tick_tape(tr, t)])
        hb.step(acts, tape)
        hb.dw.capture_transitions()
        tx = capture_np.transitions(prev, tr["actions"][t], tr["step_src"][t][:n1], tr["step_age"][t][:n1], tr["step_flags"][t][:n1],
                                    tr["step_reward"][t][:n1], tr["step_done"][t][:n1], tr["step_obs"][t][:n1], tr["step_brain"][t][:n1])
        assert np.array_equal(tx["k"], tr["step_learn_k"][t][: int(tr["step_learn_n"][t])])  # the reference's own call list
        exp = [{k: v[tx["brain"] == b] for k, v in tx.items()} for b in range(cfg["n_brains"])]
        _check_rings(hb.dw, exp, seen)
        hb.update(tape)
        prev = tr["upd_obs"][t][: int(tr["upd_n"][t])]
    assert sum(seen) > 1500


def test_transition_capture_after_fused_ticks_with_policy_outputs():
    """Fused tick + batched policy: the ring rows are the policy-read observations / outputs of THAT tick (ping-pong state)."""
    import torch
    from hip_backend import HipBackend
    from oracle import capture_np, oracle as orc
    from reinlife_amd import _lib
    from reinlife_amd.worlds import pack_brain_weights
    import golden_io
    m = np.load(golden_io.GOLDEN_DIR + "/models.npz")
    R = 6
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=True, limit_reproduction=False, incentivize_killing=True)
    hb = HipBackend(R, seed=31, **cfg)
    ow = orc.OracleWorlds(n_worlds=R, seed=31, **cfg)
    hb.dw.set_brains([(_lib.PPO, 0.0, pack_brain_weights(_lib.PPO, m["PPO_weights"])),
                      (_lib.PERD3QN, 0.1, pack_brain_weights(_lib.PERD3QN, m["PERD3QN_weights"]))])
    hb.dw.reset_synthetic(100)
    ow.reset_synthetic(100)
    hb.dw.enable_capture(capacity=5000, with_prob=True)
    seen = [0, 0]
    for t in range(12):
        hb.dw.act(want_q=True)
        torch.cuda.synchronize()
        acts = hb.dw.actions.cpu().numpy().copy()
        q = hb.dw.out_q.cpu().numpy().copy()
        prev_obs = ow.obs2.copy()
        n0 = ow.s["n_agents"].copy()
        full = np.zeros((R, hb.cap), np.int8)
        for w in range(R):
            full[w, : n0[w]] = acts[w, : n0[w]]
        ow.step(full)
        n1 = ow.s["n_agents"].copy()
        l1 = {k: ow.s[k].copy() for k in ("a_age", "a_flags", "a_brain")}
        rew, done, src, obs1 = ow.reward.copy(), ow.done.copy(), ow.src1.copy(), ow.obs1.copy()
        ow.update()
        hb.dw.tick()
        hb.dw.capture_transitions(with_policy_out=True)
        exp = [dict(action=[], reward=[], done=[], age=[], state=[], state_prime=[], prob=[]) for _ in range(2)]
        for w in range(R):  # worlds append in any order; compare as multisets per brain below
            tx = capture_np.transitions(prev_obs[w], full[w], src[w, : n1[w]], l1["a_age"][w, : n1[w]], l1["a_flags"][w, : n1[w]],
                                        rew[w, : n1[w]], done[w, : n1[w]], obs1[w, : n1[w]], l1["a_brain"][w, : n1[w]])
            s = src[w, : n1[w]][tx["k"]]
            for b in range(2):
                sel = tx["brain"] == b
                for key in ("action", "reward", "done", "age", "state", "state_prime"):
                    exp[b][key].append(tx[key][sel])
                exp[b]["prob"].append(q[w, s[sel], tx["action"][sel]])
        for b in range(2):
            r = hb.dw.replays[b]
            total = int(r["count"].item())
            want = {k: np.concatenate(v) for k, v in exp[b].items()}
            assert total == seen[b] + len(want["action"])
            idx = np.arange(seen[b], total)
            got = {k: r[k].cpu().numpy()[idx] for k in ("action", "reward", "done", "age", "prob")}
            got["state"] = r["state"].cpu().numpy()[idx]; got["state_prime"] = r["state_prime"].cpu().numpy()[idx]
            def key_rows(d):
                return sorted(map(tuple, np.concatenate([d["state_prime"].astype(np.float32), d["state"].astype(np.float32),
                                                          np.stack([d["action"], d["reward"], d["done"], d["age"], d["prob"]], 1).astype(np.float32)], 1).tolist()))
            assert key_rows(got) == key_rows(want), (t, b)
            seen[b] = total
    assert min(seen) > 1000


def test_replicas_reset_like_the_reference_and_agents_of_any_world():
    """reset(): EVERY replica starts with one agent per brain, gene = brain index (environment.py:147-149); env.agents_of(w)
    exposes replica w through the same Agent protocol as env.agents, after step() and after update_env()."""
    import warnings
    from oracle import oracle as orc
    from reinlife_amd import Environment, Models
    brains = [Models.PERD3QN(training=False), Models.DQN(training=False), Models.D3QN(training=False)]
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        env = Environment(width=30, height=30, brains=brains, max_agents=100, n_worlds=5, seed=9, training=True, print_results=False)
    assert any("inference" in str(w.message) for w in rec)   # training=True is answered with a warning, not silence
    np.random.seed(1)
    env.reset()
    for w in range(5):
        ags = env.agents_of(w)
        assert sorted(a.gene for a in ags) == [0, 1, 2] and all(a.age == 0 and a.health == 200 for a in ags)
        grid = env.worlds.s["cell_type"][w].cpu().numpy()
        assert (grid == 3).sum() == 3 and (grid == 5).sum() == 1 and 40 < (grid == 1).sum() < 150
    assert not np.array_equal(env.worlds.s["cell_type"][1].cpu().numpy(), env.worlds.s["cell_type"][2].cpu().numpy())
    # drive replica 3 through per-agent actions and follow it with the oracle
    ow = orc.OracleWorlds(5, 30, 30, 100, 3, True, False, True, seed=9)
    for w in range(5):
        ow.load_world(w, env.worlds.world(w))
    ow.observe()
    rng = np.random.RandomState(4)
    for t in range(25):
        env.act(t)
        for a in env.agents_of(3):
            a.action = int(rng.randint(0, 8))
        acts = np.zeros((5, ow.cap), np.int8)
        n = ow.s["n_agents"]
        dev = env.worlds.actions.cpu().numpy()
        for w in range(5):
            acts[w, : n[w]] = dev[w, : n[w]]
        acts[3, : n[3]] = [a.action for a in env.agents_of(3)]
        env.step()
        ow.step(acts)
        ags = env.agents_of(3)
        n1 = int(ow.s["n_agents"][3])
        assert len(ags) == n1
        assert [a.health for a in ags] == list(ow.s["a_health"][3, :n1])
        assert np.allclose([a.reward for a in ags], ow.reward[3, :n1], atol=0)
        if n1:
            assert np.array_equal(np.stack([a.state_prime for a in ags]).astype(np.float32), ow.obs1[3, :n1])
            prev = ow.obs2[3][ow.src1[3, :n1]]                       # Agent.state is still what the policy read
            assert np.array_equal(np.stack([a.state for a in ags]).astype(np.float32), prev)
        env.update_env(t)
        ow.update()
        ags = env.agents_of(3)
        n2 = int(ow.s["n_agents"][3])
        assert len(ags) == n2
        if n2:
            assert np.array_equal(np.stack([a.state for a in ags]).astype(np.float32), ow.obs2[3, :n2])


def test_agent_learn_hands_the_reference_keyword_set_to_the_brain():
    """entities.py:194-208: PPO gets prob= and no caller kwargs, DQN no caller kwargs, D3QN / PERD3QN the caller's n_epi."""
    from reinlife_amd import Environment, Models
    calls = []

    class Spy:
        def __init__(self, inner):
            self.__dict__["inner"] = inner
        def __getattr__(self, k):
            return getattr(self.inner, k)
        def learn(self, **kw):
            calls.append((self.inner.method, sorted(kw)))

    brains = [Spy(Models.PPO()), Spy(Models.DQN(training=False)), Spy(Models.PERD3QN(training=False))]
    env = Environment(width=30, height=30, brains=brains, max_agents=100, training=False, seed=2)
    np.random.seed(5)
    env.reset()
    for t in range(4):
        for a in env.agents:
            a.get_action(t)
        env.step()
        for a in env.agents:
            a.learn(n_epi=t)
        env.update_env(t)
    base = ["action", "age", "dead", "done", "reward", "state", "state_prime"]
    got = {m: k for m, k in calls}
    assert got["PPO"] == sorted(base + ["prob"]) and got["DQN"] == base and got["PERD3QN"] == sorted(base + ["n_epi"])
