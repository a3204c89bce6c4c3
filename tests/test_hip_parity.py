"""GPU: the HIP path (through the C ABI) against the reference's golden vectors and against the CPU oracle."""
import numpy as np
import pytest

import golden_io

pytestmark = pytest.mark.gpu


def _hip(**kw):
    from hip_backend import HipBackend
    return lambda n_worlds, **cfg: HipBackend(n_worlds, **kw, **cfg)


@pytest.mark.parametrize("path", golden_io.trace_files("micro_"), ids=lambda p: p.split("/")[-1])
def test_hip_replays_quirk_microworlds(path):
    golden_io.replay_trace(_hip(), path, n_worlds=2)


@pytest.mark.parametrize("path", golden_io.trace_files("trace_"), ids=lambda p: p.split("/")[-1])
def test_hip_replays_reference_traces(path):
    golden_io.replay_trace(_hip(), path, n_worlds=3)


def _compare_states(hb, ow, tag):
    for name in ("cell_type", "n_agents", "max_gene", "next_uid", "tick", "epoch"):
        got = hb.dw.s[name].cpu().numpy()
        assert np.array_equal(got, ow.s[name]), "%s: %s differs" % (tag, name)
    n = ow.s["n_agents"]
    for name in ("a_i", "a_j", "a_health", "a_age", "a_max_age", "a_gene", "a_brain", "a_uid", "a_flags", "a_action",
                 "a_fitness"):
        got = hb.dw.s[name].cpu().numpy()
        for w in range(ow.R):
            assert np.array_equal(got[w, : n[w]], ow.s[name][w, : n[w]]), "%s: %s differs in world %d" % (tag, name, w)
    for name in ("best_uid", "best_fit", "best_brain"):
        assert np.array_equal(hb.dw.s[name].cpu().numpy(), ow.s[name]), "%s: %s differs" % (tag, name)


def _compare_rows(got, want, n, tag):
    got = np.asarray(got)
    for w in range(len(n)):
        assert np.array_equal(got[w, : n[w]], want[w, : n[w]]), "%s differs in world %d" % (tag, w)


@pytest.mark.parametrize("static,limit,fused", [(True, False, False), (False, False, False), (True, True, True),
                                                (False, False, True)])
def test_hip_matches_oracle_free_running_philox(static, limit, fused):
    """Synthetic worlds, in-kernel Philox draws, random actions: GPU and oracle must stay bit-identical for many ticks
    (integer state exact; rewards / observations float32-identical), split and fused launches alike."""
    from hip_backend import HipBackend
    from oracle import oracle as orc
    R, ticks = 48, 60
    cfg = dict(width=30, height=30, max_agents=100, n_brains=3, static_families=static, limit_reproduction=limit,
               incentivize_killing=True)
    hb = HipBackend(R, seed=1234, **cfg)
    ow = orc.OracleWorlds(n_worlds=R, seed=1234, **cfg)
    hb.dw.reset_synthetic(100)
    ow.reset_synthetic(100)
    _compare_states(hb, ow, "reset")
    _compare_rows(hb.obs2, ow.obs2, ow.s["n_agents"], "reset obs")
    rng = np.random.RandomState(5)
    for t in range(ticks):
        acts = rng.randint(0, 8, size=(R, hb.cap)).astype(np.int8)
        n0 = ow.s["n_agents"].copy()
        ow.step(acts)
        if fused:
            ow_n1 = ow.s["n_agents"].copy()
            ow.update()
            hb.tick(acts)
        else:
            hb.step(acts)
            _compare_states(hb, ow, "tick %d step" % t)
            ow_n1 = ow.s["n_agents"].copy()
        assert np.array_equal(hb.n_acted, n0)
        _compare_rows(hb.reward, ow.reward, ow_n1, "tick %d reward" % t)
        _compare_rows(hb.done, ow.done, ow_n1, "tick %d done" % t)
        _compare_rows(hb.src1, ow.src1, ow_n1, "tick %d src1" % t)
        _compare_rows(hb.obs1, ow.obs1, ow_n1, "tick %d obs1" % t)
        assert np.array_equal(hb.trk_tick, ow.trk_tick) and np.array_equal(hb.trk_pop, ow.trk_pop), "tick %d tracker" % t
        assert np.array_equal(hb.trk_sum, ow.trk_sum) and np.array_equal(hb.trk_cnt, ow.trk_cnt), "tick %d tracker sums" % t
        if not fused:
            ow.update()
            hb.update()
        _compare_states(hb, ow, "tick %d update" % t)
        _compare_rows(hb.src2, ow.src2, ow.s["n_agents"], "tick %d src2" % t)
        _compare_rows(hb.obs2, ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)
        if t % 20 == 19:
            ow.refill(70, 100)
            hb.dw.refill(70, 100)
            _compare_states(hb, ow, "tick %d refill" % t)


def test_hip_fused_tick_refill_matches_oracle():
    """rl_tick_refill == step + update_env + refill(70, 100), one launch; runs long enough for many refills."""
    from hip_backend import HipBackend
    from oracle import oracle as orc
    R = 40
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=True, limit_reproduction=False,
               incentivize_killing=True)
    hb = HipBackend(R, seed=4321, world_base=1000, **cfg)
    ow = orc.OracleWorlds(n_worlds=R, seed=4321, world_base=1000, **cfg)
    hb.dw.reset_synthetic(100)
    ow.reset_synthetic(100)
    rng = np.random.RandomState(11)
    refills = 0
    for t in range(120):
        acts = rng.randint(0, 8, size=(R, hb.cap)).astype(np.int8)
        ow.step(acts); ow.update()
        refills += ow.refill(70, 100)
        hb.dw.set_actions(acts)
        hb.dw.tick_refill(70, 100)
        hb.dw.check_error_flag()
        _compare_states(hb, ow, "tick %d" % t)
        _compare_rows(hb.obs2, ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)
    assert refills > 20 and int(hb.dw.refill_count.item()) == refills


@pytest.mark.parametrize("static,limit,incentive", [(True, False, True), (False, False, True), (True, True, True), (False, True, False)],
                         ids=["static", "nonstatic", "static-limit", "nonstatic-limit-noincentive"])
@pytest.mark.parametrize("generic", [False, True], ids=["specialised", "generic"])
def test_hip_lean_tick_matches_oracle(static, limit, incentive, generic, hip_option):
    """The launch bench.py times: the lean fused tick (no tape, no tracker, no capture outputs) on the reference's default
    30x30 / 100-agent shape -- the shape-specialised kernel and, with RL_WORLD_GENERIC set, the generic code for the same
    worlds -- against the oracle: state, rewards, done flags, both permutations and both observation passes, every tick."""
    from oracle import oracle as orc
    from reinlife_amd.worlds import DeviceWorlds
    if generic:
        hip_option("world_generic", 1)   # read by the library at every launch
    else:
        hip_option("world_generic", 0)
    R = 32
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=static, limit_reproduction=limit,
               incentivize_killing=incentive)
    dw = DeviceWorlds(n_worlds=R, seed=777, world_base=64, **cfg)
    assert dw.cap == 256

    class _HB:  # the attribute _compare_states reads
        pass
    hb = _HB(); hb.dw = dw
    ow = orc.OracleWorlds(n_worlds=R, seed=777, world_base=64, **cfg)
    dw.reset_synthetic(100); ow.reset_synthetic(100)
    rng = np.random.RandomState(17)
    refills = 0
    for t in range(70):
        acts = rng.randint(0, 8, size=(R, dw.cap)).astype(np.int8)
        n0 = ow.s["n_agents"].copy()
        ow.step(acts)
        n1 = ow.s["n_agents"].copy()
        want = {k: getattr(ow, k).copy() for k in ("reward", "done", "src1", "obs1")}
        ow.update()
        want_src2 = ow.src2.copy(); n2 = ow.s["n_agents"].copy(); epoch = ow.s["epoch"].copy()
        refills += ow.refill(70, 100)
        dw.set_actions(acts)
        dw.tick_refill(70, 100)
        dw.check_error_flag()
        assert np.array_equal(dw.n_acted.cpu().numpy(), n0)
        _compare_rows(dw.reward.cpu().numpy(), want["reward"], n1, "tick %d reward" % t)
        _compare_rows(dw.done.cpu().numpy(), want["done"], n1, "tick %d done" % t)
        _compare_rows(dw.src1.cpu().numpy(), want["src1"], n1, "tick %d src1" % t)
        _compare_rows(dw.obs_state_prime().cpu().numpy(), want["obs1"], n1, "tick %d obs1" % t)
        _compare_states(hb, ow, "tick %d" % t)
        _compare_rows(dw.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)
        keep = ow.s["epoch"] == epoch   # (a refilled world has a fresh list: its update permutation is not defined)
        got_src2 = dw.src2.cpu().numpy()
        for w in np.nonzero(keep)[0]:
            assert np.array_equal(got_src2[w, : n2[w]], want_src2[w, : n2[w]]), "tick %d src2 world %d" % (t, w)
    assert refills > 5


@pytest.mark.parametrize("width,height,max_agents,thr,n_new", [(20, 15, 60, 45, 50), (7, 5, 12, 9, 10), (64, 64, 100, 90, 100),
                                                               (30, 30, 200, 150, 180)])
def test_hip_lean_tick_refills_on_other_shapes(width, height, max_agents, thr, n_new):
    """Lean fused tick + refill on shapes the specialised kernel does not cover (generic code, prepared and in-line refills):
    thresholds close to the population so that worlds refill every few ticks."""
    from oracle import oracle as orc
    from reinlife_amd.worlds import DeviceWorlds
    R = 12
    cfg = dict(width=width, height=height, max_agents=max_agents, n_brains=3, static_families=True, limit_reproduction=False,
               incentivize_killing=True)
    dw = DeviceWorlds(n_worlds=R, seed=31, **cfg)

    class _HB:
        pass
    hb = _HB(); hb.dw = dw
    ow = orc.OracleWorlds(n_worlds=R, seed=31, **cfg)
    dw.reset_synthetic(n_new); ow.reset_synthetic(n_new)
    rng = np.random.RandomState(3)
    refills = 0
    for t in range(40):
        acts = rng.randint(0, 8, size=(R, dw.cap)).astype(np.int8)
        ow.step(acts); ow.update()
        refills += ow.refill(thr, n_new)
        dw.set_actions(acts)
        dw.tick_refill(thr, n_new)
        dw.check_error_flag()
        _compare_states(hb, ow, "tick %d" % t)
        _compare_rows(dw.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)
    assert refills >= 3, refills
    assert int(dw.refill_count.item()) == refills


def test_hip_small_and_rect_grids_match_oracle():
    from hip_backend import HipBackend
    from oracle import oracle as orc
    for (w, h, ma, n) in ((7, 5, 12, 10), (30, 20, 80, 60), (64, 64, 100, 100), (3, 3, 4, 3)):
        cfg = dict(width=w, height=h, max_agents=ma, n_brains=2, static_families=True, limit_reproduction=False,
                   incentivize_killing=False)
        hb = HipBackend(5, seed=77, **cfg)
        ow = orc.OracleWorlds(n_worlds=5, seed=77, **cfg)
        hb.dw.reset_synthetic(n)
        ow.reset_synthetic(n)
        rng = np.random.RandomState(w * 100 + h)
        for t in range(40):
            acts = rng.randint(0, 8, size=(5, hb.cap)).astype(np.int8)
            ow.step(acts); ow.update()
            hb.step(acts); hb.update()
            _compare_states(hb, ow, "%dx%d tick %d" % (w, h, t))
            _compare_rows(hb.obs2, ow.obs2, ow.s["n_agents"], "%dx%d tick %d obs2" % (w, h, t))


def test_hip_policy_forward_matches_reference_and_oracle():
    """Block-scaled 2 x f16 split-precision MFMA MLPs (f32-grade) vs the reference's torch networks (golden) and the C
    oracle: 1e-5 (north_star tolerance; the measured difference is ~2e-7)."""
    import torch
    from oracle import oracle as orc
    from reinlife_amd import _lib
    from reinlife_amd.worlds import pack_brain_weights, policy_forward
    m = np.load(golden_io.GOLDEN_DIR + "/models.npz")
    obs = torch.as_tensor(m["obs"], device="cuda:0")
    for name in ("DQN", "D3QN", "PERD3QN", "PPO"):
        kind = _lib.KIND_BY_METHOD[name]
        packed = pack_brain_weights(kind, m[name + "_weights"])
        for n in (1, 31, 32, 33, obs.shape[0]):
            out = policy_forward(kind, packed, obs[:n].contiguous()).cpu().numpy()
            np.testing.assert_allclose(out, m[name + "_out"][:n], rtol=0, atol=1e-5, err_msg="%s n=%d vs reference" % (name, n))
            ora = orc.policy_forward(orc.KIND_BY_NAME[name], m[name + "_weights"], m["obs"][:n])
            np.testing.assert_allclose(out, ora, rtol=0, atol=1e-5, err_msg="%s n=%d vs oracle" % (name, n))


def test_hip_policy_split_precision_holds_over_the_f32_range():
    """The power-of-two row scaling makes the f16 split independent of magnitudes: weights / observations scaled up or
    down by large factors (activations up to ~1e6, down to ~1e-12) still match the f32 oracle to 1e-5 RELATIVE to the
    output scale, for the Q-value brains (PPO's softmax output is scale-free; it is checked unscaled above)."""
    import torch
    from oracle import oracle as orc
    from reinlife_amd import _lib
    from reinlife_amd.worlds import pack_brain_weights, policy_forward
    m = np.load(golden_io.GOLDEN_DIR + "/models.npz")
    rng = np.random.RandomState(5)
    obs = m["obs"][:200].copy()
    for name in ("DQN", "D3QN", "PERD3QN"):
        kind = _lib.KIND_BY_METHOD[name]
        for wscale, xscale in ((40.0, 25.0), (1e-3, 1e-4), (7.0, 1.0)):
            w = (m[name + "_weights"] * wscale * rng.uniform(0.5, 1.5, size=m[name + "_weights"].shape)).astype(np.float32)
            x = (obs * xscale).astype(np.float32)
            out = policy_forward(kind, pack_brain_weights(kind, w), torch.as_tensor(x, device="cuda:0")).cpu().numpy()
            ora = orc.policy_forward(orc.KIND_BY_NAME[name], w, x)
            scale = float(np.abs(ora).max())
            assert np.isfinite(out).all() and scale > 0
            np.testing.assert_allclose(out / scale, ora / scale, rtol=0, atol=1e-5, err_msg="%s w*%g x*%g" % (name, wscale, xscale))


def test_hip_policy_act_over_worlds_matches_oracle():
    """rl_policy_act: per-agent brain dispatch (bucketed by brain), greedy / eps-greedy / categorical Philox draws."""
    import torch
    from hip_backend import HipBackend
    from oracle import oracle as orc
    from reinlife_amd import _lib
    from reinlife_amd.worlds import pack_brain_weights
    m = np.load(golden_io.GOLDEN_DIR + "/models.npz")
    names = ["PERD3QN", "PPO", "DQN", "D3QN"]
    R = 24
    cfg = dict(width=30, height=30, max_agents=100, n_brains=4, static_families=True, limit_reproduction=False,
               incentivize_killing=True)
    hb = HipBackend(R, seed=99, **cfg)
    ow = orc.OracleWorlds(n_worlds=R, seed=99, **cfg)
    hb.dw.reset_synthetic(100)
    ow.reset_synthetic(100)
    eps = [0.0, 0.0, 0.3, 0.0]
    hb.dw.set_brains([(_lib.KIND_BY_METHOD[n], e, pack_brain_weights(_lib.KIND_BY_METHOD[n], m[n + "_weights"]))
                      for n, e in zip(names, eps)])
    for t in range(3):
        hb.dw.act(want_q=True)
        torch.cuda.synchronize()
        acts = hb.dw.actions.cpu().numpy()
        q = hb.dw.out_q.cpu().numpy()
        n = ow.s["n_agents"]
        mismatches = 0
        total = 0
        for b, name in enumerate(names):
            rows = [(w, k) for w in range(R) for k in range(n[w]) if ow.s["a_brain"][w, k] == b]
            assert rows
            ws = np.array([r[0] for r in rows]); ks = np.array([r[1] for r in rows])
            want_q = orc.policy_forward(orc.KIND_BY_NAME[name], m[name + "_weights"], ow.obs2[ws, ks])
            np.testing.assert_allclose(q[ws, ks], want_q, rtol=0, atol=1e-5, err_msg=name)
            # action selection on the GPU's own outputs must equal the oracle's rule bit for bit
            want_a = orc.select_actions(ow.cfg, orc.KIND_BY_NAME[name], q[ws, ks], ws, ks, ow.s["tick"], ow.s["epoch"], eps[b])
            mismatches += int((acts[ws, ks] != want_a).sum()); total += len(rows)
        assert mismatches == 0, "%d / %d actions differ" % (mismatches, total)
        full = np.zeros((R, hb.cap), np.int8)
        for w in range(R):
            full[w, : n[w]] = acts[w, : n[w]]
        ow.step(full); ow.update()
        hb.tick(full)
        _compare_rows(hb.obs2, ow.obs2, ow.s["n_agents"], "obs after act tick %d" % t)


def _mk(R, seed=3, **over):
    from hip_backend import HipBackend
    from oracle import oracle as orc
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=True, limit_reproduction=False,
               incentivize_killing=True)
    cfg.update(over)
    return HipBackend(R, seed=seed, **cfg), orc.OracleWorlds(n_worlds=R, seed=seed, **cfg), cfg


def _run_both(hb, ow, ticks, rng, fused=False, action_fn=None):
    for t in range(ticks):
        acts = (action_fn(rng) if action_fn else rng.randint(0, 8, size=(ow.R, hb.cap))).astype(np.int8)
        ow.step(acts); ow.update()
        if fused:
            hb.tick(acts)
        else:
            hb.step(acts); hb.update()
        _compare_states(hb, ow, "tick %d" % t)
        _compare_rows(hb.obs2, ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)


def test_edge_empty_worlds_repopulate_through_produce():
    """No agents at all: step/update are no-ops except _add_food and _produce (environment.py:528), which re-seeds."""
    hb, ow, _ = _mk(32, n_brains=3)
    hb.dw.reset_synthetic(0); ow.reset_synthetic(0)
    _compare_states(hb, ow, "empty reset")
    _run_both(hb, ow, 120, np.random.RandomState(1))
    assert ow.s["n_agents"].sum() > 0  # produce fired somewhere


def test_edge_full_grid_set_random_finds_no_cell():
    """A grid without empty cells: Grid.set_random returns None without drawing (grid.py:82-83); movers all conflict."""
    from oracle import oracle as orc
    hb, ow, cfg = _mk(4, width=8, height=8, max_agents=40)
    hb.dw.reset_synthetic(30); ow.reset_synthetic(30)
    for w in range(4):
        snap = ow.world(w)
        ct = snap["cell_type"].copy()
        ct[ct == orc.EMPTY] = orc.POISON if w % 2 else orc.FOOD
        snap["cell_type"] = ct
        snap["age"] = np.full_like(snap["age"], 10)  # old enough to reproduce: births must find no cell
        ow.load_world(w, snap); hb.load_world(w, snap)
    _run_both(hb, ow, 25, np.random.RandomState(2))


def test_edge_invalid_and_unset_actions_are_noops():
    """Newborns carry action -1 (entities.py:153); out-of-range actions neither move nor attack here."""
    hb, ow, _ = _mk(8)
    hb.dw.reset_synthetic(100); ow.reset_synthetic(100)
    _run_both(hb, ow, 20, np.random.RandomState(3), action_fn=lambda r: r.randint(-1, 10, size=(8, 256)))


def test_edge_minimum_grid_and_many_brains():
    hb, ow, _ = _mk(6, width=3, height=3, max_agents=4, n_brains=64)
    hb.dw.reset_synthetic(4); ow.reset_synthetic(4)
    _run_both(hb, ow, 60, np.random.RandomState(4), fused=True)
    hb, ow, _ = _mk(3, width=64, height=64, max_agents=300, n_brains=5, static_families=False)  # largest world: 120 KB of LDS
    hb.dw.reset_synthetic(500); ow.reset_synthetic(500)
    _run_both(hb, ow, 8, np.random.RandomState(5), fused=True, action_fn=lambda r: r.randint(0, 8, size=(3, hb.cap)))


def test_edge_inconsistent_tape_sets_the_error_flag():
    """A recorded draw that cannot belong to this world (index beyond the number of empty cells) must be reported through
    the device error flag, not silently wrapped (the oracle returns an error for the same tape)."""
    from oracle import oracle as orc
    from reinlife_amd import _lib
    hb, ow, _ = _mk(2)
    hb.dw.reset_synthetic(50); ow.reset_synthetic(50)
    bad = {"food_k": np.full(7, 5000, np.int32), "food_u": np.zeros(7), "repro_u": np.zeros(hb.cap), "birth_k": np.zeros(hb.cap + 1, np.int32),
           "produce_u": 0.0, "produce_choice": 0}
    acts = np.zeros((2, hb.cap), np.int8)
    with pytest.raises(RuntimeError):
        ow.step(acts, ow.make_tape([bad, bad]))
    with pytest.raises(_lib.ReinLifeHipError, match="code 1"):
        hb.step(acts, hb.make_tape([bad, bad]))
