"""Replay of tests/golden/ fixtures (recorded from the real reference by oracle/gen_golden.py) through a backend.

A backend is anything with the OracleWorlds surface (oracle/oracle.py): load_world / world / make_tape / step /
update / observe and the output arrays n_acted, reward, done, src1, obs1, src2, obs2 (numpy, [R, cap, ...]).
The CPU tests pass the oracle; the GPU tests pass tests/hip_backend.py (the product's C-ABI path).
"""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
AGENT_KEYS = ("i", "j", "health", "age", "max_age", "gene", "brain", "uid", "flags", "action", "fitness")


def trace_files(prefix=""):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def trace_cfg(tr):
    w, h, max_agents, n_brains, static, limit, incentive, cap, ticks = [int(x) for x in tr["cfg"]]
    return dict(width=w, height=h, max_agents=max_agents, n_brains=n_brains, static_families=bool(static),
                limit_reproduction=bool(limit), incentivize_killing=bool(incentive), slot_cap=cap), ticks


def initial_snapshot(tr):
    n = int(tr["init_n"])
    snap = {k: tr["init_" + k][:n] for k in AGENT_KEYS}
    snap["cell_type"] = tr["init_cell_type"]
    snap["next_uid"] = int(tr["init_next_uid"])
    snap["max_gene"] = int(tr["init_max_gene"])
    for k in ("best_uid", "best_fit", "best_brain"):
        snap[k] = tr["init_" + k]
    return snap


def tick_tape(tr, t):
    return {k: tr["tape_" + k][t] for k in ("food_k", "food_u", "repro_u", "birth_k", "produce_u", "produce_choice")}


def _eq(tag, key, got, want):
    got, want = np.asarray(got), np.asarray(want)
    if got.shape != want.shape or not np.array_equal(got, want):
        bad = np.argwhere(got != want)[:4].tolist() if got.shape == want.shape else "shape %s vs %s" % (got.shape, want.shape)
        raise AssertionError("%s: %s differs at %s\n got  %s\n want %s" % (tag, key, bad, got, want))


def check_world(tag, got, tr, phase, t, static):
    n = int(tr[phase + "_n"][t])
    _eq(tag, "cell_type", got["cell_type"], tr[phase + "_cell_type"][t])
    for k in AGENT_KEYS:
        _eq(tag, k, got[k], tr[phase + "_" + k][t][:n])
    _eq(tag, "max_gene", int(got["max_gene"]), int(tr[phase + "_max_gene"][t]))
    if not static:
        for k in ("best_uid", "best_fit", "best_brain"):
            _eq(tag, k, got[k], tr[phase + "_" + k][t])


def replay_trace(make_backend, path, n_worlds=1, obs_exact=True):
    """Free-running replay: the backend is loaded once with the initial world and must then track the reference for
    every tick -- integer state bit-exact, rewards/observations float32-identical (or 1e-5 when obs_exact=False)."""
    tr = np.load(path)
    cfg, ticks = trace_cfg(tr)
    be = make_backend(n_worlds=n_worlds, **cfg)
    cap = cfg["slot_cap"]
    static = cfg["static_families"]
    snap = initial_snapshot(tr)
    for w in range(n_worlds):
        be.load_world(w, snap)
    name = os.path.basename(path)
    obs0 = np.asarray(be.observe())
    n_init = int(tr["init_n"])
    for w in range(n_worlds):
        _cmp_obs("%s initial obs w%d" % (name, w), obs0[w, :n_init], tr["init_obs"], obs_exact)
    for t in range(ticks):
        n0 = int(tr["n0"][t])
        acts = np.zeros((n_worlds, cap), np.int8)
        acts[:, :n0] = tr["actions"][t][:n0]
        tape = be.make_tape([tick_tape(tr, t)] * n_worlds)
        be.step(acts, tape)
        n1 = int(tr["step_n"][t])
        for w in range(n_worlds):
            tag = "%s tick %d step w%d" % (name, t, w)
            check_world(tag, be.world(w), tr, "step", t, True)
            _eq(tag, "n_acted", int(np.asarray(be.n_acted)[w]), n0)
            _eq(tag, "src", np.asarray(be.src1)[w, :n1], tr["step_src"][t][:n1])
            _eq(tag, "done", np.asarray(be.done)[w, :n1], tr["step_done"][t][:n1])
            _cmp_obs(tag + " reward", np.asarray(be.reward)[w, :n1], tr["step_reward"][t][:n1], obs_exact)
            _cmp_obs(tag + " obs", np.asarray(be.obs1)[w, :n1], tr["step_obs"][t][:n1], obs_exact)
            if "trk_tick" in tr.files and getattr(be, "trk_tick", None) is not None:  # Tracker per-tick values, bit-exact
                _eq(tag, "trk_tick", np.asarray(be.trk_tick)[w], tr["trk_tick"][t])
                _eq(tag, "trk_pop", float(np.asarray(be.trk_pop)[w, 0]), float(tr["trk_pop"][t]))
        be.update(tape)
        n2 = int(tr["upd_n"][t])
        for w in range(n_worlds):
            tag = "%s tick %d update w%d" % (name, t, w)
            check_world(tag, be.world(w), tr, "upd", t, static)
            _eq(tag, "src", np.asarray(be.src2)[w, :n2], tr["upd_src"][t][:n2])
            _cmp_obs(tag + " obs", np.asarray(be.obs2)[w, :n2], tr["upd_obs"][t][:n2], obs_exact)
    return tr


def _cmp_obs(tag, got, want, exact):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    if got.shape != want.shape:
        raise AssertionError("%s: shape %s vs %s" % (tag, got.shape, want.shape))
    if exact:
        if not np.array_equal(got, want):
            bad = np.argwhere(got != want)[:4]
            raise AssertionError("%s differs at %s: got %s want %s" % (tag, bad.tolist(), got[tuple(bad[0])], want[tuple(bad[0])]))
    else:
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-5, err_msg=tag)
