"""Randomised configurations through the multi-tick launch (confidence check; GPU + oracle).   python tools/fuzz_parity.py [cases] [seed]
Per case a random world shape, capacity, brain list (any kinds, 1..6 brains), family mode, reproduction limit, refill threshold and
launch mode (plain / Tracker + epsilon schedule / + transition capture) is drawn; then
  * B runs tick by tick (rl_run(1)) and the oracle is fed B's actions: state every 10 ticks, Agent.state rows, Tracker sums -- bit-identical;
  * A runs the same ticks in launches of random lengths: at the end of every launch A's state, rows and last actions equal B's.
Worlds the launch does not support (capacity > workgroup, LDS) take the two-launch loop inside DeviceWorlds.run: same checks."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from reinlife_amd import _lib  # noqa: E402
from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
KINDS = ["DQN", "D3QN", "PERD3QN", "PPO"]


def same_state(a, b, n, tag):
    for key in a:
        x, y = a[key], b[key]
        if key.startswith("a_"):
            for w in range(len(n)):
                assert np.array_equal(x[w, :n[w]], y[w, :n[w]]), (tag, key, w)
        else:
            assert np.array_equal(x.reshape(y.shape), y), (tag, key)


def ring_rows(dw, b, lo, hi):
    r = dw.replays[b]
    idx = np.arange(lo, hi) % r["state"].shape[0]
    cols = [r["state"][idx].cpu().numpy(), r["state_prime"][idx].cpu().numpy(), r["reward"][idx].cpu().numpy()[:, None]]
    if r["prob"] is not None:
        cols.append(r["prob"][idx].cpu().numpy()[:, None])
    cols += [r[k][idx].cpu().numpy()[:, None].astype(np.float32) for k in ("action", "done", "age")]
    rows = np.ascontiguousarray(np.concatenate(cols, axis=1))
    return rows[np.lexsort(rows.T[::-1])]


def host_state(dw):
    return {k: v.cpu().numpy() for k, v in dw.s.items()}


total_steps = fused_cases = 0
for case in range(cases):
    big = rng.rand() < 0.25
    W, H = (int(rng.randint(5, 65)), int(rng.randint(5, 65))) if big else (int(rng.randint(5, 41)), int(rng.randint(5, 41)))
    max_agents = int(rng.randint(6, min(400 if big else 150, W * H // 3) + 1))
    nb = int(rng.randint(1, 7))
    kinds = [KINDS[int(rng.randint(4))] for _ in range(nb)] if rng.rand() < 0.6 else [KINDS[int(rng.randint(4))]] * nb
    static = bool(rng.rand() < 0.5)
    cfg = dict(width=W, height=H, max_agents=max_agents, n_brains=nb, static_families=static, limit_reproduction=bool(rng.rand() < 0.3),
               incentivize_killing=bool(rng.rand() < 0.7))
    R = int(rng.choice([1, 3, 8, 17]))
    seed = int(rng.randint(1 << 30))
    n0 = int(rng.randint(max(2, max_agents // 3), max_agents + 1))
    thr = int(rng.randint(1, n0 + 1)) if rng.rand() < 0.7 else -1
    mode = ["plain", "train", "capture"][int(rng.randint(3))]
    train, capture = mode != "plain", mode == "capture"
    ticks = int(rng.randint(60, 140))
    tag = "case %d: %dx%d max_agents %d brains %s %s R=%d n0=%d thr=%d %s ticks=%d" % (
        case, W, H, max_agents, "+".join(kinds), "static" if static else "non-static", R, n0, thr, mode, ticks)
    brains = [(_lib.KIND_BY_METHOD[n], float(rng.choice([0.0, 0.05, 0.3])), pack_brain_weights(_lib.KIND_BY_METHOD[n], bench.brain_weights(n, 100 + k)))
              for k, n in enumerate(kinds)]
    try:
        A, B = DeviceWorlds(n_worlds=R, seed=seed, **cfg), DeviceWorlds(n_worlds=R, seed=seed, **cfg)
    except _lib.ReinLifeHipError as e:
        print("skip " + tag + ": " + str(e)[:80], flush=True)
        continue
    ow = orc.OracleWorlds(n_worlds=R, seed=seed, **cfg)
    for dw in (A, B):
        dw.set_brains(brains)
        if train:
            dw.enable_tracking(True)
        if capture:
            dw.enable_capture(capacity=R * max_agents * ticks + 1024, with_prob=True)
        dw.reset_synthetic(n0)
    ow.reset_synthetic(n0)
    fused_cases += bool(A.run_supported())
    sched = rng.uniform(0, 0.4, size=(ticks, nb)).astype(np.float32) if train else None
    t, next_cut, seen = 0, 0, [0] * nb
    while t < ticks:
        if t == next_cut:   # A: one launch up to the next cut
            k = int(min(ticks - t, rng.choice([1, 2, 3, 7, 20, 45])))
            A.run(k, thr, n0, eps_schedule=sched[t:t + k] if train else None)
            next_cut = t + k
        B.run(1, thr, n0, eps_schedule=sched[t:t + 1] if train else None)
        acts = B.actions.cpu().numpy().copy()
        total_steps += int(ow.s["n_agents"].sum())
        ow.step(acts)
        ow.update()
        if thr >= 0:
            ow.refill(thr, n0)
        t += 1
        if t % 10 == 0 or t == ticks:
            torch.cuda.synchronize(); B.check_error_flag()
            n = ow.s["n_agents"]
            same_state(host_state(B), ow.s, n, (tag, t, "B vs oracle"))
            o = B.obs_state().cpu().numpy()
            for w in range(R):
                assert np.array_equal(o[w, :n[w]], ow.obs2[w, :n[w]]), (tag, t, "rows vs oracle", w)
            if train:
                assert np.array_equal(B.trk_sum.cpu().numpy(), ow.trk_sum) and np.array_equal(B.trk_cnt.cpu().numpy(), ow.trk_cnt), (tag, t, "tracker vs oracle")
        if t == next_cut:
            torch.cuda.synchronize(); A.check_error_flag()
            n = B.s["n_agents"].cpu().numpy()
            same_state(host_state(A), host_state(B), n, (tag, t, "A vs B"))
            oa, ob = A.obs_state().cpu().numpy(), B.obs_state().cpu().numpy()
            for w in range(R):
                assert np.array_equal(oa[w, :n[w]], ob[w, :n[w]]), (tag, t, "rows A vs B", w)
            if train:
                assert torch.equal(A.trk_sum, B.trk_sum) and torch.equal(A.trk_cnt, B.trk_cnt), (tag, t, "tracker A vs B")
            if capture:   # the launch's transitions of these ticks, as a set (worlds interleave by atomics within a tick)
                for b in range(nb):
                    ta, tb = int(A.replays[b]["count"].item()), int(B.replays[b]["count"].item())
                    assert ta == tb >= seen[b], (tag, t, "ring count", b, ta, tb)
                    assert np.array_equal(ring_rows(A, b, seen[b], ta), ring_rows(B, b, seen[b], tb)), (tag, t, "ring rows", b)
                    seen[b] = ta
    print("ok  " + tag + ("" if A.run_supported() else "  [two-launch loop]"), flush=True)
    del A, B
print("fuzz ok: %d cases (%d through the multi-tick kernel), %d agent-steps checked against the oracle" % (cases, fused_cases, total_steps))
