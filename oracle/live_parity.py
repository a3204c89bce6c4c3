"""TEST INFRASTRUCTURE ONLY -- pins the C oracle against the REAL reference, live, in the build container.

    python oracle/live_parity.py [--quick]

Runs the imported reference (oracle/ref_harness.py) tick by tick on randomised worlds -- natural populations, dense
100/200/300-agent worlds, small grids, static and non-static families -- feeding the recorded actions and RNG tape to
the oracle, and requires bit-identical integer state, float32-identical observations and rewards, after every step
and every update_env.  The oracle is NOT re-synchronised between ticks: it free-runs from the initial snapshot.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402

AGENT_KEYS = ("i", "j", "health", "age", "max_age", "gene", "brain", "uid", "flags", "action", "fitness")


def compare_world(tag, got, want, check_best):
    for key in ("cell_type",) + AGENT_KEYS:
        g, w = got[key], want[key]
        if g.shape != w.shape or not np.array_equal(g, w):
            raise AssertionError("%s: field %s differs\n got  %s\n want %s" % (tag, key, g, w))
    if int(got["max_gene"]) != int(want["max_gene"]):
        raise AssertionError("%s: max_gene %s != %s" % (tag, got["max_gene"], want["max_gene"]))
    if check_best:
        for key in ("best_uid", "best_fit", "best_brain"):
            if not np.array_equal(got[key], want[key]):
                raise AssertionError("%s: %s differs\n got  %s\n want %s" % (tag, key, got[key], want[key]))


def run_case(name, seed, ticks, n_brains=2, width=30, height=30, max_agents=100, static=True, limit=False,
             incentive=True, fill=0, p_attack=None, verbose=True):
    rh.seed_all(seed)
    env = rh.make_env(n_brains=n_brains, width=width, height=height, max_agents=max_agents, static_families=static,
                      limit_reproduction=limit, incentivize_killing=incentive)
    env.reset()
    rng = np.random.RandomState(seed + 12345)
    if fill:
        rh.fill_agents(env, fill, rng)
    cap = orc.slot_cap_for(max(max_agents, fill), width * height)
    ow = orc.OracleWorlds(1, width, height, max_agents, n_brains, static, limit, incentive, seed=seed, slot_cap=cap)
    snap0, _ = rh.snapshot_world(env)
    snap0["next_uid"] = rh.load_reference().uid_counter["next"]
    ow.load_world(0, snap0)
    # initial observation
    obs0 = ow.observe()[0, : len(env.agents)]
    ref0 = np.stack([a.state for a in env.agents]).astype(np.float32) if env.agents else np.zeros((0, 153), np.float32)
    assert np.array_equal(obs0, ref0), "%s: initial observation differs" % name

    def actions_fn(agents):
        if p_attack is None:
            return rng.randint(0, 8, size=len(agents))
        att = rng.random_sample(len(agents)) < p_attack
        return np.where(att, rng.randint(4, 8, size=len(agents)), rng.randint(0, 4, size=len(agents)))

    stats = {"agent_steps": 0, "vanished": 0, "births": 0, "max_pop": 0}
    for t in range(ticks):
        rec = rh.record_tick(env, actions_fn, cap, n_epi=t)
        n0 = len(rec["actions"])
        acts = np.zeros((1, cap), np.int8)
        acts[0, :n0] = rec["actions"]
        tape = ow.make_tape([rec["tape"]])
        ow.step(acts, tape)
        ps = rec["post_step"]
        n1 = len(ps["i"])
        tag = "%s tick %d step" % (name, t)
        want = dict(ps)
        compare_world(tag, ow.world(0), want, check_best=False)
        if int(ow.n_acted[0]) != n0:
            raise AssertionError("%s: n_acted" % tag)
        for key, got, ref in (("src", ow.src1[0, :n1], ps["src"]), ("done", ow.done[0, :n1], ps["done"]),
                              ("reward", ow.reward[0, :n1], ps["reward"].astype(np.float32)),
                              ("obs", ow.obs1[0, :n1], ps["obs"].astype(np.float32)),
                              ("l0_health", ow.l0_health[0, :n0], ps["l0_health"]),
                              ("l0_flags", ow.l0_flags[0, :n0], ps["l0_flags"]),
                              ("l0_reward", ow.l0_reward[0, :n0], ps["l0_reward"]),
                              ("l0_i", ow.l0_i[0, :n0], ps["l0_i"]), ("l0_j", ow.l0_j[0, :n0], ps["l0_j"])):
            if not np.array_equal(got, ref):
                bad = np.argwhere(got != ref)[:5]
                raise AssertionError("%s: %s differs at %s\n got  %s\n want %s" % (tag, key, bad.tolist(), got[tuple(bad[0])], ref[tuple(bad[0])]))
        ow.update(tape)
        pu = rec["post_update"]
        n2 = len(pu["i"])
        tag = "%s tick %d update" % (name, t)
        compare_world(tag, ow.world(0), pu, check_best=not static)
        for key, got, ref in (("src", ow.src2[0, :n2], pu["src"]), ("obs", ow.obs2[0, :n2], pu["obs"].astype(np.float32))):
            if not np.array_equal(got, ref):
                bad = np.argwhere(got != ref)[:5]
                raise AssertionError("%s: %s differs at %s" % (tag, key, bad.tolist()))
        stats["agent_steps"] += n0
        stats["vanished"] += n0 - n1
        stats["births"] += int((pu["src"] < 0).sum())
        stats["max_pop"] = max(stats["max_pop"], n2)
    if verbose:
        print("  ok  %-34s ticks=%-4d agent-steps=%-6d vanished=%-4d births=%-4d max_pop=%d" %
              (name, ticks, stats["agent_steps"], stats["vanished"], stats["births"], stats["max_pop"]))
    return stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    t0 = time.time()
    scale = 1 if args.quick else 4
    total = {"agent_steps": 0, "vanished": 0, "births": 0}
    cases = []
    for s in range(3 * scale):
        cases.append(dict(name="natural-static-%d" % s, seed=100 + s, ticks=150, n_brains=2 + s % 4))
        cases.append(dict(name="natural-nonstatic-%d" % s, seed=200 + s, ticks=150, n_brains=2 + s % 3, static=False))
        cases.append(dict(name="dense100-%d" % s, seed=300 + s, ticks=40, fill=100))
        cases.append(dict(name="dense100-nonstatic-%d" % s, seed=350 + s, ticks=40, fill=100, static=False, n_brains=3))
        cases.append(dict(name="dense200-%d" % s, seed=400 + s, ticks=25, fill=200, max_agents=100))
        cases.append(dict(name="dense300-attack-%d" % s, seed=500 + s, ticks=20, fill=300, max_agents=300, p_attack=0.6))
        cases.append(dict(name="small7x5-%d" % s, seed=600 + s, ticks=60, width=7, height=5, fill=12, max_agents=20))
        cases.append(dict(name="rect30x20-limit-%d" % s, seed=700 + s, ticks=60, width=30, height=20, fill=60,
                          limit=True, incentive=False, max_agents=80))
        cases.append(dict(name="movers-only-%d" % s, seed=800 + s, ticks=30, fill=250, max_agents=300, p_attack=0.0))
    for c in cases:
        st = run_case(**c)
        for k in total:
            total[k] += st[k]
    print("live parity PASSED: %d cases, %d agent-steps, %d vanish events, %d births in %.1fs" %
          (len(cases), total["agent_steps"], total["vanished"], total["births"], time.time() - t0))


if __name__ == "__main__":
    main()
