"""Whole-loop fixtures: the real reference driven by its real brains from seeds alone (build container only).

For each case the reference's brains are built, given repo-generated weights (gen_golden.model_weights), THEN
random / np.random / torch are seeded and the trainer-loop body without learn() is run
(ReinLife/Helpers/trainer.py:85-99: agent.get_action(n_epi) for every agent, env.step(), env.update_env(n_epi)).
Recorded per tick: the chosen actions and the world after step() and after update_env().  tests/test_hip_e2e_seeds.py
rebuilds the same loop on the product API with the same weights and seeds and must see the same worlds.

Data only: weights generated here, integer/float arrays observed from the reference.   python oracle/gen_golden_e2e.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gen_golden as gg  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402

OUT_DIR = gg.OUT_DIR
AGENT_KEYS = gg.AGENT_KEYS

# name -> (seed, ticks, static, [(kind, training)], width, height, max_agents)
CASES = {
    "e2e_static_mixed": (21, 160, True, [("DQN", False), ("D3QN", True), ("PERD3QN", True), ("PPO", True)], 30, 30, 100),
    "e2e_static_explore": (22, 120, True, [("DQN", True), ("PERD3QN", True)], 20, 15, 60),
    "e2e_nonstatic_greedy": (23, 200, False, [("PERD3QN", False), ("DQN", False)], 30, 30, 100),
}


def build_brains(specs, ticks):
    ref = rh.load_reference()
    torch = ref.torch
    brains, flats = [], []
    for idx, (kind, training) in enumerate(specs):
        sd = gg.model_weights(kind, 100 + idx)
        tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
        if kind == "DQN":
            b = ref.DQN(max_epi=ticks, training=training)
            b.agent.load_state_dict(tsd)
        elif kind in ("D3QN", "PERD3QN"):
            b = getattr(ref, kind)(training=training)
            b.eval_net.load_state_dict(tsd)
            b.target_net.load_state_dict(tsd)
        else:
            b = ref.PPO()
            b.model.load_state_dict(tsd)
        brains.append(b)
        flats.append(np.concatenate([sd[k].reshape(-1) for k in sd]).astype(np.float32))
    return brains, flats


def run_case(name):
    seed, ticks, static, specs, width, height, max_agents = CASES[name]
    brains, flats = build_brains(specs, ticks)
    rh.seed_all(seed)  # after construction: the constructors consume torch's generator
    env = rh.make_env(brains=brains, width=width, height=height, max_agents=max_agents, static_families=static)
    env.reset()
    cap = orc.slot_cap_for(max_agents, width * height)
    d = {"cfg": np.array([width, height, max_agents, len(brains), int(static), 0, 1, cap, ticks], np.int64),
         "seed": np.int64(seed), "kinds": np.array([orc.KIND_BY_NAME[k] for k, _ in specs], np.int32),
         "training": np.array([int(t) for _, t in specs], np.int32)}
    for idx, f in enumerate(flats):
        d["weights_%d" % idx] = f
    snaps = {"step": [], "upd": []}
    actions, rewards = [], []
    for t in range(ticks):
        for agent in env.agents:
            agent.get_action(t)
        actions.append(np.array([int(a.action) for a in env.agents], np.int8))
        env.step()
        s, ags = rh.snapshot_world(env)
        rewards.append(np.array([float(a.reward) for a in ags], np.float32))
        snaps["step"].append(s)
        env.update_env(t)
        snaps["upd"].append(rh.snapshot_world(env)[0])
    maxn = max(1, max(len(s["i"]) for ph in snaps.values() for s in ph), max(len(a) for a in actions))
    d["n0"] = np.array([len(a) for a in actions], np.int32)
    d["actions"] = np.stack([gg._pad(a, maxn) for a in actions])
    d["step_reward"] = np.stack([gg._pad(r, maxn) for r in rewards])
    for ph, lst in snaps.items():
        d[ph + "_n"] = np.array([len(s["i"]) for s in lst], np.int32)
        d[ph + "_cell_type"] = np.stack([s["cell_type"] for s in lst])
        for k in AGENT_KEYS:
            d[ph + "_" + k] = np.stack([gg._pad(s[k], maxn) for s in lst])
        d[ph + "_max_gene"] = np.array([s["max_gene"] for s in lst], np.int32)
        for k in ("best_uid", "best_fit", "best_brain"):
            d[ph + "_" + k] = np.stack([s[k] for s in lst])
    path = os.path.join(OUT_DIR, "%s.npz" % name)
    np.savez_compressed(path, **d)
    pop = d["upd_n"]
    print("wrote %s (%.0f KB): %d ticks, population %d..%d, %d agent-steps" % (path, os.path.getsize(path) / 1024, ticks, pop.min(), pop.max(), int(d["n0"].sum())))


if __name__ == "__main__":
    for name in (sys.argv[1:] or sorted(CASES)):
        run_case(name)
