"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the REAL reference (build container only).

    python oracle/gen_golden.py

Fixtures are data only: inputs (initial world, per-tick actions, RNG tape) and the outputs the reference produced
(post-step / post-update worlds, rewards, observations; network outputs for repo-generated weights).  No reference
source is stored.  tests/golden_io.py replays them through the oracle (CPU tests) and the HIP path (GPU tests).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
AGENT_KEYS = ("i", "j", "health", "age", "max_age", "gene", "brain", "uid", "flags", "action", "fitness")
DTYPES = {"i": np.uint8, "j": np.uint8, "health": np.int32, "age": np.int32, "max_age": np.int32, "gene": np.int32,
          "brain": np.int32, "uid": np.int32, "flags": np.uint8, "action": np.int8, "fitness": np.float64}


def _pad(a, n, fill=0):
    out = np.full((n,) + a.shape[1:], fill, dtype=a.dtype)
    out[: len(a)] = a
    return out


def record_trace(env, ticks, actions_fn, cap):
    """-> flat dict of arrays describing `ticks` trainer-loop ticks of the reference."""
    C = env.width * env.height
    snap0, agents0 = rh.snapshot_world(env)
    d = {"cfg": np.array([env.width, env.height, env.max_agents, len(env.brains), int(env.static_families),
                          int(env.limit_reproduction), int(env.incentivize_killing), cap, ticks], np.int64)}
    d["init_cell_type"] = snap0["cell_type"]
    d["init_n"] = np.int32(len(agents0))
    for k in AGENT_KEYS:
        d["init_" + k] = _pad(snap0[k], cap)
    d["init_next_uid"] = np.int32(rh.load_reference().uid_counter["next"])
    d["init_max_gene"] = np.int32(env.max_gene)
    for k in ("best_uid", "best_fit", "best_brain"):
        d["init_" + k] = snap0[k]
    d["init_obs"] = (np.stack([a.state for a in agents0]) if agents0 else np.zeros((0, 153))).astype(np.float32)

    env._ref_tracker = rh.make_ref_tracker(env, 10 ** 9)
    recs = [rh.record_tick(env, actions_fn, cap, n_epi=t) for t in range(ticks)]
    d["trk_tick"] = np.stack([r["post_step"]["trk_tick"] for r in recs])   # [T][G][7] Tracker per-tick values
    d["trk_pop"] = np.array([r["post_step"]["trk_pop"] for r in recs])
    maxn = max([1] + [max(len(r["actions"]), len(r["post_step"]["i"]), len(r["post_update"]["i"])) for r in recs])
    d["n0"] = np.array([len(r["actions"]) for r in recs], np.int32)
    d["actions"] = np.stack([_pad(r["actions"], cap) for r in recs])
    for k in ("food_k", "food_u", "repro_u", "birth_k", "produce_u", "produce_choice"):
        d["tape_" + k] = np.stack([np.asarray(r["tape"][k]) for r in recs])
    for phase, key in (("step", "post_step"), ("upd", "post_update")):
        d[phase + "_n"] = np.array([len(r[key]["i"]) for r in recs], np.int32)
        d[phase + "_cell_type"] = np.stack([r[key]["cell_type"] for r in recs])
        for k in AGENT_KEYS:
            d[phase + "_" + k] = np.stack([_pad(r[key][k], maxn) for r in recs])
        d[phase + "_src"] = np.stack([_pad(r[key]["src"], maxn, -1) for r in recs])
        d[phase + "_obs"] = np.stack([_pad(r[key]["obs"].astype(np.float32), maxn) for r in recs])
        d[phase + "_max_gene"] = np.array([r[key]["max_gene"] for r in recs], np.int32)
        for k in ("best_uid", "best_fit", "best_brain"):
            d[phase + "_" + k] = np.stack([r[key][k] for r in recs])
    d["step_learn_n"] = np.array([len(r["post_step"]["learn_k"]) for r in recs], np.int32)
    d["step_learn_k"] = np.stack([_pad(r["post_step"]["learn_k"], maxn, -1) for r in recs])  # Agent.learn call order
    d["step_reward"] = np.stack([_pad(r["post_step"]["reward"].astype(np.float32), maxn) for r in recs])
    d["step_done"] = np.stack([_pad(r["post_step"]["done"], maxn) for r in recs])
    for k in ("l0_health", "l0_flags", "l0_reward", "l0_i", "l0_j"):
        d["step_" + k] = np.stack([_pad(r["post_step"][k], maxn) for r in recs])
    return d


def random_actions(rng, p_attack=None):
    def fn(agents):
        if p_attack is None:
            return rng.randint(0, 8, size=len(agents))
        att = rng.random_sample(len(agents)) < p_attack
        return np.where(att, rng.randint(4, 8, size=len(agents)), rng.randint(0, 4, size=len(agents)))
    return fn


def trace_case(name, seed, ticks, n_brains=2, width=30, height=30, max_agents=100, static=True, limit=False,
               incentive=True, fill=0, p_attack=None):
    rh.seed_all(seed)
    env = rh.make_env(n_brains=n_brains, width=width, height=height, max_agents=max_agents, static_families=static,
                      limit_reproduction=limit, incentivize_killing=incentive)
    env.reset()
    rng = np.random.RandomState(seed + 999)
    if fill:
        rh.fill_agents(env, fill, rng)
    cap = orc.slot_cap_for(max(max_agents, fill), width * height)
    d = record_trace(env, ticks, random_actions(rng, p_attack), cap)
    path = os.path.join(OUT_DIR, "trace_%s.npz" % name)
    np.savez_compressed(path, **d)
    print("wrote %s (%.0f KB)" % (path, os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------------------------------------------
# micro worlds: one hand-placed tick per verified quirk (SURVEY.md 8a 'Verified quirks checklist')
# ----------------------------------------------------------------------------------------------------------------
def micro_env(agents, foods=(), width=30, height=30, n_brains=2, static=True, max_agents=100):
    """agents: list of (i, j, gene, dict of attribute overrides); foods: list of (i, j, 'Food'|'Poison'|'SuperFood')."""
    ref = rh.load_reference()
    env = rh.make_env(n_brains=n_brains, width=width, height=height, max_agents=max_agents, static_families=static)
    env.grid = ref.Grid(width, height)
    env.best_agents = []
    made = []
    for (i, j, gene, over) in agents:
        a = env._add_agent(coordinates=(i, j), brain=env.brains[gene % n_brains], gene=gene)
        for k, v in over.items():
            setattr(a, k, v)
        made.append(a)
    if not static:
        env.best_agents = [ref.env_mod.copy.deepcopy(made[0]) for _ in range(10)]
    for (i, j, kind) in foods:
        env.grid.set(i, j, getattr(ref, kind))
    env._get_observations()
    env._update_agents_state()
    return env


def micro_case(name, seed, agents, actions_by_pos, foods=(), **kw):
    """actions_by_pos: {(i,j): action} for the hand-placed agents."""
    rh.seed_all(seed)
    env = micro_env(agents, foods, **kw)
    cap = orc.slot_cap_for(env.max_agents, env.width * env.height)

    def fn(ags):
        return [actions_by_pos[(a.i, a.j)] for a in ags]

    d = record_trace(env, 1, fn, cap)
    path = os.path.join(OUT_DIR, "micro_%s.npz" % name)
    np.savez_compressed(path, **d)
    print("wrote %s" % path)


UP, RIGHT, DOWN, LEFT, A_UP, A_RIGHT, A_DOWN, A_LEFT = range(8)


def micro_cases():
    # follower-down vanishes: A(5,5) moves down into B(6,5) while B moves down to (7,5); cell(B) > cell(A)
    micro_case("follow_down_vanish", 1, [(5, 5, 0, {}), (6, 5, 1, {})], {(5, 5): DOWN, (6, 5): DOWN})
    # follower-up is fine: A(6,5) moves up into B(5,5)'s old cell, B earlier in order already left
    micro_case("follow_up_ok", 2, [(5, 5, 0, {}), (6, 5, 1, {})], {(5, 5): UP, (6, 5): UP})
    micro_case("follow_right_vanish", 3, [(5, 5, 0, {}), (5, 6, 1, {})], {(5, 5): RIGHT, (5, 6): RIGHT})
    micro_case("follow_left_ok", 4, [(5, 5, 0, {}), (5, 6, 1, {})], {(5, 5): LEFT, (5, 6): LEFT})
    # adjacent swap: the earlier agent vanishes
    micro_case("swap_earlier_vanishes", 5, [(5, 5, 0, {}), (5, 6, 1, {})], {(5, 5): RIGHT, (5, 6): LEFT})
    # mutual attack: earlier dies (health 0), later ends at 100
    micro_case("mutual_attack", 6, [(5, 5, 0, {}), (5, 6, 1, {})], {(5, 5): A_RIGHT, (5, 6): A_LEFT})
    # attacked agent steps on food afterwards: survives with 40
    micro_case("attacked_then_eats", 7, [(5, 5, 0, {}), (5, 6, 1, {})], {(5, 5): A_RIGHT, (5, 6): RIGHT},
               foods=[(5, 7, "Food")])
    # three-agent conflict chain: C->X, B->X conflict; A->B's cell then also reverts
    micro_case("conflict_chain", 8, [(5, 5, 0, {}), (5, 6, 1, {}), (4, 7, 0, {})],
               {(5, 5): RIGHT, (5, 6): RIGHT, (4, 7): DOWN})
    # health map float64 because (0,0) holds an agent; neighbours with fractional health
    micro_case("healthmap_float", 9, [(0, 0, 0, {"health": 130}), (1, 1, 1, {"health": 75}), (29, 29, 0, {"health": 200})],
               {(0, 0): A_UP, (1, 1): A_UP, (29, 29): A_UP})
    # health map int64-truncated otherwise
    micro_case("healthmap_int", 10, [(2, 2, 0, {"health": 130}), (3, 3, 1, {"health": 75}), (4, 4, 0, {"health": 210})],
               {(2, 2): A_UP, (3, 3): A_UP, (4, 4): A_UP})
    # poison drives health negative -> the agent shows 1.0 in the FOOD plane of its neighbours; dead kin in gene plane
    micro_case("negative_health_food_plane", 11, [(5, 5, 0, {"health": 30}), (5, 7, 0, {}), (6, 6, 1, {})],
               {(5, 5): RIGHT, (5, 7): A_UP, (6, 6): A_UP}, foods=[(5, 6, "Poison")])
    # super food: ate_super_food -1 -> 1.0, max_age 50 -> 60
    micro_case("super_food", 12, [(5, 5, 0, {})], {(5, 5): RIGHT}, foods=[(5, 6, "SuperFood")])
    # toroidal wrap: up from row 0, left from column 0, and attack through the wall
    micro_case("wrap_moves", 13, [(0, 0, 0, {}), (0, 5, 1, {}), (29, 5, 0, {}), (7, 0, 1, {}), (7, 29, 0, {})],
               {(0, 0): LEFT, (0, 5): A_UP, (29, 5): UP, (7, 0): A_LEFT, (7, 29): RIGHT})
    # alone in the world: reward is 0 (alive == 1)
    micro_case("alone_reward_zero", 14, [(9, 9, 0, {})], {(9, 9): UP})
    # old age death + corpse becomes food + dead kin visible in the kin plane during the step observation
    micro_case("old_age_death", 15, [(5, 5, 0, {"age": 49}), (5, 7, 0, {}), (7, 5, 1, {})],
               {(5, 5): A_UP, (5, 7): A_UP, (7, 5): A_UP})
    # kin attack sets inter_killed; non-kin sets intra_killed (names swapped in the reference, environment.py:696-699)
    micro_case("kill_flags", 16, [(5, 5, 0, {}), (5, 6, 0, {}), (8, 8, 0, {}), (8, 9, 1, {})],
               {(5, 5): A_RIGHT, (5, 6): UP, (8, 8): A_RIGHT, (8, 9): UP})
    # non-static: best agent replacement by a fitter live agent
    micro_case("best_agents_nonstatic", 17, [(5, 5, 0, {"fitness": 3.5}), (9, 9, 1, {"fitness": 7.25})],
               {(5, 5): UP, (9, 9): UP}, static=False)


# ----------------------------------------------------------------------------------------------------------------
# model forward vectors
# ----------------------------------------------------------------------------------------------------------------
def model_weights(kind_name, seed):
    """Repo-generated weights in state-dict order, nn.Linear-like scale (uniform +-1/sqrt(fan_in))."""
    shapes = {
        "DQN": [("fc1", 128, 153), ("fc2", 64, 128), ("fc3", 8, 64)],
        "D3QN": [("fc", 128, 153), ("adv_fc1", 128, 128), ("adv_fc2", 8, 128), ("value_fc1", 128, 128), ("value_fc2", 1, 128)],
        "PPO": [("fc1", 256, 153), ("fc2", 256, 256), ("fc_pi", 8, 256), ("fc_v", 1, 256)],
    }["D3QN" if kind_name == "PERD3QN" else kind_name]
    rng = np.random.RandomState(seed)
    sd = {}
    for name, n_out, n_in in shapes:
        b = 1.0 / np.sqrt(n_in)
        sd[name + ".weight"] = rng.uniform(-b, b, size=(n_out, n_in)).astype(np.float32)
        sd[name + ".bias"] = rng.uniform(-b, b, size=(n_out,)).astype(np.float32)
    return sd


def model_vectors():
    ref = rh.load_reference()
    torch = ref.torch
    # observation rows: a dense trace's post-update observations (realistic inputs) + a few random rows
    tr = np.load(os.path.join(OUT_DIR, "trace_dense100.npz"))
    rows = [tr["upd_obs"][t, : tr["upd_n"][t]] for t in range(0, len(tr["upd_n"]), 3)]
    rng = np.random.RandomState(7)
    rows.append(rng.uniform(-1, 1, size=(32, 153)).astype(np.float32))
    obs = np.concatenate(rows).astype(np.float32)[:640]
    out = {"obs": obs}
    for kind_name, seed in (("DQN", 11), ("D3QN", 12), ("PERD3QN", 13), ("PPO", 14)):
        sd = model_weights(kind_name, seed)
        flat = np.concatenate([sd[k].reshape(-1) for k in sd])
        tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
        with torch.no_grad():
            if kind_name == "DQN":
                brain = ref.DQN(training=False)
                brain.agent.load_state_dict(tsd)
                res = np.stack([brain.agent.forward(torch.from_numpy(o)).numpy() for o in obs])
                acts = np.array([brain.get_action(o.astype(np.float64), 0) for o in obs], np.int8)
            elif kind_name in ("D3QN", "PERD3QN"):
                brain = (ref.D3QN if kind_name == "D3QN" else ref.PERD3QN)(training=False)
                brain.eval_net.load_state_dict(tsd)
                res = np.stack([brain.eval_net.forward(torch.from_numpy(o[None])).numpy()[0] for o in obs])
                acts = np.array([brain.get_action(o.astype(np.float64), 0) for o in obs], np.int8)
            else:
                brain = ref.PPO()
                brain.model.load_state_dict(tsd)
                res = np.stack([brain.model.pi(torch.from_numpy(o)).numpy() for o in obs])
                acts = res.argmax(1).astype(np.int8)  # sampling uses torch's RNG; only the probabilities are pinned
        out[kind_name + "_weights"] = flat.astype(np.float32)
        out[kind_name + "_out"] = res.astype(np.float32)
        out[kind_name + "_greedy"] = acts
    path = os.path.join(OUT_DIR, "models.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.0f KB)" % (path, os.path.getsize(path) / 1024))


def reset_vectors():
    """Environment.reset() of the reference under fixed numpy seeds (environment.py:133-158): grid + agents."""
    out = {}
    for seed, n_brains, w, h in ((1, 2, 30, 30), (2, 3, 30, 30), (3, 5, 30, 20), (4, 2, 7, 5)):
        rh.seed_all(seed)
        env = rh.make_env(n_brains=n_brains, width=w, height=h)
        env.reset()
        snap, agents = rh.snapshot_world(env)
        key = "s%d_b%d_%dx%d" % (seed, n_brains, w, h)
        out[key + "_cell_type"] = snap["cell_type"]
        for k in ("i", "j", "gene", "uid"):
            out[key + "_" + k] = snap[k]
        out[key + "_obs"] = np.stack([a.state for a in agents]).astype(np.float32)
    path = os.path.join(OUT_DIR, "resets.npz")
    np.savez_compressed(path, **out)
    print("wrote %s" % path)


def state_dict_keys():
    """Names and shapes of the reference networks' state dicts (the load_model wire contract)."""
    import json
    ref = rh.load_reference()
    nets = {"DQN": ref.DQN().agent, "D3QN": ref.D3QN().eval_net, "PERD3QN": ref.PERD3QN().eval_net, "PPO": ref.PPO().model}
    out = {k: [[name, list(t.shape)] for name, t in net.state_dict().items()] for k, net in nets.items()}
    path = os.path.join(OUT_DIR, "state_dict_keys.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote %s" % path)
    # scalar brain attributes Saver._save_params writes to parameters_*.json (saver.py:170-194)
    import inspect
    brains = {"DQN": ref.DQN(max_epi=100), "D3QN": ref.D3QN(), "PERD3QN": ref.PERD3QN(), "PPO": ref.PPO()}
    params = {k: {n: v for n, v in inspect.getmembers(b, lambda a: not inspect.isroutine(a)) if type(v) in (float, int, bool, str) and not n.startswith("__")}  # no docstrings: data only
              for k, b in brains.items()}
    path = os.path.join(OUT_DIR, "brain_param_keys.json")
    json.dump(params, open(path, "w"), indent=1)
    print("wrote %s" % path)


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    if "--extras-only" in sys.argv:
        reset_vectors()
        state_dict_keys()
        return
    trace_case("natural_static", 11, 120, n_brains=3)
    trace_case("natural_nonstatic", 12, 150, n_brains=2, static=False)
    trace_case("dense100", 13, 30, fill=100)
    trace_case("dense100_nonstatic", 14, 30, fill=100, static=False, n_brains=3)
    trace_case("dense200_attack", 15, 12, fill=200, max_agents=100, p_attack=0.5)
    trace_case("movers250", 16, 10, fill=250, max_agents=300, p_attack=0.0)
    trace_case("small7x5", 17, 60, width=7, height=5, fill=12, max_agents=20)
    trace_case("rect30x20_limit", 18, 40, width=30, height=20, fill=60, limit=True, incentive=False, max_agents=80)
    micro_cases()
    model_vectors()
    reset_vectors()
    state_dict_keys()


if __name__ == "__main__":
    main()
