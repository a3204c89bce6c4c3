"""TEST INFRASTRUCTURE ONLY -- reference-import harness (runs in the build container, never on the GPU box).

Imports the *real* MaartenGr/ReinLife from /root/reference (read-only; nothing is copied) and records, per tick,
everything the oracle / HIP path need to reproduce a trajectory bit-exactly:

  * the pre-step world (type grid + row-major agent list),
  * the actions every agent took,
  * the RNG tape: the ordered values each random draw site returned
      np   : Grid.set_random  (grid.py:75 randint, grid.py:77 random)   via _add_food (environment.py:763-776),
             _reproduce (environment.py:515) and _produce (environment.py:539,545)
      py   : random.random() gates (environment.py:501, :528) and random.choice (environment.py:536,538,544)
  * the post-step world (env.agents order after step, rewards, done, state_prime),
  * the post-update world (env.agents order after update_env, state, max_gene, best_agents).

The recording is done by replacing the `np` / `random` / `copy` *module attributes* of the reference's modules with
thin proxies (the reference's files are not modified).  This module is imported only by oracle/gen_golden.py and
oracle/live_parity.py.
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

REFERENCE_ROOT = "/root/reference"

import copy as _copy
import random as _pyrandom

import numpy as _np

FLAG_DEAD, FLAG_REPRODUCED, FLAG_KILLED, FLAG_ATE_SUPER, FLAG_INTER, FLAG_INTRA = 1, 2, 4, 8, 16, 32
N_BEST = 10
OBS_DIM = 153
FOOD_TRIES = 7  # 3 food + 3 poison + 1 super food (environment.py:767-776)


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "ReinLife"))


class _Log:
    def __init__(self):
        self.entries = []
        self.ctx = None
        self.entity = None

    def add(self, kind, value, extra=None):
        self.entries.append((self.ctx, self.entity, kind, value, extra))

    def clear(self):
        self.entries = []


LOG = _Log()


class _NpRandomProxy:
    """Stands in for `np.random` inside the reference's grid.py / environment.py."""

    def randint(self, lo, hi=None, *a, **k):
        r = _np.random.randint(lo, hi, *a, **k)
        LOG.add("np.randint", int(r), int(hi) if hi is not None else None)
        return r

    def random(self, *a, **k):
        r = _np.random.random(*a, **k)
        LOG.add("np.random", float(r))
        return r

    def __getattr__(self, name):
        return getattr(_np.random, name)


class _NpProxy:
    random = _NpRandomProxy()

    def __getattr__(self, name):
        return getattr(_np, name)


class _PyRandomProxy:
    """Stands in for the `random` module inside the reference's environment.py."""

    def random(self):
        r = _pyrandom.random()
        LOG.add("py.random", float(r))
        return r

    def choice(self, seq):
        r = _pyrandom.choice(seq)
        LOG.add("py.choice", r, len(seq))
        return r

    def randint(self, a, b):
        r = _pyrandom.randint(a, b)
        LOG.add("py.randint", int(r), (a, b))
        return r

    def __getattr__(self, name):
        return getattr(_pyrandom, name)


class _CopyProxy:
    """deepcopy that marks copied Agents as 'not a grid agent' (uid -1): `x not in best_agents` is an identity test
    in the reference (environment.py:738), and the 10 initial best agents are copies (environment.py:149)."""

    def __init__(self, agent_cls):
        self._agent_cls = agent_cls

    def deepcopy(self, x):
        r = _copy.deepcopy(x)
        if isinstance(r, self._agent_cls):
            r._uid = -1
        return r

    def __getattr__(self, name):
        return getattr(_copy, name)


_REF = None


def load_reference():
    """Import the reference once, install the recorders, return a namespace of its public objects."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise RuntimeError("reference not present at %s (this harness only runs in the build container)" % REFERENCE_ROOT)
    sys.modules.setdefault("pygame", types.ModuleType("pygame"))  # render.py:3 imports pygame; never used headless
    import matplotlib

    matplotlib.use("Agg")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import torch

    torch.set_num_threads(1)
    from ReinLife.World import environment as env_mod
    from ReinLife.World import grid as grid_mod
    from ReinLife.World import entities as ent_mod
    from ReinLife.Models import DQN, D3QN, PERD3QN, PPO

    # --- recorders -------------------------------------------------------------------------------------------
    grid_mod.np = _NpProxy()
    env_mod.np = _NpProxy()
    env_mod.random = _PyRandomProxy()
    env_mod.copy = _CopyProxy(ent_mod.Agent)

    uid_counter = {"next": 0}
    orig_agent_init = ent_mod.Agent.__init__

    def agent_init(self, *a, **k):
        orig_agent_init(self, *a, **k)
        self._uid = uid_counter["next"]
        uid_counter["next"] += 1

    ent_mod.Agent.__init__ = agent_init

    orig_set_random = grid_mod.Grid.set_random

    def set_random(self, entity, p, **kwargs):
        prev = LOG.entity
        LOG.entity = entity.__name__
        try:
            return orig_set_random(self, entity, p, **kwargs)
        finally:
            LOG.entity = prev

    grid_mod.Grid.set_random = set_random

    def _ctx_wrap(name, ctx):
        orig = getattr(env_mod.Environment, name)

        def wrapped(self, *a, **k):
            prev = LOG.ctx
            LOG.ctx = ctx
            try:
                return orig(self, *a, **k)
            finally:
                LOG.ctx = prev

        setattr(env_mod.Environment, name, wrapped)

    _ctx_wrap("_add_food", "food")
    _ctx_wrap("_reproduce", "repro")
    _ctx_wrap("_produce", "produce")
    _ctx_wrap("_init_food", "init_food")

    _REF = types.SimpleNamespace(
        Environment=env_mod.Environment, Grid=grid_mod.Grid, Agent=ent_mod.Agent, Food=ent_mod.Food,
        Poison=ent_mod.Poison, SuperFood=ent_mod.SuperFood, Empty=ent_mod.Empty, DQN=DQN, D3QN=D3QN, PERD3QN=PERD3QN,
        PPO=PPO, uid_counter=uid_counter, env_mod=env_mod, grid_mod=grid_mod, ent_mod=ent_mod, torch=torch)
    return _REF


def seed_all(seed):
    ref = load_reference()
    _pyrandom.seed(seed)
    _np.random.seed(seed)
    ref.torch.manual_seed(seed)


class NullBrain:
    """A brain that is never asked for an action (the harness writes agent.action itself); learn() records what
    Agent.learn hands over (entities.py:194-208)."""

    def __init__(self, method="DQN"):
        self.method = method
        self.learned = []

    def learn(self, **kwargs):
        self.learned.append(kwargs)

    def apply_gaussian_noise(self):
        pass


def make_env(n_brains=2, width=30, height=30, max_agents=100, static_families=True, limit_reproduction=False,
             incentivize_killing=True, brains=None):
    ref = load_reference()
    if brains is None:
        brains = [NullBrain() for _ in range(n_brains)]
    for k, b in enumerate(brains):
        b._tmpl = k  # survives deepcopy: identifies which initial brain a copied brain descends from
    ref.uid_counter["next"] = 0
    env = ref.Environment(width=width, height=height, max_agents=max_agents, brains=brains, training=False,
                          print_results=False, static_families=static_families,
                          limit_reproduction=limit_reproduction, incentivize_killing=incentivize_killing)
    return env


def fill_agents(env, n_total, rng):
    """SURVEY 8c recipe 5: top the world up to n_total agents at random empty cells."""
    ref = load_reference()
    while len(env.grid.get_entities(3)) < n_total:
        g = int(rng.randint(0, len(env.brains)))
        env._add_agent(random_loc=True, brain=env.brains[g], gene=g)
    env._get_observations()
    env._update_agents_state()


# ------------------------------------------------------------------------------------------------------------------
# snapshots
# ------------------------------------------------------------------------------------------------------------------
def _agent_flags(a):
    f = 0
    if a.dead:
        f |= FLAG_DEAD
    if a.reproduced:
        f |= FLAG_REPRODUCED
    if a.killed:
        f |= FLAG_KILLED
    if a.ate_super_food == 1.0:
        f |= FLAG_ATE_SUPER
    if a.inter_killed:
        f |= FLAG_INTER
    if a.intra_killed:
        f |= FLAG_INTRA
    return f


def snapshot_agents(agents):
    n = len(agents)
    d = {
        "i": _np.array([a.i for a in agents], dtype=_np.uint8).reshape(n),
        "j": _np.array([a.j for a in agents], dtype=_np.uint8).reshape(n),
        "health": _np.array([a.health for a in agents], dtype=_np.int32).reshape(n),
        "age": _np.array([a.age for a in agents], dtype=_np.int32).reshape(n),
        "max_age": _np.array([a.max_age for a in agents], dtype=_np.int32).reshape(n),
        "gene": _np.array([a.gene for a in agents], dtype=_np.int32).reshape(n),
        "brain": _np.array([getattr(a.brain, "_tmpl", -1) for a in agents], dtype=_np.int32).reshape(n),
        "uid": _np.array([a._uid for a in agents], dtype=_np.int32).reshape(n),
        "flags": _np.array([_agent_flags(a) for a in agents], dtype=_np.uint8).reshape(n),
        "action": _np.array([a.action for a in agents], dtype=_np.int8).reshape(n),
        "fitness": _np.array([float(a.fitness) for a in agents], dtype=_np.float64).reshape(n),
    }
    return d


def snapshot_world(env):
    """Type grid + on-grid agents in row-major order (= Grid.get_entities order, grid.py:60-67)."""
    grid = _np.array([[int(env.grid.grid[i, j].entity_type) for j in range(env.width)] for i in range(env.height)],
                     dtype=_np.uint8)
    agents = [env.grid.grid[i, j] for i in range(env.height) for j in range(env.width)
              if int(env.grid.grid[i, j].entity_type) == 3]
    d = snapshot_agents(agents)
    d["cell_type"] = grid.reshape(-1)
    d["max_gene"] = _np.int32(env.max_gene)
    best = env.best_agents
    d["best_uid"] = _np.array([b._uid for b in best] + [-1] * (N_BEST - len(best)), dtype=_np.int32)
    d["best_fit"] = _np.array([float(b.fitness) for b in best] + [0.0] * (N_BEST - len(best)), dtype=_np.float64)
    d["best_brain"] = _np.array([getattr(b.brain, "_tmpl", 0) for b in best] + [0] * (N_BEST - len(best)),
                                dtype=_np.int32)
    return d, agents


def _states(agents, attr):
    if not agents:
        return _np.zeros((0, OBS_DIM), dtype=_np.float64)
    return _np.stack([_np.asarray(getattr(a, attr), dtype=_np.float64) for a in agents])


def extract_tape(entries, slot_cap, best_agents):
    """Turn the ordered log of one tick into the per-site tape arrays (SURVEY 8c 'RNG tape contract')."""
    food_k = _np.full(FOOD_TRIES, -1, dtype=_np.int32)
    food_u = _np.full(FOOD_TRIES, 2.0, dtype=_np.float64)
    repro_u = _np.full(slot_cap, -1.0, dtype=_np.float64)
    birth_k = _np.full(slot_cap + 1, -1, dtype=_np.int32)
    produce_u = _np.float64(-1.0)
    produce_choice = _np.int32(-1)
    per_type = {"Food": 0, "Poison": 0, "SuperFood": 0}
    base = {"Food": 0, "Poison": 3, "SuperFood": 6}
    n_repro = 0
    n_birth = 0
    cur_food_slot = None
    for ctx, entity, kind, value, extra in entries:
        if ctx == "food":
            if kind == "np.randint":
                cur_food_slot = base[entity] + per_type[entity]
                per_type[entity] += 1
                food_k[cur_food_slot] = value
            elif kind == "np.random":
                food_u[cur_food_slot] = value
        elif ctx == "repro":
            if kind == "py.random":
                repro_u[n_repro] = value
                n_repro += 1
            elif kind == "np.randint":
                birth_k[n_birth] = value
                n_birth += 1
            elif kind == "np.random":
                assert value < 1.0
            else:
                raise AssertionError("unexpected draw in _reproduce: %s" % kind)
        elif ctx == "produce":
            if kind == "py.random":
                produce_u = _np.float64(value)
            elif kind == "py.choice":
                if isinstance(value, (int, _np.integer)):
                    produce_choice = _np.int32(value)  # static families: the chosen gene (environment.py:536,538)
                else:  # non-static: an Agent out of best_agents (environment.py:544) -> its list index
                    idx = [k for k, b in enumerate(best_agents) if b is value]
                    produce_choice = _np.int32(idx[0])
            elif kind == "np.randint":
                birth_k[n_birth] = value
                n_birth += 1
            elif kind == "np.random":
                assert value < 1.0
    return {"food_k": food_k, "food_u": food_u, "repro_u": repro_u, "birth_k": birth_k,
            "produce_u": _np.array(produce_u), "produce_choice": _np.array(produce_choice),
            "n_repro_draws": _np.int32(n_repro), "n_births_drawn": _np.int32(n_birth)}


TRACKER_VARS = ("Avg Population Size", "Avg Population Age", "Avg Population Fitness", "Best Population Age",
                "Avg Number of Attacks", "Avg Number of Kills", "Avg Number of Intra Kills")


def make_ref_tracker(env, update_interval):
    """A real reference Tracker (ReinLife/Helpers/tracker.py) configured like Environment.__init__ does (env.py:125-131)."""
    from ReinLife.Helpers.tracker import Tracker
    return Tracker(update_interval=update_interval, interactive=False, print_results=False, google_colab=False,
                   nr_genes=len(env.brains), static_families=env.static_families, brains=env.brains)


def tracker_tick_values(tracker):
    """Last appended per-tick values as [G][7] + number of populations."""
    G = tracker.nr_genes
    vals = _np.array([[tracker.track_results[v][g][-1] for v in TRACKER_VARS] for g in range(G)], dtype=_np.float64)
    return vals, float(tracker.track_results["Avg Number of Populations"][-1])


def record_tick(env, actions_fn, slot_cap, n_epi=0):
    """Run one trainer-loop tick (trainer.py:85-99 minus learn/render) on the real reference, recording everything.

    actions_fn(agents) -> sequence of ints, one per agent of env.agents (the L0 list)."""
    pre, l0 = snapshot_world(env)
    assert [id(a) for a in l0] == [id(a) for a in env.agents], "env.agents is not the row-major grid list"
    state_l0 = _states(l0, "state")
    acts = list(actions_fn(l0))
    for a, act in zip(l0, acts):
        a.action = int(act)

    LOG.clear()
    env.step()
    step_entries = list(LOG.entries)
    l1 = list(env.agents)
    post_step, l1_check = snapshot_world(env)
    assert [id(a) for a in l1] == [id(a) for a in l1_check]
    idx0 = {id(a): k for k, a in enumerate(l0)}
    post_step["src"] = _np.array([idx0[id(a)] for a in l1], dtype=_np.int16)
    post_step["reward"] = _np.array([float(a.reward) for a in l1], dtype=_np.float64)
    post_step["done"] = _np.array([1 if a.done else 0 for a in l1], dtype=_np.uint8)
    post_step["obs"] = _states(l1, "state_prime")
    # L0-level results incl. vanished agents (W5 rewards count them)
    post_step["l0_health"] = _np.array([a.health for a in l0], dtype=_np.int32)
    post_step["l0_flags"] = _np.array([_agent_flags(a) for a in l0], dtype=_np.uint8)
    post_step["l0_reward"] = _np.array([float(a.reward) for a in l0], dtype=_np.float64)
    post_step["l0_i"] = _np.array([a.i for a in l0], dtype=_np.uint8)
    post_step["l0_j"] = _np.array([a.j for a in l0], dtype=_np.uint8)
    post_step["l0_fitness"] = _np.array([float(a.fitness) for a in l0], dtype=_np.float64)

    # trainer.py:95-96: every agent of the post-step list is asked to learn; Agent.learn filters age > 1
    for b in env.brains:
        if hasattr(b, "learned"):
            del b.learned[:]
    order = []
    orig_learn = {}
    for k, a in enumerate(l1):
        before = {id(b): len(getattr(b, "learned", ())) for b in [a.brain]}
        a.learn(n_epi=n_epi) if a.brain.method not in ("DQN", "PPO", "A2C", "PERDQN") else a.learn()
        if hasattr(a.brain, "learned") and len(a.brain.learned) > before[id(a.brain)]:
            kw = a.brain.learned[-1]
            assert kw["state"] is a.state and kw["state_prime"] is a.state_prime and kw["action"] == a.action
            assert kw["reward"] == a.reward and kw["done"] == a.done and kw["age"] == a.age and kw["dead"] == a.dead
            order.append(k)
    post_step["learn_k"] = _np.array(order, dtype=_np.int16)  # post-step list indices handed to brain.learn, in call order

    trk = getattr(env, "_ref_tracker", None)
    if trk is not None:  # environment.py:206-207 calls tracker.update_results(self.agents, n_epi) first thing in update_env
        trk._track_results(env.agents)
        post_step["trk_tick"], pops = tracker_tick_values(trk)
        post_step["trk_pop"] = _np.float64(pops)

    LOG.clear()
    env.update_env(n_epi)
    upd_entries = list(LOG.entries)
    l2 = list(env.agents)
    post_update, l2_check = snapshot_world(env)
    assert [id(a) for a in l2] == [id(a) for a in l2_check]
    idx1 = {id(a): k for k, a in enumerate(l1)}
    post_update["src"] = _np.array([idx1.get(id(a), -1) for a in l2], dtype=_np.int16)
    post_update["obs"] = _states(l2, "state")

    # best_agents indices refer to the list *after* _update_best_agents (environment.py:209 precedes :212)
    tape = extract_tape(step_entries + upd_entries, slot_cap, list(env.best_agents))
    return {"pre": pre, "state_l0": state_l0, "actions": _np.array(acts, dtype=_np.int8), "tape": tape,
            "post_step": post_step, "post_update": post_update}
