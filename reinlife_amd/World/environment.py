"""Environment: the reference's protocol (ReinLife/World/environment.py:16-215) over the HIP world kernels.

Same constructor keywords, `reset() / step() / update_env(n_epi)`, `agents` (row-major list), `brains`, `max_gene`,
`action_space = 8`, `observation_space = 153`.  Extras (keyword-only): `n_worlds` independent replicas on the device,
`device`, `seed`.  `env.agents` are lightweight views of world 0, built from the device state when they are first read after a change (a loop
that never reads them never synchronises); the batched fast paths are `env.act(n_epi)` (Agent.get_action for every agent of every
world on the GPU) and `env.run(n_epi, n_ticks)` (whole chunks of the trainer loop in one launch each).

Random numbers.  `rng="reference"` (the default for a single world) makes the SAME draws from Python's `random`,
`np.random` and torch's global generator, in the same order and with the same arguments, as the reference's
reset() / step() / update_env() (environment.py:133-158, 488-547, 741-776; grid.py:69-83): seed the three generators as
you would for the reference and the world follows the reference's trajectory bit for bit (tests/test_hip_seed_compat.py
replays the golden traces from their seeds alone).  The draws of _add_food depend on the grid after movement, so a step
is two launches (rl_step_split -> host draws -> rl_step_food).  `rng="philox"` (the default and only choice for
n_worlds > 1) draws inside the kernels from counter-based Philox streams keyed by `seed`.  The Tracker's statistics are accumulated in the step kernel (Helpers/tracker.py); Saver and a headless painter (Helpers/render.py) mirror the reference's; pygame windows are out of scope.
"""
import os

import numpy as np
import torch

from .. import _lib
from ..worlds import DeviceWorlds
from .utils import Actions, EntityTypes


_A_FIELDS = tuple(n for n in _lib.STATE_FIELDS if n.startswith("a_"))


class _WorldMirror:
    """Host copy of ONE world's agent list and per-agent outputs, in env.agents order.  Built on first use after a step() /
    update_env() / run() (Environment._mat): a loop that never looks at env.agents never pays for it.  The small arrays come with
    the state snapshot; the observation rows (Agent.state / state_prime: 612 B per agent) are fetched only when somebody reads them."""

    def __init__(self, env, world, phase):
        self.env, self.world, self.phase = env, world, phase
        self.stamp = env._stamp     # the device state this mirror describes (Environment._refresh moves on)
        self.n = 0
        self.host = None            # {"a_i": ..., ...}
        self._state = None          # [n,153] float64: Agent.state
        self._state_prime = None    # [n,153] or None (between update_env and the next step)
        self.reward = self.done = self.src1 = None
        self.actions = None         # [cap] int8
        self.q = None               # [cap,8] float32: the policy's outputs for this tick (Environment.act, rng="reference")
        self.dirty = False          # actions were set through AgentView.action and are not on the device yet
        self.views = None           # the AgentView list handed out by agents_of()

    def _rows(self, t):
        if self.stamp != self.env._stamp:
            raise RuntimeError("this AgentView describes an earlier tick: Agent.state / state_prime must be read before the next "
                               "step() / update_env() / run() (take env.agents again)")
        return t.cpu().numpy().astype(np.float64)

    @property
    def state(self):
        if self._state is None:
            w, n = self.env.worlds, self.n
            if self.phase == "step":   # Agent.state is still the observation the policy read, in the pre-step order
                self._state = self._rows(w.obs_state()[self.world])[self.src1[:n].astype(np.int64)]
            else:
                self._state = self._rows(w.obs_state()[self.world, :n])
        return self._state

    @property
    def state_prime(self):
        if self._state_prime is None and self.phase == "step":
            self._state_prime = self._rows(self.env.worlds.obs_state_prime()[self.world, :self.n])
        return self._state_prime


class AgentView:
    """One entry of env.agents / env.agents_of(world): the reference's Agent attributes (entities.py:131-170) read from the
    device state of that world."""

    def __init__(self, env, k, mirror=None):
        self._env, self._k = env, k
        self._m = mirror if mirror is not None else env._mat(0)
        self.prob = None
        self.info = ""

    def _f(self, name):
        return self._m.host["a_" + name][self._k]

    i = property(lambda s: int(s._f("i")))
    j = property(lambda s: int(s._f("j")))
    coordinates = property(lambda s: [s.i, s.j])
    health = property(lambda s: int(s._f("health")))
    max_health = 200
    age = property(lambda s: int(s._f("age")))
    max_age = property(lambda s: int(s._f("max_age")))
    gene = property(lambda s: int(s._f("gene")))
    fitness = property(lambda s: float(s._f("fitness")))
    dead = property(lambda s: bool(s._f("flags") & _lib.F_DEAD))
    reproduced = property(lambda s: bool(s._f("flags") & _lib.F_REPRODUCED))
    killed = property(lambda s: int(bool(s._f("flags") & _lib.F_KILLED)))
    ate_super_food = property(lambda s: 1.0 if s._f("flags") & _lib.F_ATE_SUPER else -1)
    brain = property(lambda s: s._env.brains[int(s._f("brain"))])
    entity_type = EntityTypes.agent

    @property
    def action(self):
        if self._m.actions is None:   # act() ran after this view was built: the policy's choice is fetched on first use
            self._env._mat(self._m.world)
        return int(self._m.actions[self._k])

    @action.setter
    def action(self, a):
        if self._m.actions is None:
            self._env._mat(self._m.world)
        self._m.actions[self._k] = int(a)
        self._m.dirty = True

    @property
    def state(self):
        return self._m.state[self._k]

    @property
    def state_prime(self):
        return self._m.state_prime[self._k] if self._m.state_prime is not None else self.state

    @property
    def reward(self):
        return float(self._m.reward[self._k]) if self._m.reward is not None else None

    @property
    def done(self):
        return bool(self._m.done[self._k]) if self._m.done is not None else False

    def get_action(self, n_epi, out=None):  # entities.py:215-222
        b = self.brain
        state = self.state if out is None else None   # (with the outputs at hand the brain does not look at the row: no fetch)
        if b.method == "PPO":
            r = b.get_action(state, out=out)
            self.action, self.prob = r if isinstance(r, tuple) else (r, None)
        else:
            self.action = b.get_action(state, n_epi, out=out)

    def learn(self, **kwargs):
        """entities.py:194-208: the keyword set the reference hands to brain.learn, per brain method (caller kwargs such as
        n_epi only reach the brains the reference forwards them to).  The brains of this build ignore learn() (training is
        outside its scope); a subclass that overrides learn() receives the reference's arguments."""
        if self.age > 1:
            common = dict(age=self.age, dead=self.dead, action=self.action, state=self.state, reward=self.reward,
                          state_prime=self.state_prime, done=self.done)
            method = self.brain.method
            if method == "PPO":
                self.brain.learn(prob=self.prob, **common)
            elif method in ("DQN", "A2C", "PERDQN"):
                self.brain.learn(**common)
            else:  # DRQN, D3QN, PERD3QN, ...
                self.brain.learn(**common, **kwargs)


class Environment:
    def __init__(self, width=30, height=30, brains=None, grid_size=16, max_agents=50, update_interval=500, print_results=True,
                 static_families=True, interactive_results=False, google_colab=False, training=True, save=False,
                 pastel_colors=False, limit_reproduction=False, incentivize_killing=True, *, n_worlds=1, device=None,
                 seed=0, rng=None, synthetic_agents=None, refill_below=None, dist=None, world_base=None):
        if not brains:
            raise ValueError("Environment needs a non-empty list of brains")
        self.width, self.height = width, height
        self.actions, self.entities = Actions, EntityTypes
        self.brains = brains
        self.max_agents = max_agents
        self.max_gene = len(brains)
        self.static_families = static_families
        self.training = training
        self.save = save
        self.google_colab = google_colab
        self.limit_reproduction = limit_reproduction
        self.incentivize_killing = incentivize_killing
        self.action_space = 8
        self.observation_space = 153
        self.n_worlds = n_worlds
        # Replica sharding: under an initialised torch.distributed process group (passed as `dist`, or found) this
        # process is ONE rank of a job whose replicas are `n_worlds` per rank -- global replica ids rank * n_worlds .. + n_worlds - 1
        # key the Philox streams (results do not depend on the number of ranks), the device defaults to cuda:LOCAL_RANK, and the
        # Tracker pools its per-interval statistics over all ranks with ONE all-gather per closed interval (+ one layout check when it is built).
        self.dist, self.rank, self.world_size = resolve_dist(dist)
        self.world_base = int(world_base) if world_base is not None else self.rank * n_worlds
        if device is None:
            device = "cuda:%d" % (int(os.environ.get("LOCAL_RANK", "0")) if self.dist is not None else 0)
        self.device = device
        # rng="reference" means "this process's one world consumes the process-global generators like the reference does": that cannot
        # be a shard of a replicated job (every rank would build the same world from the same seeds and the Tracker would pool
        # duplicates), so under a process group of several ranks, or at a non-zero world_base, the default is the keyed Philox streams
        sharded = self.world_size > 1 or self.world_base != 0
        self.rng = rng or ("reference" if (n_worlds == 1 and not sharded) else "philox")
        if self.rng not in ("reference", "philox") or (self.rng == "reference" and n_worlds != 1):
            raise ValueError("rng must be 'reference' (single world only) or 'philox'")
        if self.rng == "reference" and sharded:
            raise ValueError("rng='reference' draws from the process-global generators and ignores world_base: it cannot be one shard of "
                             "a replicated job (world_size %d, world_base %d) -- use rng='philox'" % (self.world_size, self.world_base))
        # SURVEY.md 8d's synthetic benchmark worlds through this API (keyword-only extras, rng="philox"): reset() fills every world with
        # `synthetic_agents` agents at random cells (gene / brain uniform over the brains) instead of one agent per brain, and a world
        # whose population drops below `refill_below` after update_env is re-generated (what bench.py times: 100 / 70)
        self.synthetic_agents, self.refill_below = synthetic_agents, refill_below
        if (synthetic_agents is not None or refill_below is not None) and self.rng != "philox":
            raise ValueError("synthetic_agents / refill_below need rng='philox'")
        if refill_below is not None and synthetic_agents is None:
            raise ValueError("refill_below needs synthetic_agents (the population a re-generated world starts with)")
        self.best_agents = []
        # environment.py:118: the painter is built first (pastel colours draw from `random` here, its background tiles at
        # the first render() -- both matter for same-seed runs)
        from ..Helpers.render import Visualize
        self.viz = Visualize(self.width, self.height, grid_size, pastel=pastel_colors)
        self.frame = None
        self.worlds = DeviceWorlds(n_worlds=n_worlds, width=width, height=height, max_agents=max_agents,
                                   n_brains=len(brains), static_families=static_families,
                                   limit_reproduction=limit_reproduction, incentivize_killing=incentivize_killing, seed=seed,
                                   device=device, world_base=self.world_base)
        # results tracker (environment.py:125-131); numeric part only, fed by the step kernel
        from ..Helpers.tracker import Tracker
        self.tracker = Tracker(update_interval=update_interval, interactive=False, print_results=print_results, nr_genes=len(brains),
                               static_families=static_families, brains=brains, worlds=self.worlds if training else None,
                               dist=self.dist)
        self._mirrors = {}          # world -> _WorldMirror, built on demand (_mat): env.agents is world 0's
        self._stamp = 0             # counts the changes of the device state (_refresh)
        self._n_empty = None        # rng="reference": empty cells after this tick's _add_food, as far as the host's draws say
        self._grid = self._max_gene = None
        self._phase = "update"
        self._acted = False         # act() ran since the last step() / update_env(): worlds.actions holds the policy's choice
        self._empty = True          # no world has been built yet (before reset())
        self._brains_bound, self._bound_key = False, None
        self._ticks_since_check = 0
        self.loop_seconds = None    # wall time of the last trainer() / tester() loop on this environment (set by them)
        if training:
            warn_inference_only()

    # -- the host view of the device state: materialised on first read after every state change ------------------------------
    @property
    def agents(self):
        """env.agents (environment.py:186, 214): world 0's agents in row-major order, as AgentView objects."""
        return [] if self._empty else self._mat(0).views

    @property
    def grid(self):
        """Grid.get_numpy() of world 0 (grid.py:49-57): [height, width] uint8 cell codes."""
        if self._grid is None and not self._empty:
            self._sync()
            self._grid = self.worlds.s["cell_type"][0].cpu().numpy().reshape(self.height, self.width)
        return self._grid

    @property
    def max_gene(self):
        if self._max_gene is None:
            self._sync()
            self._max_gene = int(self.worlds.s["max_gene"][0].item())
        return self._max_gene

    @max_gene.setter
    def max_gene(self, v):
        self._max_gene = int(v)

    # world 0's mirror under the names the rest of this class uses
    _mirror0 = property(lambda s: s._mat(0))
    _host = property(lambda s: s._mat(0).host)
    _state_host = property(lambda s: s._mat(0).state)

    # -- brains -------------------------------------------------------------------------------------------------
    def _bind_brains(self):
        """The brains' packed weights and current exploration rates on the device.  Weights are packed and uploaded when they have
        changed (packed_weights() keys its cache on the parameters' storage and version counters); otherwise only the rates move."""
        packed = [b.packed_weights(self.device) for b in self.brains]
        key = tuple((b.kind, id(p)) for b, p in zip(self.brains, packed))   # (a repack makes a new tensor object)
        eps = [float(getattr(b, "epsilon", 0.0)) for b in self.brains]
        if self._brains_bound and key == self._bound_key:
            self.worlds._set_epsilons(eps)
            return
        self.worlds.set_brains([(b.kind, e, p) for b, e, p in zip(self.brains, eps, packed)])
        self._brains_bound, self._bound_key = True, key

    def _epsilon_schedule(self, n_epi, k):
        """[k, n_brains] float32: the brains' exploration rates in episodes n_epi .. n_epi + k - 1 (their own update rules:
        DQN.py:67-69, D3QN.py:84-89); leaves every brain in the state it has after episode n_epi + k - 1."""
        eps = np.empty((k, len(self.brains)), np.float32)
        for b, brain in enumerate(self.brains):
            sched = getattr(brain, "epsilon_schedule", None)
            if sched is not None:
                eps[:, b] = sched(n_epi, k)
            else:   # a brain class of the caller's: its own update rule, episode by episode
                for t in range(k):
                    brain.update_epsilon(n_epi + t)
                    eps[t, b] = getattr(brain, "epsilon", 0.0)
        return eps

    # -- reference protocol ---------------------------------------------------------------------------------------
    def reset(self):
        """environment.py:133-158.  World 0 follows the reference's np.random draw order exactly; further replicas are built
        by the same rule on the device (rl_reset_families)."""
        if self.synthetic_agents is not None:
            self.worlds.reset_synthetic(self.synthetic_agents)
            self._empty = False
            self._refresh(after="update")
            return
        # replicas: the same construction (one agent per brain, gene = its index; environment.py:147-149) on the device, from the
        # Philox streams keyed by (seed, global replica id): ONE launch for any number of worlds ...
        if self.rng == "philox" and (self.n_worlds > 1 or self.world_base != 0):
            self.worlds.reset_families()
        # ... and the job's replica 0 alone (world 0 of the rank with world_base 0) consumes the process-global np.random, in the
        # reference's order: a sharded job builds the same replicas whatever the number of ranks.  (rng="reference" -- one world per
        # process, the reference's own generator calls -- always does.)
        if self.world_base == 0 or self.rng == "reference":
            snap = host_reset(self.width, self.height, len(self.brains))
            self.worlds.load_world(0, snap)
            self.worlds.observe()
        self._empty = False
        self._refresh(after="update")

    def act(self, n_epi=0):
        """Agent.get_action for every agent of every world, batched on the GPU (trainer.py:88-89).

        rng="reference": one batched forward pass per brain, then the brains' own get_action rules agent by agent in
        list order, so the epsilon-greedy coins / categorical samples are the reference's generator calls in the
        reference's order (the forward pass draws nothing, so batching it changes no draw).
        rng="philox": forward pass, epsilon-greedy / categorical selection and the draws all happen in the kernel."""
        if self.rng == "reference":
            # ONE policy launch for all agents (rl_policy_act: per-brain row lists built on the device, every brain kind side by side),
            # queued before the host has looked at anything; its outputs travel with the state snapshot (one copy, one wait).  The
            # kernel's own action choice is overwritten below: the brains' rules run on the host, in list order, on these outputs.
            self._bind_brains()
            self.worlds.act(want_q=True)
            self._acted = True
            m = self._mat(0)
            if m.q is None:   # (the mirror was built before this call: somebody read env.agents first)
                m.q = self.worlds.out_q[0].cpu().numpy()
            q, qt = m.q, None
            for k, agent in enumerate(m.views):
                if agent.brain.method == "PPO":   # (Categorical(prob).sample() wants a tensor; argmax rules take the numpy row)
                    qt = torch.from_numpy(q) if qt is None else qt
                    agent.get_action(n_epi, out=qt[k])
                else:
                    agent.get_action(n_epi, out=q[k])
            return
        for b in self.brains:
            b.update_epsilon(n_epi)
        self._bind_brains()
        self.worlds.act()
        self._acted = True
        for m in self._mirrors.values():   # mirrors built before this call: their actions are the policy's choice now
            m.actions = None               # (fetched on first use: AgentView.action)
            m.dirty = False

    def step(self):
        """environment.py:160-186"""
        dirty = [m for m in self._mirrors.values() if m.dirty]
        if dirty:
            if self.n_worlds == 1:   # (the mirror holds the whole row: nothing to read back)
                full = dirty[0].actions[None]
            else:
                full = self.worlds.actions.cpu().numpy()
                for m in dirty:
                    full[m.world] = m.actions
            for m in dirty:
                m.dirty = False
            self.worlds.set_actions(full)
        if self.rng == "reference":
            # the draws of _add_food need the grid after movement: first half, ONE snapshot (the counts AND the post-step agent list --
            # update_env's draws are made from it without another look at the device), host draws, second half
            self.worlds.step_split()
            snap = self._snapshot()
            nf, npo, ns, ne = (int(x) for x in snap["pre_counts"][0])
            tape, self._n_empty = self._draw_add_food(nf, npo, ns, ne)
            self.worlds.step_food(self.worlds.make_tape([tape]))
            self._refresh(after="step")
            self._mirrors[0] = self._build_mirror(0, snap)   # (its cell_type predates the food: env.grid fetches its own)
        else:
            self.worlds.step()
            self._refresh(after="step")

    def update_env(self, n_epi=0):
        """environment.py:188-215"""
        if self.training:
            if self.tracker.update_results(None, n_epi):   # (the per-tick statistics were accumulated inside the step launch)
                self._ticks_since_check = 0                # an interval closed: the error flag came back with its sums (Tracker.resolve)
        self._ticks_since_check += 1
        if self._ticks_since_check >= 256:   # a loop that never reads env.agents / env.grid still learns of a device error
            self._sync()
            self._ticks_since_check = 0
        if self.rng == "reference":
            tape, produced = self._draw_update()
            self.worlds.update(self.worlds.make_tape([tape]))
            self._refresh(after="update")
            if produced and not self.static_families:  # Agent.mutate_brain (entities.py:210-213) draws torch.randn
                new = np.nonzero(self._host["a_gene"] == self.max_gene)[0]
                if len(new) and self.brains[int(self._host["a_brain"][new[0]])].method == "PERD3QN":
                    torch.randn((128, 153))  # PERD3QN.py:127-130; brains are shared by index here, only the stream advances
        else:
            self.worlds.update()
            if self.refill_below is not None:
                self.worlds.refill(self.refill_below, self.synthetic_agents)
            self._refresh(after="update")

    # -- the reference's random draws, made on the host from the global generators (rng="reference") -----------
    def _draw_add_food(self, n_food, n_poison, n_super, n_empty):
        """_add_food (environment.py:763-776) -> Grid.set_random (grid.py:69-83): randint(0, #empty) then random(), nothing
        when the grid is full (randint raises before drawing)."""
        C = self.width * self.height
        k = np.zeros(_lib.FOOD_TRIES, np.int32)
        u = np.full(_lib.FOOD_TRIES, 2.0)
        for tries, enabled, p in (((0, 1, 2), n_food <= C / 10, 0.2), ((3, 4, 5), n_poison <= C / 20, 0.2), ((6,), n_super == 0, 1.0)):
            if not enabled:
                continue
            for t in tries:
                if n_empty == 0:
                    continue
                k[t] = np.random.randint(0, n_empty)
                u[t] = np.random.random()
                n_empty -= u[t] < p
        return {"food_k": k, "food_u": u, "repro_u": [], "birth_k": [], "produce_u": 0.0, "produce_choice": 0}, n_empty

    def _draw_update(self):
        """_reproduce / _produce (environment.py:488-547) over the post-step agent list (host mirror)."""
        import random
        h = self._host
        n1 = len(h["a_age"])
        # (empty cells after _add_food: known from step()'s own draws; a caller who changed the world in between reads the grid)
        n_empty = self._n_empty if self._n_empty is not None else int((self.grid == _lib.EMPTY).sum())
        repro_u, birth_k = [], []
        produce_u, choice, produced = 0.0, 0, False

        def place():
            nonlocal n_empty
            if n_empty == 0:
                return False
            birth_k.append(np.random.randint(0, n_empty))
            np.random.random()  # the p=1 coin
            n_empty -= 1
            return True

        if n1 <= self.max_agents:
            for a in range(n1):
                if (h["a_flags"][a] & (_lib.F_DEAD | _lib.F_REPRODUCED)) or h["a_age"][a] <= 5:
                    continue
                r = random.random()
                repro_u.append(r)
                if r > 0.95:
                    place()
            produce_u = random.random()
            if produce_u > 0.95:
                if self.static_families:
                    genes = set([int(g) for g in h["a_gene"]])
                    not_alive = list(set(range(len(self.brains))).difference(genes))
                    choice = random.choice(not_alive) if not_alive else random.choice([x for x in range(len(self.brains))])
                else:
                    choice = random.choice(range(_lib.N_BEST))
                produced = place()
        return ({"food_k": np.zeros(_lib.FOOD_TRIES, np.int32), "food_u": np.zeros(_lib.FOOD_TRIES), "repro_u": repro_u,
                 "birth_k": birth_k, "produce_u": produce_u, "produce_choice": choice}, produced)

    def render_feed(self, world=0):
        """What the reference's painter reads (render.py:90-200) for one world: agents i/j/gene/health/killed/dead + food cells."""
        from ..Helpers.render import RenderFeed
        return RenderFeed.from_world(self.width, self.height, self.worlds.world(world))

    def render(self, fps=10, world=0):
        """environment.py:217-231.  No pygame window here: the frame is painted into `self.frame` (uint8 RGB,
        [height*grid_size, width*grid_size, 3]); returns True = keep going, like an open pygame window does."""
        self.frame = self.viz.frame(self.render_feed(world))
        return True

    def save_results(self, main_folder="experiments"):
        """environment.py:233-256: brains + parameters + results.json + settings.json in the reference's layout."""
        from ..Helpers.saver import SavedAgent, Saver
        settings = {"Update interval": self.tracker.update_interval, "Width": self.width, "Height": self.height,
                    "Max agents": self.max_agents, "Families": self.static_families,
                    # (extra key, not in the reference's file) this build never updates weights: say so next to them
                    "Weights": "as loaded / initialised -- reinlife_amd runs inference only, learn() is a no-op"}
        if self.static_families:
            agents = [SavedAgent(g, b) for g, b in enumerate(self.brains)]
        else:  # the brains the best agents descend from (inference-time copies share their weights)
            bb = self.worlds.s["best_brain"][0].cpu().numpy()
            agents = [SavedAgent(int(self.max_gene), self.brains[int(b)]) for b in bb]
        return Saver(main_folder, google_colab=self.google_colab).save(agents, self.static_families, self.tracker.results, settings)

    # -- fused loop ----------------------------------------------------------------------------------------------------
    def run(self, n_epi=0, n_ticks=1, max_chunk=4096):
        """`n_ticks` iterations of the trainer loop (trainer.py:85-99: get_action for every agent -> step -> update_env) starting at
        episode `n_epi`, with as few launches as the configuration allows: rng="philox" runs whole chunks in ONE launch each
        (rl_run_ex: worlds resident in LDS, the brains' per-episode epsilon as a schedule, the Tracker's statistics accumulated in the
        launch); chunks end where the reference's Tracker closes an interval (n_epi % update_interval == 0, tracker.py:107-121); episode
        0's statistics never reach an aggregate (tracker.py:279-282) and are left out inside the launch.  Same results as calling
        act() / step() / update_env() tick by tick (tests/test_hip_round3.py)."""
        if self.rng != "philox":
            for t in range(n_ticks):
                self.act(n_epi + t); self.step(); self.update_env(n_epi + t)
            return
        interval = self.tracker.update_interval
        thr = -1 if self.refill_below is None else self.refill_below
        while n_ticks > 0:
            k = min(n_ticks, max_chunk)
            if self.training and n_epi + k - 1 >= interval:   # up to and including the next episode that closes a Tracker interval
                k = min(k, (max(n_epi, 1) + interval - 1) // interval * interval - n_epi + 1)
            eps = self._epsilon_schedule(n_epi, k)
            self._bind_brains()   # (the brains' current epsilon = the last row; weights are uploaded only when they changed)
            if self.training and n_epi == 0:
                # Tracker.update_results(n_epi=0) zeroes the running sums (tracker.py:279-282): whatever an earlier loop on this
                # environment left in them goes, and episode 0 itself stays out of them inside the launch (trk_skip)
                self.worlds.reset_tracking()
            self.worlds.run(k, thr, self.synthetic_agents or 0, eps_schedule=None if (eps == eps[-1]).all() else eps,
                            trk_skip=1 if (self.training and n_epi == 0) else 0)
            # The interval the PREVIOUS chunk closed: its sums (and the error flag) were copied out behind that chunk; the host
            # reads, divides and prints them now, with this chunk queued -- the device does not wait for any of it.
            self.tracker.resolve()
            self._refresh(after="update")
            last = n_epi + k - 1
            if self.training and last > 0 and last % interval == 0:
                self.tracker.update_results(None, last, defer=True)   # the device half of the close: queued behind the chunk
            n_epi += k
            n_ticks -= k
        self.tracker.resolve()   # (a close behind the last chunk: a corrupted world stops the loop here at the latest)

    # -- host mirrors ----------------------------------------------------------------------------------------------------
    def agents_of(self, world):
        """env.agents of replica `world` (row-major list of AgentView): the same protocol as env.agents, which is world 0's.
        Views are valid until the next step() / update_env() / run()."""
        if not 0 <= world < self.n_worlds:
            raise IndexError("world %d of %d" % (world, self.n_worlds))
        return self._mat(world).views

    def _sync(self):
        torch.cuda.synchronize(self.worlds.device)
        self.worlds.check_error_flag()

    def _snapshot(self):
        """The device state on the host behind everything queued so far: ONE copy and ONE wait for a small handle (DeviceWorlds.snapshot);
        the error flag comes with it."""
        snap = self.worlds.snapshot()
        self.worlds.raise_on_error_flag(snap["err"])
        self._ticks_since_check = 0
        return snap

    def _build_mirror(self, world, snap):
        m = _WorldMirror(self, world, self._phase)
        n = m.n = int(snap["n_agents"][world])
        if world == 0 and "max_gene" in snap:
            self._max_gene = int(snap["max_gene"][0])
        m.host = {k: snap[k][world, :n] for k in _A_FIELDS}
        if self._phase == "step":
            m.src1, m.reward, m.done = snap["src1"][world], snap["reward"][world, :n], snap["done"][world, :n]
        if self._phase == "update" and self._acted:   # act() has run: what the policy chose for this tick (and what it computed)
            m.actions = snap["actions"][world].copy()
            m.q = snap["out_q"][world]
        else:                                          # Agent.action: the action last taken (-1 for newborns)
            m.actions = np.concatenate([m.host["a_action"], np.full(self.worlds.cap - n, -1, np.int8)])
        self._mirrors[world] = m   # (before the views: AgentView looks its mirror up)
        m.views = [AgentView(self, k, m) for k in range(n)]
        return m

    def _mat(self, world):
        """The host mirror of one world, built from the device state on first use."""
        m = self._mirrors.get(world)
        if m is None:
            w = self.worlds
            if hasattr(w, "snapshot") and w._arena.numel() <= w.SNAPSHOT_MAX_BYTES:
                m = self._build_mirror(world, self._snapshot())
            else:   # a big handle: this world's rows only
                self._sync()
                n = int(w.s["n_agents"][world].item())
                snap = {k: w.s[k][world:world + 1, :n].cpu().numpy() for k in w.s if k.startswith("a_")}
                snap["n_agents"] = np.array([n])
                for k in ("src1", "reward", "done", "actions", "out_q"):
                    snap[k] = getattr(w, k)[world:world + 1].cpu().numpy()
                snap = {k: _Shift(v, world) for k, v in snap.items()}   # (a plain dict: keys() / [] like a snapshot)
                m = self._build_mirror(world, snap)
        if m.actions is None:          # act() ran after this mirror was built: what the policy chose for this tick
            m.actions = self.worlds.actions[world].cpu().numpy().copy()
        return m

    def _refresh(self, after):
        """The device state changed: drop every host copy (they are rebuilt on first read -- no synchronisation here)."""
        self._phase = after
        self._acted = False
        self._mirrors = {}
        self._grid = self._max_gene = None
        self._stamp += 1
        if after != "step":
            self._n_empty = None


class _Shift:
    """A one-world slice that answers to the world's index in the full array (big handles: Environment._mat)."""

    def __init__(self, a, world):
        self.a, self.world = a, world

    def __getitem__(self, key):
        if isinstance(key, tuple):
            assert key[0] == self.world
            return self.a[(0,) + key[1:]]
        assert key == self.world
        return self.a[0]


def resolve_dist(dist=None):
    """(process-group module or None, rank, world size): `dist` as given (torch.distributed or a stand-in with the same calls), else
    torch.distributed when a default process group is initialised, else a single-process job."""
    if dist is None:
        import torch.distributed as td
        if td.is_available() and td.is_initialized():
            dist = td
    if dist is not None and dist.is_initialized():
        return dist, int(dist.get_rank()), int(dist.get_world_size())
    return None, 0, 1


def warn_inference_only():
    import warnings
    warnings.warn("reinlife_amd runs ReinLife's per-tick path (world tick + policy inference) only: training=True keeps the "
                  "reference's loop, epsilon schedules and Tracker, but brain.learn() is a no-op and the brains' weights are NOT "
                  "updated; results saved from such a run hold the initial (untrained) weights.  Train with the reference, or feed "
                  "DeviceWorlds.enable_capture() / capture_transitions() to your own learner.", stacklevel=3)


def host_reset(width, height, n_brains, rng=None):
    """Environment.reset's world construction with the reference's exact global-np.random draw order
    (environment.py:147-154 -> _add_agent :483-484 -> Grid.set_random grid.py:69-83; _init_food :741-761).  `rng`: a
    np.random.RandomState to draw from instead of the process-global generator (replicas other than world 0)."""
    rnd = rng if rng is not None else np.random
    C = width * height
    grid = np.zeros(C, np.uint8)
    agents = []

    def set_random(kind, p):
        empties = np.nonzero(grid == _lib.EMPTY)[0]  # row-major, like np.where on the 2-D grid
        if len(empties) == 0:
            return None
        k = rnd.randint(0, len(empties))
        if rnd.random_sample() < p:
            grid[empties[k]] = kind
            return int(empties[k])
        return None

    for g in range(n_brains):
        cell = set_random(_lib.AGENT, 1.0)
        if cell is not None:
            agents.append((cell, g))
    for kind, prob in ((_lib.FOOD, 0.1), (_lib.POISON, 0.05)):
        for _ in range(C):
            if rnd.random_sample() < prob:
                set_random(kind, 1)
    set_random(_lib.SUPER_FOOD, 1)
    order = sorted(range(len(agents)), key=lambda a: agents[a][0])
    n = len(agents)
    cells = np.array([agents[a][0] for a in order], dtype=np.int64).reshape(n)
    snap = {"cell_type": grid,
            "i": (cells // width).astype(np.uint8), "j": (cells % width).astype(np.uint8),
            "health": np.full(n, 200, np.int32), "age": np.zeros(n, np.int32), "max_age": np.full(n, 50, np.int32),
            "gene": np.array([agents[a][1] for a in order], np.int32).reshape(n),
            "brain": np.array([agents[a][1] for a in order], np.int32).reshape(n),
            "uid": np.array(order, np.int32).reshape(n), "flags": np.zeros(n, np.uint8), "action": np.full(n, -1, np.int8),
            "fitness": np.zeros(n, np.float64), "max_gene": n_brains, "next_uid": n,
            "best_uid": np.full(_lib.N_BEST, -1, np.int32), "best_fit": np.zeros(_lib.N_BEST), "best_brain": np.zeros(_lib.N_BEST, np.int32)}
    return snap
