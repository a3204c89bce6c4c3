"""GPU: Environment(rng="reference") follows the reference's trajectory from the SEEDS alone.

The golden traces (tests/golden/trace_*.npz, recorded from the real reference by oracle/gen_golden.py) were produced after
`random.seed(s); np.random.seed(s); torch.manual_seed(s)`.  Here the same three seeds are set, the product Environment is
reset / stepped / updated with the recorded actions, and every tick must equal the trace: integer state bit-exact,
rewards and observations float32-identical.  No recorded tape is used: the product makes the reference's own draws from
the global generators (reinlife_amd/World/environment.py:_draw_add_food/_draw_update).
"""
import os
import random

import numpy as np
import pytest
import torch

import golden_io as gio

pytestmark = pytest.mark.gpu

# name -> (seed, agents the recorder topped the world up to after reset(), oracle/gen_golden.py:285-292)
CASES = {"natural_static": (11, 0), "natural_nonstatic": (12, 0), "dense100": (13, 100), "dense100_nonstatic": (14, 100),
         "dense200_attack": (15, 200), "movers250": (16, 250), "small7x5": (17, 12), "rect30x20_limit": (18, 60)}


class FixedBrain:
    """Stands in for the recorder's action-less brains: the test writes agent.action itself."""
    method = "DQN"
    kind = 0
    epsilon = 0.0

    def update_epsilon(self, n_epi):
        pass


@pytest.mark.parametrize("name", sorted(CASES))
def test_same_seeds_same_trajectory(name):
    from reinlife_amd import Environment
    from reinlife_amd.World.environment import host_reset
    seed, fill = CASES[name]
    tr = np.load(os.path.join(gio.GOLDEN_DIR, "trace_%s.npz" % name))
    cfg, ticks = gio.trace_cfg(tr)
    static = cfg["static_families"]
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    env = Environment(width=cfg["width"], height=cfg["height"], brains=[FixedBrain() for _ in range(cfg["n_brains"])],
                      max_agents=cfg["max_agents"], static_families=static, limit_reproduction=cfg["limit_reproduction"],
                      incentivize_killing=cfg["incentivize_killing"], training=False, print_results=False)
    assert env.rng == "reference"
    env.reset()
    if fill:
        # the recorder then added agents through Grid.set_random (global np.random: randint(0, #empty), random());
        # make the same draws, then take the recorded world (the added genes came from a private generator)
        n_empty = int((env.grid == 0).sum())
        for k in range(fill - cfg["n_brains"]):
            np.random.randint(0, n_empty - k)
            np.random.random()
        env.worlds.load_world(0, gio.initial_snapshot(tr))
        env.worlds.observe()
        env._refresh(after="update")
    n_init = int(tr["init_n"])
    assert len(env.agents) == n_init
    gio._cmp_obs(name + " initial obs", np.stack([a.state for a in env.agents]), tr["init_obs"], True)
    for t in range(ticks):
        n0 = int(tr["n0"][t])
        assert len(env.agents) == n0
        for a, act in zip(env.agents, tr["actions"][t][:n0]):
            a.action = int(act)
        env.step()
        n1 = int(tr["step_n"][t])
        tag = "%s tick %d step" % (name, t)
        gio.check_world(tag, env.worlds.world(0), tr, "step", t, True)
        gio._eq(tag, "done", [a.done for a in env.agents], tr["step_done"][t][:n1].astype(bool))
        gio._cmp_obs(tag + " reward", [a.reward for a in env.agents], tr["step_reward"][t][:n1], True)
        if n1:
            gio._cmp_obs(tag + " obs", np.stack([a.state_prime for a in env.agents]), tr["step_obs"][t][:n1], True)
        env.update_env(t)
        n2 = int(tr["upd_n"][t])
        tag = "%s tick %d update" % (name, t)
        gio.check_world(tag, env.worlds.world(0), tr, "upd", t, static)
        if n2:
            gio._cmp_obs(tag + " obs", np.stack([a.state for a in env.agents]), tr["upd_obs"][t][:n2], True)
