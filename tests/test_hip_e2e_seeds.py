"""GPU: the whole inference loop (brains' action selection + step + update_env) follows the reference from SEEDS alone.

tests/golden/e2e_*.npz (oracle/gen_golden_e2e.py) hold runs of the real reference with its real DQN / D3QN / PERD3QN / PPO
brains carrying repo-generated weights: per tick the actions its brains chose and the world after step() and update_env().
Here the product's brains get the same weights, `random` / `np.random` / torch get the same seeds, and the same loop
(agent.get_action(n_epi) per agent, env.step(), env.update_env(n_epi)) must reproduce every action and every world:
the epsilon-greedy coins and PPO's categorical samples come from the same generator calls as the reference's
(Models/DQN.py:134-139, D3QN.py:168-172, PPO.py:164-169), Q values / probabilities from the MFMA policy kernel.
"""
import os
import random

import numpy as np
import pytest
import torch

import golden_io as gio

pytestmark = pytest.mark.gpu

KIND_NAMES = {0: "DQN", 1: "D3QN", 2: "PERD3QN", 3: "PPO"}


def _load_flat(net, flat):
    sd, off = {}, 0
    for name, t in net.state_dict().items():
        n = t.numel()
        sd[name] = torch.from_numpy(flat[off:off + n].reshape(tuple(t.shape)).copy())
        off += n
    assert off == len(flat)
    net.load_state_dict(sd)


def make_brains(tr, ticks):
    from reinlife_amd import Models
    brains = []
    for idx, (kind, training) in enumerate(zip(tr["kinds"], tr["training"])):
        name = KIND_NAMES[int(kind)]
        flat = tr["weights_%d" % idx]
        if name == "DQN":
            b = Models.DQN(max_epi=ticks, training=bool(training))
            _load_flat(b.agent, flat)
        elif name == "PPO":
            b = Models.PPO()
            _load_flat(b.model, flat)
        else:
            b = getattr(Models, name)(training=bool(training))
            _load_flat(b.eval_net, flat)
            _load_flat(b.target_net, flat)
        b.invalidate()
        brains.append(b)
    return brains


@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("name", ["e2e_static_mixed", "e2e_static_explore", "e2e_nonstatic_greedy"])
def test_loop_from_seeds_matches_reference(name, batched):
    """batched=False: literal per-agent agent.get_action(n_epi); True: env.act(n_epi) (one forward launch per brain, the
    draws still made agent by agent) -- both must follow the reference."""
    from reinlife_amd import Environment
    tr = np.load(os.path.join(gio.GOLDEN_DIR, name + ".npz"))
    cfg, ticks = gio.trace_cfg(tr)
    static = cfg["static_families"]
    brains = make_brains(tr, ticks)
    seed = int(tr["seed"])
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)  # after construction, like the recorder
    env = Environment(width=cfg["width"], height=cfg["height"], brains=brains, max_agents=cfg["max_agents"],
                      static_families=static, training=False, print_results=False)
    env.reset()
    for t in range(ticks):
        n0 = int(tr["n0"][t])
        assert len(env.agents) == n0
        if batched:
            env.act(t)
        else:
            for agent in env.agents:
                agent.get_action(t)
        gio._eq("%s tick %d" % (name, t), "actions", [a.action for a in env.agents], tr["actions"][t][:n0])
        env.step()
        n1 = int(tr["step_n"][t])
        tag = "%s tick %d step" % (name, t)
        gio.check_world(tag, env.worlds.world(0), tr, "step", t, True)
        gio._cmp_obs(tag + " reward", [a.reward for a in env.agents], tr["step_reward"][t][:n1], True)
        env.update_env(t)
        gio.check_world("%s tick %d update" % (name, t), env.worlds.world(0), tr, "upd", t, static)
