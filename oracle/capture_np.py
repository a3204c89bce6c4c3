"""TEST INFRASTRUCTURE ONLY -- numpy restatement of what the reference's trainer loop hands to brain.learn each tick
(ReinLife/Helpers/trainer.py:95-96 -> World/entities.py:194-208): for every agent of the post-step env.agents, in list
order, if age > 1:  (age, dead, action, state, reward, state_prime, done)  [+ prob for PPO].  `state` is the observation
the policy read before the step, `state_prime` the post-step observation.  Pinned at fixture-generation time against the
kwargs the real Agent.learn passed (oracle/ref_harness.py record_tick asserts them)."""
import numpy as np


def transitions(prev_state, actions, src, age, flags, reward, done, state_prime, brain):
    """All arguments are per-world 1-D/2-D arrays: prev_state [n0,153] (pre-step list order), actions [n0], and the
    post-step list arrays src/age/flags/reward/done [n1], state_prime [n1,153], brain [n1].
    -> dict of arrays in Agent.learn call order."""
    k = np.nonzero(np.asarray(age) > 1)[0]
    s = np.asarray(src)[k]
    return {"k": k.astype(np.int16), "brain": np.asarray(brain)[k], "age": np.asarray(age)[k],
            "dead": (np.asarray(flags)[k] & 1).astype(np.uint8), "action": np.asarray(actions)[s],
            "state": np.asarray(prev_state)[s], "reward": np.asarray(reward)[k], "state_prime": np.asarray(state_prime)[k],
            "done": np.asarray(done)[k]}
