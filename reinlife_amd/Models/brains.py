"""The four brains of the hot path with the reference's constructor keywords, attribute names and state-dict keys
(ReinLife/Models/DQN.py:18-63, D3QN.py:16-80, PERD3QN.py:10-79, PPO.py:10-52), so `load_model=` accepts the
reference's `pretrained/*.pt` files.  The forward pass and action selection run in libreinlife_hip.so (f32-grade block-scaled f16 MFMA):
batched over all agents through Environment.act(), or one state at a time through get_action().

Training (replay buffers, optimizers, learn()) is outside this build's scope (BASELINE.json north_star): learn() is
accepted and ignored so that trainer() loops written for the reference keep running; a warning is issued once.
"""
import random
import warnings

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .utils import BasicBrain


class _Qnet(nn.Module):  # DQN.py:118-124
    def __init__(self, input_dim):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, 128)
        self.fc2 = nn.Linear(128, 64)
        self.fc3 = nn.Linear(64, 8)


class _Dueling(nn.Module):  # D3QN.py:149-159 / PERD3QN.py:186-196
    def __init__(self, observation_dim, action_dim):
        super().__init__()
        self.fc = nn.Linear(observation_dim, 128)
        self.adv_fc1 = nn.Linear(128, 128)
        self.adv_fc2 = nn.Linear(128, action_dim)
        self.value_fc1 = nn.Linear(128, 128)
        self.value_fc2 = nn.Linear(128, 1)


class _PPONet(nn.Module):  # PPO.py:95-98
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, 256)
        self.fc2 = nn.Linear(256, 256)
        self.fc_pi = nn.Linear(256, output_dim)
        self.fc_v = nn.Linear(256, 1)


_warned = [False]


class _HipBrain(BasicBrain):
    kind = None

    def _net(self):
        raise NotImplementedError

    def state_dict_flat(self):
        """float32 parameters in state-dict (registration) order: the layout rl_policy_pack_weights expects."""
        return np.concatenate([p.detach().cpu().numpy().astype(np.float32).reshape(-1) for p in self._net().state_dict().values()])

    def packed_weights(self, device="cuda:0"):
        """Packed MFMA layout of the current parameters on `device`, cached.  The cache key carries every parameter's storage
        address and in-place version counter, so load_state_dict(), optimizer steps, target/eval syncs or a replaced module
        repack automatically at the next use."""
        net = self._net()
        # The key must notice every way the weights can change -- in-place updates (optimizer steps, load_state_dict: _version), a Parameter
        # object swapped for a new one (net.fc1.weight = nn.Parameter(...), weight_norm / parametrize, .to() under
        # overwrite_module_params_on_conversion: another data_ptr), a replaced sub-module or network (ids) -- without walking
        # nn.Module.parameters() / buffers() through their generators on every call (50 us per brain: more than a short launch's whole host
        # side).  So the SLOTS the tensors live in ((dict, name) per parameter and buffer) are listed once per set of module objects, and
        # every call reads the tensors that are in those slots NOW.
        mods = (id(net),) + tuple(map(id, net._modules.values()))   # (the reference's networks are flat: Linear children only)
        if getattr(self, "_slots_of", None) != mods:
            self._slots = [(m._parameters, k) for m in net.modules() for k in m._parameters] + [(m._buffers, k) for m in net.modules() for k in m._buffers]
            self._slots_of = mods
        key = (str(device),) + mods + tuple((t.data_ptr(), t._version) for t in (d[k] for d, k in self._slots) if t is not None)
        if getattr(self, "_packed_key", None) != key:
            from ..worlds import pack_brain_weights
            self._packed = pack_brain_weights(self.kind, self.state_dict_flat(), device)
            self._packed_key = key
        return self._packed

    def invalidate(self):
        self._packed_key = self._slots_of = None

    def forward_batch(self, states, device="cuda:0"):
        """[n,153] observations -> [n,8] Q values / probabilities on the GPU."""
        from ..worlds import policy_forward
        obs = torch.as_tensor(np.ascontiguousarray(states, dtype=np.float32), device=device).reshape(-1, _lib.OBS_DIM)
        return policy_forward(self.kind, self.packed_weights(device), obs)

    def learn(self, **kwargs):
        if not _warned[0]:
            warnings.warn("reinlife_amd brains run inference only; learn() is ignored (training is outside this build's scope)")
            _warned[0] = True


class DQNAgent(_HipBrain):
    kind = _lib.DQN

    def __init__(self, input_dim=153, output_dim=8, max_epi=0, learning_rate=0.0005, train_freq=20, load_model=False,
                 training=True):
        super().__init__(input_dim, output_dim, "DQN")
        self.agent = _Qnet(input_dim)
        self.max_epi = max_epi
        self.epsilon = 0.20
        self.train_freq = train_freq
        self.training = training
        if not self.training:
            self.epsilon = 0
        if load_model:
            self.agent.load_state_dict(torch.load(load_model))
            self.agent.eval()

    def _net(self):
        return self.agent

    def update_epsilon(self, n_epi):  # DQN.py:67-69
        if self.training and n_epi % 30 == 0:
            self.epsilon = max(0.01, 0.20 - 0.20 * (n_epi / self.max_epi))

    def epsilon_schedule(self, n_epi, k):
        """update_epsilon for episodes n_epi .. n_epi + k - 1 in one go: the rate in force in each of them (float64, the same
        arithmetic), the brain left as after the last."""
        out = np.empty(k, np.float64)
        t = 0
        while t < k:
            self.update_epsilon(n_epi + t)           # (acts on multiples of 30 only)
            nxt = min(k, t + 30 - (n_epi + t) % 30)  # the rate holds until the next multiple
            out[t:nxt] = self.epsilon
            t = nxt
        return out

    def get_action(self, state, n_epi, out=None):
        """`out`: this state's Q values when the caller already ran the forward pass in a batch (Environment.act)."""
        self.update_epsilon(n_epi)
        if out is None:
            out = self.forward_batch(np.asarray(state)[None])[0]
        coin = random.random()  # forward first, then the coin (DQN.py:134-139)
        if coin < self.epsilon:
            return random.randint(0, 7)
        return int(out.argmax().item())


class _DuelingAgent(_HipBrain):
    def __init__(self, method, input_dim, output_dim, exploration, soft_update_freq, train_freq, batch_size, gamma,
                 load_model, training):
        super().__init__(input_dim, output_dim, method)
        self.target_net = _Dueling(input_dim, output_dim)
        self.eval_net = _Dueling(input_dim, output_dim)
        self.eval_net.load_state_dict(self.target_net.state_dict())
        self.exploration, self.soft_update_freq, self.train_freq = exploration, soft_update_freq, train_freq
        self.batch_size, self.gamma = batch_size, gamma
        self.n_epi = 0
        self.epsilon, self.epsilon_min, self.decay = 0.9, 0.05, 0.99
        self.training = training
        if not self.training:
            self.epsilon = 0
        if load_model:
            self.eval_net.load_state_dict(torch.load(load_model))
            self.eval_net.eval()

    def _net(self):
        return self.eval_net

    def update_epsilon(self, n_epi):  # D3QN.py:84-89 / PERD3QN.py:82-86
        if self.training and n_epi > self.n_epi:
            if self.epsilon > self.epsilon_min:
                self.epsilon = self.epsilon * self.decay
            self.n_epi = n_epi

    def epsilon_schedule(self, n_epi, k):
        """update_epsilon for episodes n_epi .. n_epi + k - 1 in one go (the same float64 products in the same order); once the
        rate has reached its floor -- after ~290 episodes -- the rest of the range is one fill."""
        out = np.empty(k, np.float64)
        t = 0
        if self.training:   # (update_epsilon's rule on locals: a short schedule is host time in front of a short launch)
            e, last, floor, decay = self.epsilon, self.n_epi, self.epsilon_min, self.decay
            while t < k and e > floor:
                ne = n_epi + t
                if ne > last:
                    e = e * decay   # (e > floor holds here)
                    last = ne
                out[t] = e
                t += 1
            self.epsilon, self.n_epi = e, last
        if t < k:
            out[t:] = self.epsilon
            if self.training:
                self.n_epi = max(self.n_epi, n_epi + k - 1)
        return out

    def get_action(self, state, n_epi, out=None):
        self.update_epsilon(n_epi)
        if random.random() > self.epsilon:  # D3QN.py:168-172 / PERD3QN.py:205-209
            q = self.forward_batch(np.asarray(state)[None])[0] if out is None else out
            return int(q.argmax().item())
        return random.choice(list(range(self.output_dim)))

    def apply_gaussian_noise(self):  # net effect on the weights: none (PERD3QN.py:127-130)
        pass


class D3QNAgent(_DuelingAgent):
    kind = _lib.D3QN

    def __init__(self, input_dim=153, output_dim=8, exploration=1000, soft_update_freq=200, train_freq=20,
                 learning_rate=1e-3, gamma=0.99, batch_size=64, capacity=10000, load_model=False, training=True):
        super().__init__("D3QN", input_dim, output_dim, exploration, soft_update_freq, train_freq, batch_size, gamma,
                         load_model, training)


class PERD3QNAgent(_DuelingAgent):
    kind = _lib.PERD3QN

    def __init__(self, input_dim=153, output_dim=8, exploration=1000, soft_update_freq=200, train_freq=20,
                 learning_rate=1e-3, batch_size=64, capacity=10000, gamma=0.99, load_model=False, training=True):
        super().__init__("PERD3QN", input_dim, output_dim, exploration, soft_update_freq, train_freq, batch_size, gamma,
                         load_model, training)


class PPOAgent(_HipBrain):
    kind = _lib.PPO

    def __init__(self, input_dim=153, output_dim=8, learning_rate=0.0005, gamma=0.98, lmbda=0.95, eps_clip=0.1, k_epoch=3,
                 train_freq=20, load_model=False):
        super().__init__(input_dim, output_dim, "PPO")
        self.model = _PPONet(input_dim, output_dim)
        self.load_model = load_model
        self.train_freq = train_freq
        self.epsilon = 0.0
        if self.load_model:
            self.model.load_state_dict(torch.load(load_model))
            self.model.eval()

    def _net(self):
        return self.model

    def update_epsilon(self, n_epi):
        pass

    def epsilon_schedule(self, n_epi, k):
        return np.zeros(k, np.float64)

    def get_action(self, s, out=None):
        prob = (self.forward_batch(np.asarray(s)[None])[0] if out is None else out).cpu()
        a = int(torch.distributions.Categorical(prob).sample().item())  # PPO.py:164-169
        return a if self.load_model else (a, prob)
