"""Same exports as ReinLife/Models/__init__.py:1-5 (PERDQN is out of scope)."""
from .brains import D3QNAgent as D3QN
from .brains import DQNAgent as DQN
from .brains import PERD3QNAgent as PERD3QN
from .brains import PPOAgent as PPO
from .utils import BasicBrain

__all__ = ["D3QN", "DQN", "PERD3QN", "PPO", "BasicBrain"]
