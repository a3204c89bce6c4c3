"""Adapter giving reinlife_amd.worlds.DeviceWorlds (the C-ABI / HIP path) the OracleWorlds test surface."""
import numpy as np

from reinlife_amd.worlds import DeviceWorlds


class HipBackend:
    def __init__(self, n_worlds, fused=False, **cfg):
        self.dw = DeviceWorlds(n_worlds=n_worlds, seed=cfg.pop("seed", 0), world_base=cfg.pop("world_base", 0), **cfg)
        self.cap = self.dw.cap
        self.dw.enable_tracking(True)
        self.fused = fused

    def load_world(self, w, snap):
        self.dw.load_world(w, snap)

    def world(self, w):
        return self.dw.world(w)

    def make_tape(self, tapes):
        return self.dw.make_tape(tapes)

    def observe(self):
        return self.dw.observe().cpu().numpy()

    def step(self, actions, tape=None):
        self.dw.step(actions, tape)
        self.dw.check_error_flag()

    def update(self, tape=None):
        self.dw.update(tape)
        self.dw.check_error_flag()

    def tick(self, actions, tape=None):
        self.dw.tick(actions, tape)
        self.dw.check_error_flag()

    n_acted = property(lambda s: s.dw.n_acted.cpu().numpy())
    reward = property(lambda s: s.dw.reward.cpu().numpy())
    done = property(lambda s: s.dw.done.cpu().numpy())
    src1 = property(lambda s: s.dw.src1.cpu().numpy())
    src2 = property(lambda s: s.dw.src2.cpu().numpy())
    obs1 = property(lambda s: s.dw.obs_state_prime().cpu().numpy())
    obs2 = property(lambda s: s.dw.obs_state().cpu().numpy())
    trk_tick = property(lambda s: s.dw.trk_tick.cpu().numpy())
    trk_pop = property(lambda s: s.dw.trk_pop.cpu().numpy())
    trk_sum = property(lambda s: s.dw.trk_sum.cpu().numpy())
    trk_cnt = property(lambda s: s.dw.trk_cnt.cpu().numpy())
