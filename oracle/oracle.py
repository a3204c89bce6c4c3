"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/rl_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
(reinlife_amd/) never does.  State lives in numpy struct-of-arrays with exactly the layout of `rlo_state`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "librl_oracle.so")

OBS_DIM = 153
N_BEST = 10
FOOD_TRIES = 7
EMPTY, FOOD, POISON, AGENT, KIN, SUPER = 0, 1, 2, 3, 4, 5
F_DEAD, F_REPRODUCED, F_KILLED, F_ATE_SUPER, F_INTER, F_INTRA = 1, 2, 4, 8, 16, 32
DQN, D3QN, PERD3QN, PPO = 0, 1, 2, 3
KIND_BY_NAME = {"DQN": DQN, "D3QN": D3QN, "PERD3QN": PERD3QN, "PPO": PPO}


def build(force=False):
    src = os.path.join(HERE, "rl_oracle.c")
    hdr = os.path.join(HERE, "rl_oracle.h")
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return LIB_PATH
    subprocess.check_call(["make", "-C", HERE, "-s", "-B"])
    return LIB_PATH


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_agents", C.c_int32), ("n_brains", C.c_int32),
                ("slot_cap", C.c_int32), ("n_worlds", C.c_int32), ("static_families", C.c_int32),
                ("limit_reproduction", C.c_int32), ("incentivize_killing", C.c_int32), ("world_base", C.c_int32),
                ("seed", C.c_uint64)]


_STATE_FIELDS = [  # (name, dtype, per-world shape suffix as a function of (C, cap))
    ("cell_type", np.uint8, "C"), ("n_agents", np.int32, ""), ("a_i", np.uint8, "cap"), ("a_j", np.uint8, "cap"),
    ("a_health", np.int32, "cap"), ("a_age", np.int32, "cap"), ("a_max_age", np.int32, "cap"),
    ("a_gene", np.int32, "cap"), ("a_brain", np.int32, "cap"), ("a_uid", np.int32, "cap"),
    ("a_flags", np.uint8, "cap"), ("a_action", np.int8, "cap"), ("a_fitness", np.float64, "cap"),
    ("max_gene", np.int32, ""), ("next_uid", np.int32, ""), ("tick", np.int32, ""), ("epoch", np.int32, ""),
    ("best_uid", np.int32, "best"), ("best_fit", np.float64, "best"), ("best_brain", np.int32, "best")]


class State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _, _ in _STATE_FIELDS]


class Tape(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("food_k", "food_u", "repro_u", "birth_k", "produce_u", "produce_choice")]


class StepOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("n_acted", "reward", "done", "src", "obs", "l0_health", "l0_flags",
                                          "l0_reward", "l0_i", "l0_j", "trk_tick", "trk_sum", "trk_cnt", "trk_pop")]


class UpdateOut(C.Structure):
    _fields_ = [("src", C.c_void_p), ("obs", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.rlo_policy_n_params.restype = C.c_int64
    return _lib


def slot_cap_for(max_agents, n_cells):
    cap = ((2 * max_agents + 2 + 63) // 64) * 64
    return cap


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleWorlds:
    """R independent worlds in the oracle's SoA layout."""

    def __init__(self, n_worlds=1, width=30, height=30, max_agents=100, n_brains=2, static_families=True,
                 limit_reproduction=False, incentivize_killing=True, seed=0, slot_cap=None, world_base=0):
        self.R, self.W, self.H, self.C = n_worlds, width, height, width * height
        self.cap = slot_cap or slot_cap_for(max_agents, self.C)
        self.cfg = Config(width, height, max_agents, n_brains, self.cap, n_worlds, int(static_families),
                          int(limit_reproduction), int(incentivize_killing), world_base, seed)
        dims = {"C": (self.C,), "cap": (self.cap,), "best": (N_BEST,), "": ()}
        self.s = {}
        for name, dt, suf in _STATE_FIELDS:
            self.s[name] = np.zeros((self.R,) + dims[suf], dtype=dt)
        self.s["best_uid"][:] = -1
        self.s["max_gene"][:] = n_brains
        self._state = State(*[_ptr(self.s[n]) for n, _, _ in _STATE_FIELDS])
        # step/update outputs
        self.n_acted = np.zeros(self.R, np.int32)
        self.reward = np.zeros((self.R, self.cap), np.float32)
        self.done = np.zeros((self.R, self.cap), np.uint8)
        self.src1 = np.full((self.R, self.cap), -1, np.int16)
        self.obs1 = np.zeros((self.R, self.cap, OBS_DIM), np.float32)
        self.l0_health = np.zeros((self.R, self.cap), np.int32)
        self.l0_flags = np.zeros((self.R, self.cap), np.uint8)
        self.l0_reward = np.zeros((self.R, self.cap), np.float64)
        self.l0_i = np.zeros((self.R, self.cap), np.uint8)
        self.l0_j = np.zeros((self.R, self.cap), np.uint8)
        self.src2 = np.full((self.R, self.cap), -1, np.int16)
        self.obs2 = np.zeros((self.R, self.cap, OBS_DIM), np.float32)
        self.G = n_brains if static_families else 1
        self.trk_tick = np.zeros((self.R, self.G, 7), np.float64)
        self.trk_sum = np.zeros((self.R, self.G, 7), np.float64)
        self.trk_cnt = np.zeros((self.R, self.G, 7), np.int32)
        self.trk_pop = np.zeros((self.R, 3), np.float64)
        self._step_out = StepOut(*[_ptr(a) for a in (self.n_acted, self.reward, self.done, self.src1, self.obs1,
                                                     self.l0_health, self.l0_flags, self.l0_reward, self.l0_i,
                                                     self.l0_j, self.trk_tick, self.trk_sum, self.trk_cnt, self.trk_pop)])
        self._upd_out = UpdateOut(_ptr(self.src2), _ptr(self.obs2))

    # -- state I/O ----------------------------------------------------------------------------------------------
    def load_world(self, w, snap):
        """snap: dict as produced by ref_harness.snapshot_world (row-major agent list)."""
        n = len(snap["i"])
        assert n <= self.cap
        self.s["cell_type"][w] = snap["cell_type"]
        self.s["n_agents"][w] = n
        for key in ("i", "j", "health", "age", "max_age", "gene", "brain", "uid", "flags", "action", "fitness"):
            self.s["a_" + key][w, :n] = snap[key]
        self.s["max_gene"][w] = snap.get("max_gene", self.cfg.n_brains)
        self.s["next_uid"][w] = snap.get("next_uid", (int(snap["uid"].max()) + 1) if n else 0)
        for key in ("best_uid", "best_fit", "best_brain"):
            if key in snap:
                self.s[key][w] = snap[key]

    def world(self, w):
        n = int(self.s["n_agents"][w])
        d = {k[2:]: self.s[k][w, :n].copy() for k in self.s if k.startswith("a_")}
        d["cell_type"] = self.s["cell_type"][w].copy()
        for key in ("max_gene", "next_uid", "tick", "epoch"):
            d[key] = int(self.s[key][w])
        for key in ("best_uid", "best_fit", "best_brain"):
            d[key] = self.s[key][w].copy()
        return d

    def make_tape(self, tapes):
        """tapes: list (one per world) of dicts from ref_harness.extract_tape -> keeps arrays alive on self."""
        R, cap = self.R, self.cap
        self._t = {"food_k": np.zeros((R, FOOD_TRIES), np.int32), "food_u": np.zeros((R, FOOD_TRIES), np.float64),
                   "repro_u": np.zeros((R, cap), np.float64), "birth_k": np.zeros((R, cap + 1), np.int32),
                   "produce_u": np.zeros(R, np.float64), "produce_choice": np.zeros(R, np.int32)}
        for w, t in enumerate(tapes):
            self._t["food_k"][w] = t["food_k"]
            self._t["food_u"][w] = t["food_u"]
            m = min(cap, len(t["repro_u"]))
            self._t["repro_u"][w, :m] = t["repro_u"][:m]
            m = min(cap + 1, len(t["birth_k"]))
            self._t["birth_k"][w, :m] = t["birth_k"][:m]
            self._t["produce_u"][w] = t["produce_u"]
            self._t["produce_choice"][w] = t["produce_choice"]
        return Tape(*[_ptr(self._t[n]) for n in ("food_k", "food_u", "repro_u", "birth_k", "produce_u",
                                                 "produce_choice")])

    # -- the path -----------------------------------------------------------------------------------------------
    def step(self, actions, tape=None, w0=0, w1=None):
        actions = np.ascontiguousarray(actions, dtype=np.int8)
        assert actions.shape == (self.R, self.cap)
        rc = lib().rlo_step(C.byref(self.cfg), C.byref(self._state), _ptr(actions),
                            C.byref(tape) if tape is not None else None, C.byref(self._step_out), w0,
                            self.R if w1 is None else w1)
        if rc:
            raise RuntimeError("rlo_step failed: %d" % rc)

    def update(self, tape=None, w0=0, w1=None):
        rc = lib().rlo_update(C.byref(self.cfg), C.byref(self._state), C.byref(tape) if tape is not None else None,
                              C.byref(self._upd_out), w0, self.R if w1 is None else w1)
        if rc:
            raise RuntimeError("rlo_update failed: %d" % rc)

    def observe(self, w0=0, w1=None):
        rc = lib().rlo_observe(C.byref(self.cfg), C.byref(self._state), _ptr(self.obs2), w0,
                               self.R if w1 is None else w1)
        if rc:
            raise RuntimeError("rlo_observe failed: %d" % rc)
        return self.obs2

    def reset_synthetic(self, n_agents, w0=0, w1=None):
        rc = lib().rlo_reset_synthetic(C.byref(self.cfg), C.byref(self._state), n_agents, _ptr(self.obs2), w0,
                                       self.R if w1 is None else w1)
        if rc:
            raise RuntimeError("rlo_reset_synthetic failed: %d" % rc)

    def reset_families(self, w0=0, w1=None):
        rc = lib().rlo_reset_families(C.byref(self.cfg), C.byref(self._state), _ptr(self.obs2), w0, self.R if w1 is None else w1)
        if rc:
            raise RuntimeError("rlo_reset_families failed: %d" % rc)

    def refill(self, threshold, n_agents, w0=0, w1=None):
        rc = lib().rlo_refill(C.byref(self.cfg), C.byref(self._state), threshold, n_agents, _ptr(self.obs2), w0,
                              self.R if w1 is None else w1)
        if rc < 0:
            raise RuntimeError("rlo_refill failed: %d" % rc)
        return rc


def policy_forward(kind, weights, obs):
    """weights: flat float32 vector in state-dict order; obs [n,153] float32 -> [n,8] float32."""
    weights = np.ascontiguousarray(weights, np.float32)
    obs = np.ascontiguousarray(obs, np.float32)
    assert weights.size == lib().rlo_policy_n_params(kind), (weights.size, lib().rlo_policy_n_params(kind))
    out = np.zeros((obs.shape[0], 8), np.float32)
    rc = lib().rlo_policy_forward(kind, _ptr(weights), _ptr(obs), obs.shape[0], _ptr(out))
    if rc:
        raise RuntimeError("rlo_policy_forward failed: %d" % rc)
    return out


def select_actions(cfg, kind, out, world_of_row, index_in_world, tick_of_world, epoch_of_world, eps):
    out = np.ascontiguousarray(out, np.float32)
    n = out.shape[0]
    world_of_row = np.ascontiguousarray(world_of_row, np.int32)
    index_in_world = np.ascontiguousarray(index_in_world, np.int32)
    tick_of_world = np.ascontiguousarray(tick_of_world, np.int32)
    epoch_of_world = np.ascontiguousarray(epoch_of_world, np.int32)
    actions = np.zeros(n, np.int8)
    rc = lib().rlo_select_actions(C.byref(cfg), kind, _ptr(out), n, _ptr(world_of_row), _ptr(index_in_world),
                                  _ptr(tick_of_world), _ptr(epoch_of_world), C.c_float(eps), _ptr(actions))
    if rc:
        raise RuntimeError("rlo_select_actions failed: %d" % rc)
    return actions


def philox(seed, epoch, world, tick, site, index):
    out = (C.c_uint32 * 4)()
    lib().rlo_philox(C.c_uint64(seed), C.c_uint32(epoch), C.c_uint32(world), C.c_uint32(tick), C.c_uint32(site),
                     C.c_uint32(index), out)
    return [int(x) for x in out]
