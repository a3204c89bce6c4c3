"""ctypes binding of libreinlife_hip.so (the C ABI declared in include/reinlife_hip.h).

There is no CPU fallback: if the HIP library is missing and cannot be built, importing it raises."""
import ctypes as C
import os

from . import build as _build

OBS_DIM = 153
N_ACTIONS = 8
N_BEST = 10
FOOD_TRIES = 7
EMPTY, FOOD, POISON, AGENT, KIN, SUPER_FOOD = 0, 1, 2, 3, 4, 5
F_DEAD, F_REPRODUCED, F_KILLED, F_ATE_SUPER, F_INTER_KILLED, F_INTRA_KILLED = 1, 2, 4, 8, 16, 32
DQN, D3QN, PERD3QN, PPO = 0, 1, 2, 3
KIND_BY_METHOD = {"DQN": DQN, "D3QN": D3QN, "PERD3QN": PERD3QN, "PPO": PPO}


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_agents", C.c_int32), ("n_brains", C.c_int32),
                ("slot_cap", C.c_int32), ("n_worlds", C.c_int32), ("static_families", C.c_int32),
                ("limit_reproduction", C.c_int32), ("incentivize_killing", C.c_int32), ("world_base", C.c_int32),
                ("seed", C.c_uint64)]


STATE_FIELDS = ("cell_type", "n_agents", "a_i", "a_j", "a_health", "a_age", "a_max_age", "a_gene", "a_brain", "a_uid",
                "a_flags", "a_action", "a_fitness", "max_gene", "next_uid", "tick", "epoch", "best_uid", "best_fit",
                "best_brain")
TAPE_FIELDS = ("food_k", "food_u", "repro_u", "birth_k", "produce_u", "produce_choice")
STEP_OUT_FIELDS = ("n_acted", "reward", "done", "src", "obs", "acted_total", "trk_tick", "trk_sum", "trk_cnt", "trk_pop", "n_post",
                   "age", "brain")
TRK_VARS = 7
EPS_INLINE_MAX = 256   # rl_run_opts.eps_schedule_on_host: floats that fit into the kernel arguments
UPDATE_OUT_FIELDS = ("src", "obs")


class State(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in STATE_FIELDS]


class Tape(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in TAPE_FIELDS]


class StepOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in STEP_OUT_FIELDS]


class UpdateOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in UPDATE_OUT_FIELDS]


class Replay(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("state", "state_prime", "action", "reward", "done", "prob", "age", "count")] + [("capacity", C.c_int64)]


class RunOpts(C.Structure):  # rl_run_opts
    _fields_ = [("threshold", C.c_int32), ("n_agents", C.c_int32), ("refill_count", C.c_void_p), ("eps_schedule", C.c_void_p),
                ("trk_skip_ticks", C.c_int32), ("eps_schedule_on_host", C.c_int32), ("replays", C.c_void_p), ("policy_out", C.c_void_p)]


class Brain(C.Structure):
    _fields_ = [("kind", C.c_int32), ("epsilon", C.c_float), ("packed", C.c_void_p)]


# every symbol include/reinlife_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
ABI = [
    ("rl_last_error", C.c_char_p, []),
    ("rl_version", C.c_char_p, []),
    ("rl_create", C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    ("rl_destroy", None, [_P]),
    ("rl_bind_state", C.c_int, [_P, C.POINTER(State)]),
    ("rl_bind_error_flag", C.c_int, [_P, _P]),
    ("rl_bind_phase_profile", C.c_int, [_P, _P, C.c_int]),
    ("rl_reset_synthetic", C.c_int, [_P, C.c_int, _P, _P]),
    ("rl_reset_families", C.c_int, [_P, _P, _P]),
    ("rl_refill", C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P]),
    ("rl_observe", C.c_int, [_P, _P, _P]),
    ("rl_step", C.c_int, [_P, _P, C.POINTER(Tape), C.POINTER(StepOut), _P]),
    ("rl_step_split", C.c_int, [_P, _P, C.POINTER(StepOut), _P, _P]),
    ("rl_step_food", C.c_int, [_P, C.POINTER(Tape), _P, _P]),
    ("rl_update", C.c_int, [_P, C.POINTER(Tape), C.POINTER(UpdateOut), _P]),
    ("rl_tick", C.c_int, [_P, _P, C.POINTER(Tape), C.POINTER(StepOut), C.POINTER(UpdateOut), _P]),
    ("rl_tick_refill", C.c_int, [_P, _P, C.POINTER(StepOut), C.POINTER(UpdateOut), C.c_int, C.c_int, _P, _P]),
    ("rl_run_supported", C.c_int, [_P, C.POINTER(Brain), C.c_int]),
    ("rl_run", C.c_int, [_P, C.POINTER(Brain), C.c_int, C.c_int, _P, C.POINTER(StepOut), C.POINTER(_P), C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    ("rl_run_ex", C.c_int, [_P, C.POINTER(Brain), C.c_int, C.c_int, _P, C.POINTER(StepOut), C.POINTER(_P), C.c_int, _P, C.POINTER(RunOpts), _P]),
    ("rl_capture_transitions", C.c_int, [_P, _P, _P, _P, C.POINTER(StepOut), C.POINTER(Replay), C.c_int, _P]),
    ("rl_policy_n_params", C.c_int64, [C.c_int]),
    ("rl_policy_packed_floats", C.c_int64, [C.c_int]),
    ("rl_policy_pack_weights", C.c_int, [C.c_int, _P, _P]),
    ("rl_policy_forward", C.c_int, [C.c_int, _P, _P, C.c_int64, _P, _P]),
    ("rl_policy_work_bytes", C.c_size_t, [_P]),
    ("rl_bind_policy_work", C.c_int, [_P, _P]),
    ("rl_policy_act", C.c_int, [_P, C.POINTER(Brain), C.c_int, _P, _P, _P, _P, _P]),
    ("rl_set_option", C.c_int, [C.c_char_p, C.c_char_p]),
    ("rl_get_option", C.c_int, [_P, C.c_char_p]),
    ("rl_philox", None, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                         C.POINTER(C.c_uint32 * 4)]),
]

_lib = None


class ReinLifeHipError(RuntimeError):
    pass


def lib():
    """Load the HIP library, building it first when it is absent OR stale: build() compares the content digest stamped beside every
    object and the library (*.srchash) with the sources' and recompiles what differs -- a library older than its sources (an ABI change
    since it was built) is never loaded silently.  REINLIFE_HIP_LIB names another build instead (tuning: A/B libraries, the tuning
    library with the measurement switches); it is loaded as it is.  Never falls back to anything that is not the HIP library."""
    global _lib
    if _lib is None:
        path = os.environ.get("REINLIFE_HIP_LIB")
        if not path:
            path = _build.LIB_PATH
            try:   # current by its stamps: a few file digests, no lock, no compiler needed; otherwise build() (serialised by an flock,
                #    files moved into place atomically: N ranks finding it stale together compile once and never map a half-written file)
                if os.environ.get("REINLIFE_REBUILD") or not _build.library_is_current():
                    _build.build(force=bool(os.environ.get("REINLIFE_REBUILD")))
            except Exception as e:  # noqa: BLE001
                raise ReinLifeHipError("libreinlife_hip.so is missing or older than its sources and could not be built with hipcc: %s" % e)
        handle = C.CDLL(path)
        for name, res, args in ABI:
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def set_option(name, value=None):
    """rl_set_option: a process-level tuning / test switch (include/reinlife_hip.h "options"); handles created afterwards use it.
    value None restores what the environment said when the library was loaded."""
    check(lib().rl_set_option(name.encode(), None if value is None else str(value).encode()), "rl_set_option(%s)" % name)


def check(rc, what):
    if rc != 0:
        raise ReinLifeHipError("%s failed (%d): %s" % (what, rc, lib().rl_last_error().decode()))


def slot_cap_for(max_agents):
    """Capacity of the per-world agent arrays: births can overshoot max_agents up to 2n+1 (environment.py:501)."""
    return ((2 * max_agents + 2 + 63) // 64) * 64
