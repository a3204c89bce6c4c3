"""DeviceWorlds: R independent ReinLife worlds resident in HBM, driven through the C ABI (include/reinlife_hip.h).

PyTorch is plumbing only: it owns the device buffers and the stream; all compute is libreinlife_hip.so.
Layout = `rl_state` (struct-of-arrays over worlds; a world's agent list is its on-grid agents in row-major cell
order, i.e. the reference's env.agents order -- World/grid.py:60-67).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_STATE_SPEC = [  # name, torch dtype, shape suffix
    ("cell_type", torch.uint8, "C"), ("n_agents", torch.int32, ""), ("a_i", torch.uint8, "cap"),
    ("a_j", torch.uint8, "cap"), ("a_health", torch.int32, "cap"), ("a_age", torch.int32, "cap"),
    ("a_max_age", torch.int32, "cap"), ("a_gene", torch.int32, "cap"), ("a_brain", torch.int32, "cap"),
    ("a_uid", torch.int32, "cap"), ("a_flags", torch.uint8, "cap"), ("a_action", torch.int8, "cap"),
    ("a_fitness", torch.float64, "cap"), ("max_gene", torch.int32, ""), ("next_uid", torch.int32, ""),
    ("tick", torch.int32, ""), ("epoch", torch.int32, ""), ("best_uid", torch.int32, "best"),
    ("best_fit", torch.float64, "best"), ("best_brain", torch.int32, "best")]


# Pinned staging buffers for run()'s epsilon schedules, one small ring per device for the whole process: pinning a fresh buffer per
# launch (or per Environment: trainer() builds a new one on every call) costs more than a short launch's kernel.  A slot is
# [pinned host buffer, device buffer, event recorded behind the slot's last upload].
_EPS_RING = {}


_NP = {torch.uint8: np.uint8, torch.int8: np.int8, torch.int16: np.int16, torch.int32: np.int32, torch.float32: np.float32,
       torch.float64: np.float64}


def _public_raw_stream(device_index):
    return torch.cuda.current_stream(device_index).cuda_stream


# the current stream's handle for the C ABI, once per launch: torch's private accessor where this build has it (no Stream object per call,
# ~1 us), the public torch.cuda.current_stream(...).cuda_stream otherwise (a CPU-only wheel, a renamed symbol) -- same handle either way
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or _public_raw_stream


class _Snapshot:
    """DeviceWorlds.snapshot(): name -> numpy array over one host copy of the state arena; the typed views are made on first use."""

    def __init__(self, buf, layout):
        self._buf, self._layout, self._views = buf, layout, {}

    def __getitem__(self, name):
        v = self._views.get(name)
        if v is None:
            o, nb, dt, shape = self._layout[name]
            v = self._views[name] = self._buf[o:o + nb].view(_NP[dt]).reshape(shape)
        return v

    def __contains__(self, name):
        return name in self._layout

    def keys(self):
        return self._layout.keys()


class TapeRing:
    """The draws a host makes for a tick (Tape: food_k, food_u, repro_u, birth_k, produce_u, produce_choice), staged for the device.  The six
    arrays of a tape are carved out of ONE pinned host buffer and ONE device buffer -- one host-to-device copy per tape (a host that draws
    every tick uploads two tapes per tick) -- and there is a ring of SLOTS such pairs: every make() returns a FRESH Tape struct over the next
    slot's device buffer, so a caller may prepare the step's and the update's tape before launching either (until round 6 every tape aliased
    one device buffer and the second make() silently replaced the first one's draws).  A tape stays valid until the SLOTS-th make() after
    it; its slot's buffers are then rewritten in stream order -- behind every launch queued so far -- and the pinned buffer only once its
    last upload has executed.  With a CPU device (tests/test_host_api_cpu.py) the "upload" is a plain copy and nothing is pinned."""
    SLOTS = 4

    def __init__(self, n_worlds, cap, device, cur_stream=None):
        self.R, self.cap, self.device = n_worlds, cap, torch.device(device)
        self._cur_stream = cur_stream
        R = n_worlds
        spec = [("food_k", np.int32, (R, _lib.FOOD_TRIES)), ("food_u", np.float64, (R, _lib.FOOD_TRIES)), ("repro_u", np.float64, (R, cap)),
                ("birth_k", np.int32, (R, cap + 1)), ("produce_u", np.float64, (R,)), ("produce_choice", np.int32, (R,))]
        lay, off = {}, 0
        for name, dt, shape in spec:
            nb = int(np.prod(shape)) * np.dtype(dt).itemsize
            lay[name] = (off, nb, dt, shape)
            off += (nb + 255) // 256 * 256
        self.layout, self.nbytes = lay, off
        on_gpu = self.device.type == "cuda"
        self.slots = []
        for _ in range(self.SLOTS):
            host = torch.zeros(off, dtype=torch.uint8)
            if on_gpu:
                host = host.pin_memory()
            buf = host.numpy()
            dev = torch.zeros(off, dtype=torch.uint8, device=self.device)
            self.slots.append({"host": host, "dev": dev, "event": None,
                               "views": {n: buf[o:o + nb].view(dt).reshape(shape) for n, (o, nb, dt, shape) in lay.items()}})
        self.made = 0

    def make(self, tapes):
        R, cap = self.R, self.cap
        slot = self.slots[self.made % self.SLOTS]
        self.made += 1
        if slot["event"] is not None:
            slot["event"].synchronize()
        host = slot["views"]
        for w, t in enumerate(tapes):
            host["food_k"][w] = t["food_k"]
            host["food_u"][w] = t["food_u"]
            m = min(cap, len(t["repro_u"]))
            host["repro_u"][w, :m] = t["repro_u"][:m]
            host["repro_u"][w, m:] = 0
            m = min(cap + 1, len(t["birth_k"]))
            host["birth_k"][w, :m] = t["birth_k"][:m]
            host["birth_k"][w, m:] = 0
            host["produce_u"][w] = t["produce_u"]
            host["produce_choice"][w] = t["produce_choice"]
        if len(tapes) < R:   # (worlds beyond the list: zeros)
            for name in _lib.TAPE_FIELDS:
                host[name][len(tapes):] = 0
        slot["dev"].copy_(slot["host"], non_blocking=True)
        if self.device.type == "cuda":
            if slot["event"] is None:
                slot["event"] = torch.cuda.Event()
            slot["event"].record(self._cur_stream() if self._cur_stream else torch.cuda.current_stream(self.device))
        base = slot["dev"].data_ptr()
        tape = _lib.Tape(*[C.c_void_p(base + self.layout[n][0]) for n in _lib.TAPE_FIELDS])
        tape._keep = slot["dev"]   # (the struct keeps its buffer alive)
        return tape


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _Readback:
    """Host copies in flight (DeviceWorlds.readback): wait() blocks until THEY have arrived, not until the stream is idle."""

    def __init__(self, host, done):
        self.host, self.done = host, done

    def wait(self):
        self.done.synchronize()
        return [h.numpy() for h in self.host]


class DeviceWorlds:
    def __init__(self, n_worlds=1, width=30, height=30, max_agents=100, n_brains=2, static_families=True,
                 limit_reproduction=False, incentivize_killing=True, seed=0, slot_cap=None, device="cuda:0",
                 world_base=0):
        if not torch.cuda.is_available():
            raise _lib.ReinLifeHipError("DeviceWorlds needs an MI355X: torch.cuda.is_available() is False "
                                        "(there is no CPU fallback)")
        self.lib = _lib.lib()
        self.device = torch.device(device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.R, self.W, self.H, self.C = n_worlds, width, height, width * height
        self.cap = slot_cap or _lib.slot_cap_for(max_agents)
        self.max_agents, self.n_brains = max_agents, n_brains
        self.world_base = int(world_base)   # global replica id of world 0 of this handle (keys the Philox streams)
        self.cfg = _lib.Config(width, height, max_agents, n_brains, self.cap, n_worlds, int(static_families),
                               int(limit_reproduction), int(incentivize_killing), world_base, seed)
        self.handle = C.c_void_p()
        _lib.check(self.lib.rl_create(C.byref(self.cfg), C.byref(self.handle)), "rl_create")
        dims = {"C": (self.C,), "cap": (self.cap,), "best": (_lib.N_BEST,), "": ()}
        with torch.cuda.device(self.device):
            R, cap = self.R, self.cap
            # The world state and every small per-tick output live in ONE allocation (`_arena`, each array 256-byte aligned inside it), so
            # that a host which has to look at a world every tick -- Environment(rng="reference"): the reference's generator draws are
            # made on the host -- fetches all of it with ONE copy (snapshot()) instead of one small copy per array.  The kernels see
            # plain pointers, as before.
            spec = [(n, dt, (R,) + dims[suf]) for n, dt, suf in _STATE_SPEC]
            spec += [("actions", torch.int8, (R, cap)), ("n_acted", torch.int32, (R,)), ("reward", torch.float32, (R, cap)),
                     ("done", torch.uint8, (R, cap)), ("src1", torch.int16, (R, cap)), ("src2", torch.int16, (R, cap)),
                     ("n_post", torch.int32, (R,)), ("pre_counts", torch.int32, (R, 4)), ("err", torch.int32, (4,)),
                     ("out_q", torch.float32, (R, cap, 8))]
            self._layout, off = {}, 0
            for name, dt, shape in spec:
                nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dt).element_size()
                self._layout[name] = (off, nbytes, dt, shape)
                off += (nbytes + 255) // 256 * 256
            self._arena = torch.zeros(off, dtype=torch.uint8, device=self.device)
            self._arena_host = None
            carve = {name: self._arena[o:o + nb].view(dt).view(shape) for name, (o, nb, dt, shape) in self._layout.items()}
            self.s = {n: carve[n] for n, _, _ in _STATE_SPEC}
            self.s["best_uid"].fill_(-1)
            self.s["max_gene"].fill_(n_brains)
            self.actions, self.n_acted, self.reward, self.done = carve["actions"], carve["n_acted"], carve["reward"], carve["done"]
            self.src1, self.src2 = carve["src1"].fill_(-1), carve["src2"].fill_(-1)
            self.n_post, self.pre_counts, self.err, self.out_q = carve["n_post"], carve["pre_counts"], carve["err"], carve["out_q"]
            # +1 row of padding keeps 16-byte row reads of the policy kernel inside the allocation
            self.obs1 = torch.zeros((R * cap + 1, _lib.OBS_DIM), dtype=torch.float32, device=self.device)
            # Agent.state lives in a ping-pong pair: a tick writes the new observations into the other buffer, so the
            # observations the policy read for that tick stay available for transition capture
            self._obs2 = [torch.zeros((R * cap + 1, _lib.OBS_DIM), dtype=torch.float32, device=self.device) for _ in range(2)]
            self._cur = 0
            self.age1 = torch.zeros((R, cap), dtype=torch.int32, device=self.device)
            self.brain1 = torch.zeros((R, cap), dtype=torch.int32, device=self.device)
            self.refill_count = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.acted_total = torch.zeros(1, dtype=torch.int64, device=self.device)
            self.G = n_brains if static_families else 1
            self.trk_tick = torch.zeros((R, self.G, _lib.TRK_VARS), dtype=torch.float64, device=self.device)
            self.trk_sum = torch.zeros((R, self.G, _lib.TRK_VARS), dtype=torch.float64, device=self.device)
            self.trk_cnt = torch.zeros((R, self.G, _lib.TRK_VARS), dtype=torch.int32, device=self.device)
            self.trk_pop = torch.zeros((R, 3), dtype=torch.float64, device=self.device)
        self._state = _lib.State(*[_ptr(self.s[n]) for n in _lib.STATE_FIELDS])
        _lib.check(self.lib.rl_bind_state(self.handle, C.byref(self._state)), "rl_bind_state")
        _lib.check(self.lib.rl_bind_error_flag(self.handle, _ptr(self.err)), "rl_bind_error_flag")
        self.tracking = False
        self._trk_dirty = False     # a launch has added to trk_sum / trk_cnt / trk_pop since they were last zeroed
        self.replays = None
        self._build_step_out()
        self._work = None
        self._ticked = False
        self._brains = None
        self._tape_keep = None
        self._run_pair = None
        self._eps_keep = None
        self._capture_prob = False
        self._fused_key = self._fused = None
        self.launches = 0                        # kernel-launching C-ABI calls made by run() so far (bench.py reports it)
        self._side = None                        # side stream of readback()

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                self.lib.rl_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:  # noqa: BLE001
            pass

    # -- helpers ------------------------------------------------------------------------------------------------
    def _stream(self):
        # (the raw handle of torch's current stream on this device: ~0.3 us, where torch.cuda.current_stream() builds a Stream object
        # in ~5 us -- eight launches per tick of a host-driven loop)
        return C.c_void_p(_raw_stream(self._dev_index))

    def _cur_stream(self):
        """torch's current stream on this device as a Stream object (for Event.record), rebuilt only when the raw handle has changed."""
        raw = _raw_stream(self._dev_index)
        if getattr(self, "_cs_raw", None) != raw:
            self._cs_raw, self._cs = raw, torch.cuda.current_stream(self.device)
        return self._cs

    def obs_state(self):
        """[R, cap, 153] view of Agent.state (post-update observation)."""
        return self.obs2[: self.R * self.cap].view(self.R, self.cap, _lib.OBS_DIM)

    def obs_state_prime(self):
        """[R, cap, 153] view of Agent.state_prime (post-step observation)."""
        return self.obs1[: self.R * self.cap].view(self.R, self.cap, _lib.OBS_DIM)

    def check_error_flag(self):
        self.raise_on_error_flag(self.err.cpu().numpy())

    @staticmethod
    def raise_on_error_flag(e):
        if e[0] != 0:
            raise _lib.ReinLifeHipError("device error flag: code %d world %d detail (%d, %d)" % tuple(int(x) for x in e))

    def readback(self, tensors):
        """Small device tensors on their way to pinned host memory BEHIND everything queued on the current stream so far and NEXT TO
        whatever is queued afterwards: the copies run on a side stream that waits for an event recorded here, so a caller can queue
        the next multi-tick launch first and collect the values while it runs (`.wait()` -> list of numpy arrays).  The Tracker
        closes its intervals this way (Helpers/tracker.py): the device does not idle for the host's read-back, the statistics'
        arithmetic and the next launch's set-up."""
        cur = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        self._side.wait_event(ready)
        host = []
        with torch.cuda.stream(self._side):
            for t in tensors:
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                t.record_stream(self._side)
                host.append(h)
            done = torch.cuda.Event()
            done.record(self._side)
        return _Readback(host, done)

    SNAPSHOT_MAX_BYTES = 1 << 20

    def snapshot(self):
        """Host copy of the whole state arena (every rl_state array, actions, n_acted, reward, done, src1 / src2, step_split's counts, the
        error flag, the policy outputs) behind everything queued on the current stream: ONE device-to-host copy and ONE wait, whatever
        the number of arrays -> {name: numpy array}.  For hosts that look at their worlds every tick (a single world driven by the
        reference's generators); refused for big handles, whose callers read single worlds (world())."""
        nbytes = self._arena.numel()
        if nbytes > self.SNAPSHOT_MAX_BYTES:
            raise _lib.ReinLifeHipError("snapshot(): %d bytes of state; read single worlds of a handle this large with world()" % nbytes)
        if self._arena_host is None:
            self._arena_host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
            self._arena_np = self._arena_host.numpy()
            self._arena_done = torch.cuda.Event()
        self._arena_host.copy_(self._arena, non_blocking=True)
        self._arena_done.record(self._cur_stream())
        self._arena_done.synchronize()
        return _Snapshot(self._arena_np.copy(), self._layout)   # (a copy: the pinned buffer is reused by the next snapshot)

    # -- host <-> device state (parity I/O) ----------------------------------------------------------------------
    def load_world(self, w, snap):
        n = len(snap["i"])
        assert n <= self.cap
        dev = self.device
        self.s["cell_type"][w] = torch.as_tensor(np.asarray(snap["cell_type"], np.uint8), device=dev)
        self.s["n_agents"][w] = n
        for key in ("i", "j", "health", "age", "max_age", "gene", "brain", "uid", "flags", "action", "fitness"):
            t = self.s["a_" + key]
            t[w, :n] = torch.as_tensor(np.ascontiguousarray(snap[key]), device=dev).to(t.dtype)
        self.s["max_gene"][w] = int(snap.get("max_gene", self.n_brains))
        self.s["next_uid"][w] = int(snap.get("next_uid", (int(np.max(snap["uid"])) + 1) if n else 0))
        for key in ("best_uid", "best_fit", "best_brain"):
            if key in snap:
                self.s[key][w] = torch.as_tensor(np.asarray(snap[key]), device=dev).to(self.s[key].dtype)
        _lib.check(self.lib.rl_bind_state(self.handle, C.byref(self._state)), "rl_bind_state")  # state rewritten by the host

    def world(self, w):
        n = int(self.s["n_agents"][w].item())
        d = {k[2:]: self.s[k][w, :n].cpu().numpy() for k in self.s if k.startswith("a_")}
        d["cell_type"] = self.s["cell_type"][w].cpu().numpy()
        for key in ("max_gene", "next_uid", "tick", "epoch"):
            d[key] = int(self.s[key][w].item())
        for key in ("best_uid", "best_fit", "best_brain"):
            d[key] = self.s[key][w].cpu().numpy()
        return d

    def make_tape(self, tapes):
        """tapes: one dict per world (food_k, food_u, repro_u, birth_k, produce_u, produce_choice) -> device Tape (TapeRing.make): a FRESH
        struct over its own device buffer per call, valid until the fourth make_tape() after it."""
        if getattr(self, "_tape_ring", None) is None:
            self._tape_ring = TapeRing(self.R, self.cap, self.device, self._cur_stream)
        return self._tape_ring.make(tapes)

    # -- the path -----------------------------------------------------------------------------------------------
    @property
    def obs2(self):
        return self._obs2[self._cur]

    def _build_step_out(self):
        t = (self.trk_tick, self.trk_sum, self.trk_cnt, self.trk_pop) if self.tracking else (None, None, None, None)
        c = (self.n_post, self.age1, self.brain1) if self.replays is not None else (None, None, None)
        self._step_out = _lib.StepOut(_ptr(self.n_acted), _ptr(self.reward), _ptr(self.done), _ptr(self.src1), _ptr(self.obs1),
                                      _ptr(self.acted_total), *[_ptr(x) for x in t], *[_ptr(x) for x in c])

    def _next_upd_out(self):
        """update_env writes Agent.state into the OTHER buffer of the ping-pong pair, which then becomes current."""
        self._cur ^= 1
        return _lib.UpdateOut(_ptr(self.src2), _ptr(self._obs2[self._cur]))

    def prev_state(self):
        """[R, cap, 153] observations the policy read for the last tick (valid between a tick/update and the next one)."""
        return self._obs2[self._cur ^ 1][: self.R * self.cap].view(self.R, self.cap, _lib.OBS_DIM)

    def enable_tracking(self, on=True):
        """Accumulate the Tracker statistics (Helpers/tracker.py) inside step()/tick() launches."""
        self.tracking = bool(on)
        self._build_step_out()

    def enable_capture(self, capacity, with_prob=False):
        """Allocate one replay ring per brain (rl_replay): filled by capture_transitions() after a stand-alone tick, or by run() inside
        the multi-tick launch (every tick of it)."""
        self._capture_prob = bool(with_prob)
        self.replays = []
        arr = (_lib.Replay * self.n_brains)()
        for b in range(self.n_brains):
            r = {"state": torch.zeros((capacity, _lib.OBS_DIM), dtype=torch.float32, device=self.device),
                 "state_prime": torch.zeros((capacity, _lib.OBS_DIM), dtype=torch.float32, device=self.device),
                 "action": torch.zeros(capacity, dtype=torch.int8, device=self.device),
                 "reward": torch.zeros(capacity, dtype=torch.float32, device=self.device),
                 "done": torch.zeros(capacity, dtype=torch.uint8, device=self.device),
                 "prob": torch.zeros(capacity, dtype=torch.float32, device=self.device) if with_prob else None,
                 "age": torch.zeros(capacity, dtype=torch.int32, device=self.device),
                 "count": torch.zeros(1, dtype=torch.int64, device=self.device)}
            self.replays.append(r)
            arr[b] = _lib.Replay(*[_ptr(r[n]) for n in ("state", "state_prime", "action", "reward", "done", "prob", "age", "count")], capacity)
        self._replay_arr = arr
        self._build_step_out()  # the step launches now also emit n_post / age / brain of the post-step list

    def capture_transitions(self, with_policy_out=False):
        """trainer.py:95-96 on the device: call after step()/tick() of a tick whose actions are still in self.actions."""
        if self.replays is None:
            raise _lib.ReinLifeHipError("enable_capture() was not called")
        _lib.check(self.lib.rl_capture_transitions(self.handle, _ptr(self._obs2[self._cur ^ 1] if self._ticked else self._obs2[self._cur]),
                                                   _ptr(self.actions), _ptr(self.out_q) if with_policy_out else None,
                                                   C.byref(self._step_out), self._replay_arr, self.n_brains, self._stream()),
                   "rl_capture_transitions")

    def reset_tracking(self):
        if self._trk_dirty:   # (fresh accumulators are zero: a new Environment's first launch is not preceded by three memsets)
            self.trk_sum.zero_(); self.trk_cnt.zero_(); self.trk_pop[:, 1:].zero_()
            self._trk_dirty = False

    def set_actions(self, actions):
        if torch.is_tensor(actions):
            a = actions.to(device=self.device, dtype=torch.int8)
            assert tuple(a.shape) == (self.R, self.cap)
            self.actions.copy_(a)
            return
        # a host array: through a pinned staging buffer (two, alternating; each reused only once its last upload has executed)
        if getattr(self, "_act_ring", None) is None:
            self._act_ring = [[torch.zeros((self.R, self.cap), dtype=torch.int8).pin_memory(), None] for _ in range(2)]
            self._act_next = 0
        slot = self._act_ring[self._act_next & 1]
        self._act_next += 1
        if slot[1] is not None:
            slot[1].synchronize()
        a = np.asarray(actions)
        assert a.shape == (self.R, self.cap)
        slot[0].numpy()[...] = a      # (casts to int8 like the former as_tensor(...).to(int8))
        self.actions.copy_(slot[0], non_blocking=True)
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record(self._cur_stream())

    def step(self, actions=None, tape=None):
        if actions is not None:
            self.set_actions(actions)
        self._trk_dirty = self._trk_dirty or self.tracking
        _lib.check(self.lib.rl_step(self.handle, _ptr(self.actions), C.byref(tape) if tape is not None else None,
                                    C.byref(self._step_out), self._stream()), "rl_step")
        self._ticked = False  # Agent.state of this tick is still the current buffer

    def step_split(self, actions=None):
        """First half of a split step; returns the device tensor [R,4] (food, poison, super food, empty cells after movement)."""
        if actions is not None:
            self.set_actions(actions)
        self._trk_dirty = self._trk_dirty or self.tracking
        _lib.check(self.lib.rl_step_split(self.handle, _ptr(self.actions), C.byref(self._step_out), _ptr(self.pre_counts),
                                          self._stream()), "rl_step_split")
        self._ticked = False
        return self.pre_counts

    def step_food(self, tape):
        _lib.check(self.lib.rl_step_food(self.handle, C.byref(tape), _ptr(self.obs1), self._stream()), "rl_step_food")

    def update(self, tape=None):
        uo = self._next_upd_out()
        _lib.check(self.lib.rl_update(self.handle, C.byref(tape) if tape is not None else None, C.byref(uo), self._stream()),
                   "rl_update")
        self._ticked = True

    def tick(self, actions=None, tape=None):
        """step() + update_env() of one trainer-loop iteration in ONE kernel launch."""
        if actions is not None:
            self.set_actions(actions)
        uo = self._next_upd_out()
        self._trk_dirty = self._trk_dirty or self.tracking
        _lib.check(self.lib.rl_tick(self.handle, _ptr(self.actions), C.byref(tape) if tape is not None else None,
                                    C.byref(self._step_out), C.byref(uo), self._stream()), "rl_tick")
        self._ticked = True

    def tick_refill(self, threshold, n_agents):
        """tick() + refill(threshold, n_agents) in one launch (Philox draws)."""
        uo = self._next_upd_out()
        self._trk_dirty = self._trk_dirty or self.tracking
        _lib.check(self.lib.rl_tick_refill(self.handle, _ptr(self.actions), C.byref(self._step_out), C.byref(uo),
                                           threshold, n_agents, _ptr(self.refill_count), self._stream()), "rl_tick_refill")
        self._ticked = True

    def run_supported(self):
        return self._brains is not None and bool(self.lib.rl_run_supported(self.handle, self._brains, self.n_brains))

    def run(self, n_ticks, threshold=-1, n_agents=0, eps_schedule=None, trk_skip=0, want_q=False):
        """n_ticks x (act() + tick_refill(threshold, n_agents)) -- or + tick() when threshold < 0 -- in ONE launch when the
        configuration allows it (rl_run_ex: every world stays in LDS between ticks; the Tracker accumulators are maintained in
        the launch when tracking is on), else the same loop over the two launches.  Either way the buffers afterwards hold the
        last tick's outputs.
        eps_schedule: optional [n_ticks, n_brains] float32 (host array or device tensor) -- the brains' exploration rate in every
        tick (the reference's brains decay epsilon per episode); None = the rates given to set_brains().
        trk_skip: the Tracker's running sums leave out the first trk_skip ticks (episode 0, tracker.py:279-282).
        want_q: the policy's outputs of the LAST tick (Q values / PPO probabilities) are left in self.out_q (rl_run_opts.policy_out)."""
        if self._brains is None:
            raise _lib.ReinLifeHipError("set_brains() was not called")
        if n_ticks <= 0:
            return
        eps_host = None
        if eps_schedule is not None:
            if not torch.is_tensor(eps_schedule) or not eps_schedule.is_cuda:
                eps_host = np.ascontiguousarray(eps_schedule.cpu().numpy() if torch.is_tensor(eps_schedule) else eps_schedule, dtype=np.float32)
                assert eps_host.shape == (n_ticks, self.n_brains)
            else:
                eps_schedule = eps_schedule.to(torch.float32).contiguous()
                assert tuple(eps_schedule.shape) == (n_ticks, self.n_brains)
        # (with many worlds per GPU -- several per CU -- the two stand-alone launches are faster: 8.0e8 against 6.1e8 agent-steps/s
        # at 1024 worlds, the cross-world policy tiles waste fewer rows; the run_always option forces the single launch)
        # (what the decision depends on: the brains' kinds -- set_brains() drops the cached answer -- and the handle's option snapshot)
        if self._fused_key is None:
            self._fused_key = True
            self._fused = self.run_supported() and (self.R <= 768 or self.lib.rl_get_option(self.handle, b"run_always") == 1)
        fused = self._fused
        if not fused:
            if eps_schedule is not None and eps_host is None:
                eps_host = eps_schedule.cpu().numpy()   # ONE read-back, not one per tick
            keep = None
            if self.tracking and trk_skip > 0:   # like the kernel: the first trk_skip ticks stay out of the running sums, earlier sums stay
                keep = (self.trk_sum.clone(), self.trk_cnt.clone(), self.trk_pop[:, 1:].clone())
            for t in range(n_ticks):
                if eps_host is not None:
                    self._set_epsilons(eps_host[t].tolist())
                self.act(want_q=want_q or (self.replays is not None and self._capture_prob))
                if threshold >= 0:
                    self.tick_refill(threshold, n_agents)
                else:
                    self.tick()
                self.launches += 2
                if self.replays is not None:
                    self.capture_transitions(with_policy_out=self._capture_prob)
                if keep is not None and t + 1 == trk_skip:
                    self.trk_sum.copy_(keep[0]); self.trk_cnt.copy_(keep[1]); self.trk_pop[:, 1:].copy_(keep[2])
            return
        inline = eps_host is not None and eps_host.size <= _lib.EPS_INLINE_MAX
        if eps_host is not None and not inline:
            eps_schedule = self._stage_schedule(eps_host)
        if self._run_pair is None:
            self._run_pair = (C.c_void_p * 2)(_ptr(self._obs2[0]), _ptr(self._obs2[1]))
        cap = self.replays is not None
        # (a short schedule rides in the kernel arguments -- rl_run_opts.eps_schedule_on_host: the launch waits for no upload)
        eps_arg = C.c_void_p(eps_host.ctypes.data) if inline else _ptr(eps_schedule)
        opts = _lib.RunOpts(threshold, n_agents, _ptr(self.refill_count), eps_arg, trk_skip, 1 if inline else 0,
                            C.cast(self._replay_arr, C.c_void_p) if cap else None, _ptr(self.out_q) if (want_q or (cap and self._capture_prob)) else None)
        self._trk_dirty = self._trk_dirty or self.tracking
        _lib.check(self.lib.rl_run_ex(self.handle, self._brains, self.n_brains, n_ticks, _ptr(self.actions), C.byref(self._step_out),
                                      self._run_pair, self._cur, _ptr(self.src2), C.byref(opts), self._stream()), "rl_run_ex")
        self._eps_keep = None if inline else eps_schedule   # the launch reads a device table asynchronously
        self.launches += 1
        self._cur = (self._cur + n_ticks) & 1
        self._ticked = True

    def _stage_schedule(self, host):
        """A host [n_ticks, n_brains] float32 schedule -> device through the process-wide ring of pinned buffers (_EPS_RING).  A slot's
        host buffer is rewritten only once its previous upload has executed (its event); its device buffer is rewritten by a copy
        queued on this stream, i.e. behind the launch that read it when that launch was on this stream -- and eight further uploads
        lie between two uses of a slot."""
        n = host.size
        ring = _EPS_RING.setdefault(str(self.device), {"slots": [], "next": 0})
        if len(ring["slots"]) < 8:
            cap = max(8192, 1 << int(n - 1).bit_length())
            slot = [torch.empty(cap, dtype=torch.float32).pin_memory(), torch.empty(cap, dtype=torch.float32, device=self.device),
                    torch.cuda.Event()]
            ring["slots"].append(slot)
        else:
            slot = ring["slots"][ring["next"] % 8]
            ring["next"] += 1
            slot[2].synchronize()
            if slot[0].numel() < n:
                cap = 1 << int(n - 1).bit_length()
                slot[0], slot[1] = torch.empty(cap, dtype=torch.float32).pin_memory(), torch.empty(cap, dtype=torch.float32, device=self.device)
        slot[0][:n].copy_(torch.from_numpy(host.reshape(-1)))
        dev = slot[1][:n]
        dev.copy_(slot[0][:n], non_blocking=True)
        slot[2].record(torch.cuda.current_stream(self.device))
        return dev.view(host.shape)

    def _set_epsilons(self, eps):
        for b, e in enumerate(eps):
            self._brains[b].epsilon = float(e)

    def observe(self):
        _lib.check(self.lib.rl_observe(self.handle, _ptr(self.obs2), self._stream()), "rl_observe")
        return self.obs_state()

    def reset_synthetic(self, n_agents):
        _lib.check(self.lib.rl_reset_synthetic(self.handle, n_agents, _ptr(self.obs2), self._stream()),
                   "rl_reset_synthetic")

    def reset_families(self):
        """Environment.reset() for every world on the device: one agent per brain (gene = its index) at random cells."""
        _lib.check(self.lib.rl_reset_families(self.handle, _ptr(self.obs2), self._stream()), "rl_reset_families")

    def refill(self, threshold, n_agents):
        _lib.check(self.lib.rl_refill(self.handle, threshold, n_agents, _ptr(self.obs2), _ptr(self.refill_count),
                                      self._stream()), "rl_refill")

    # -- policy -------------------------------------------------------------------------------------------------
    def set_brains(self, brains):
        """brains: list of (kind:int, epsilon:float, packed: device float32 tensor)."""
        assert len(brains) == self.n_brains
        self._brain_keep = [b[2] for b in brains]
        arr = (_lib.Brain * len(brains))()
        for k, (kind, eps, packed) in enumerate(brains):
            assert packed.is_cuda and packed.dtype == torch.float32 and packed.is_contiguous()
            assert packed.numel() == self.lib.rl_policy_packed_floats(kind)
            arr[k] = _lib.Brain(kind, float(eps), packed.data_ptr())
        self._brains = arr
        self._fused_key = None   # (the fused / two-launch decision depends on the brains' kinds)
        if self._work is None:
            nbytes = self.lib.rl_policy_work_bytes(self.handle)
            self._work = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=self.device)
            _lib.check(self.lib.rl_bind_policy_work(self.handle, _ptr(self._work)), "rl_bind_policy_work")

    def act(self, want_q=False):
        """Agent.get_action for every agent of every world: obs_state -> self.actions (and self.out_q)."""
        if self._brains is None:
            raise _lib.ReinLifeHipError("set_brains() was not called")
        _lib.check(self.lib.rl_policy_act(self.handle, self._brains, self.n_brains, _ptr(self.obs2), _ptr(self.actions),
                                          _ptr(self.out_q) if want_q else None, _ptr(self._work), self._stream()),
                   "rl_policy_act")


def pack_brain_weights(kind, state_dict_flat, device="cuda:0"):
    """state-dict-order float32 vector (host) -> packed MFMA layout on the device."""
    lib = _lib.lib()
    flat = np.ascontiguousarray(state_dict_flat, dtype=np.float32)
    if flat.size != lib.rl_policy_n_params(kind):
        raise ValueError("brain kind %d expects %d parameters, got %d" % (kind, lib.rl_policy_n_params(kind), flat.size))
    packed = np.zeros(lib.rl_policy_packed_floats(kind), np.float32)
    _lib.check(lib.rl_policy_pack_weights(kind, flat.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)),
               "rl_policy_pack_weights")
    return torch.as_tensor(packed, device=device)


def policy_forward(kind, packed, obs, out=None):
    """Dense batch: obs [n,153] device float32 (allocation must extend >= 12 bytes past the last row, or n rows of a
    larger buffer) -> [n,8]."""
    lib = _lib.lib()
    n = obs.shape[0]
    if out is None:
        out = torch.empty((n, 8), dtype=torch.float32, device=obs.device)
    stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
    _lib.check(lib.rl_policy_forward(kind, _ptr(packed), _ptr(obs), n, _ptr(out), stream), "rl_policy_forward")
    return out
