"""CPU: libreinlife_hip.so loads, exports every symbol include/reinlife_hip.h declares, and its host-only entry
points (validation, weight packing, Philox) behave.  No compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from reinlife_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "reinlife_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rl_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 19
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export %s" % name
    assert declared == {n for n, _, _ in _lib.ABI}, "ctypes ABI table out of sync with the header"


def test_create_validates_like_the_reference():
    lib = _lib.lib()
    h = C.c_void_p()
    ok = _lib.Config(30, 30, 100, 2, 256, 1, 1, 0, 1, 0, 0)
    assert lib.rl_create(C.byref(ok), C.byref(h)) == 0
    acts = (C.c_int8 * 256)()
    assert lib.rl_step(h, acts, None, None, None) == -2          # RL_E_UNBOUND: rl_bind_state was not called
    assert b"rl_bind_state" in lib.rl_last_error()
    lib.rl_destroy(h)
    for bad in (dict(width=2), dict(height=2), dict(slot_cap=100), dict(slot_cap=128), dict(n_brains=0), dict(n_worlds=0),
                dict(width=100, height=100)):
        cfg = _lib.Config(30, 30, 100, 2, 256, 1, 1, 0, 1, 0, 0)
        for k, v in bad.items():
            setattr(cfg, k, v)
        assert lib.rl_create(C.byref(cfg), C.byref(h)) < 0, bad     # Grid asserts width/height >= 3 (grid.py:23-24)
        assert lib.rl_last_error()


def test_philox_matches_oracle_and_known_answers():
    from oracle import oracle as orc
    lib = _lib.lib()
    out = (C.c_uint32 * 4)()
    lib.rl_philox(0, 0, 0, 0, 0, 0, C.byref(out))
    assert list(out) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    rng = np.random.RandomState(0)
    for _ in range(50):
        a = [int(x) for x in rng.randint(0, 2**31, size=6)]
        lib.rl_philox(a[0] * 7919 + 3, a[1] % 5, a[2], a[3], a[4] % 10, a[5], C.byref(out))
        assert list(out) == orc.philox(a[0] * 7919 + 3, a[1] % 5, a[2], a[3], a[4] % 10, a[5])


def _bf16_planes_sum(u16, shape):
    """[..., 3 planes, 64 lanes, 8] bf16 -> the f32 values the three planes add up to."""
    parts = (u16.astype(np.uint32) << 16).view(np.float32).reshape(shape).astype(np.float64)
    return parts.sum(axis=-3)


@pytest.mark.parametrize("name", ["DQN", "D3QN", "PERD3QN", "PPO"])
def test_weight_packing_keeps_every_weight_exactly(name):
    """rl_policy_pack_weights: MFMA layers are stored as three bf16 planes (hi + mid + lo == the f32 weight, exactly),
    permuted into fragment order with the input layer's bias folded in as column 153; the other biases stay f32.  Every parameter must be recoverable, none duplicated."""
    from oracle import oracle as orc
    lib = _lib.lib()
    kind = _lib.KIND_BY_METHOD[name]
    n = lib.rl_policy_n_params(kind)
    assert n == orc.lib().rlo_policy_n_params(kind)
    # a 24-bit-mantissa pattern per parameter, all distinct and nonzero
    flat = ((np.arange(n, dtype=np.float64) + 1.0) * (1.0 + 2.0 ** -23) * 2.0 ** -10).astype(np.float32)
    assert len(np.unique(flat)) == n
    packed = np.zeros(lib.rl_policy_packed_floats(kind), np.float32)
    assert lib.rl_policy_pack_weights(kind, flat.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)) == 0
    u16 = packed.view(np.uint16)
    # (hidden width, [(tin, tout, head outputs)...]) per kind, in packed order
    h1, branches = {"DQN": (128, [(4, 2, 8)]), "D3QN": (128, [(4, 4, 8), (4, 4, 1)]), "PERD3QN": (128, [(4, 4, 8), (4, 4, 1)]),
                    "PPO": (256, [(8, 8, 8)])}[name]
    got = []
    off = 0  # in 4-byte units
    t1 = h1 // 32
    cnt = 10 * t1 * 3 * 64 * 4
    got.append(_bf16_planes_sum(u16[off * 2:(off + cnt) * 2], (10, t1, 3, 64, 8)).reshape(-1))
    off += cnt
    for tin, tout, nout in branches:
        cnt = 2 * tin * tout * 3 * 64 * 4
        got.append(_bf16_planes_sum(u16[off * 2:(off + cnt) * 2], (2 * tin, tout, 3, 64, 8)).reshape(-1))
        off += cnt
        got.append(packed[off:off + 32 * tout].astype(np.float64))       # hidden bias, accumulator order
        off += 32 * tout
        cnt = tout * 2 * 3 * 64 * 4
        got.append(_bf16_planes_sum(u16[off * 2:(off + cnt) * 2], (tout * 2, 3, 64, 8)).reshape(-1))  # head fragments
        off += cnt
        got.append(packed[off:off + nout].astype(np.float64))             # head bias
        off += nout
    assert off == len(packed)
    vals = np.concatenate(got)
    nz = np.sort(vals[vals != 0])
    used = n - (257 if name == "PPO" else 0)          # PPO's value head is not evaluated when acting (PPO.py:164-169)
    want = np.sort(flat[:used].astype(np.float64))   # state-dict order: PPO's unused fc_v comes last
    assert len(nz) == used and np.array_equal(nz, want)
    assert lib.rl_policy_pack_weights(9, flat.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)) < 0


def test_device_worlds_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from reinlife_amd.worlds import DeviceWorlds
    with pytest.raises(_lib.ReinLifeHipError):
        DeviceWorlds(n_worlds=1)
