"""CPU: the oracle (oracle/rl_oracle.c) reproduces every golden vector recorded from the real reference."""
import numpy as np
import pytest

import golden_io
from oracle import oracle as orc


def make_oracle(n_worlds, **cfg):
    return orc.OracleWorlds(n_worlds=n_worlds, seed=0, **cfg)


@pytest.mark.parametrize("path", golden_io.trace_files("trace_"), ids=lambda p: p.split("/")[-1])
def test_oracle_replays_reference_traces(path):
    golden_io.replay_trace(make_oracle, path, n_worlds=1)


@pytest.mark.parametrize("path", golden_io.trace_files("micro_"), ids=lambda p: p.split("/")[-1])
def test_oracle_replays_quirk_microworlds(path):
    golden_io.replay_trace(make_oracle, path, n_worlds=2)


def _load(name):
    return np.load([p for p in golden_io.trace_files("micro_") if name in p][0])


def test_fixtures_actually_exhibit_the_quirks():
    """Guards the fixtures themselves: each micro world shows the behaviour it is named after."""
    tr = _load("follow_down_vanish")
    assert tr["n0"][0] == 2 and tr["step_n"][0] == 1  # follower vanished from the grid
    assert _load("follow_up_ok")["step_n"][0] == 2
    assert _load("follow_right_vanish")["step_n"][0] == 1
    assert _load("follow_left_ok")["step_n"][0] == 2
    assert _load("swap_earlier_vanishes")["step_n"][0] == 1
    tr = _load("mutual_attack")
    assert list(tr["step_health"][0][:2]) == [0, 100]
    tr = _load("attacked_then_eats")
    assert 40 in list(tr["step_health"][0][:2]) and not (tr["step_flags"][0][:2] & 1).all()
    tr = _load("conflict_chain")
    assert tr["step_n"][0] == 3 and (tr["step_i"][0][:3] == tr["init_i"][:3]).all()
    tr = _load("super_food")
    assert tr["step_max_age"][0][0] == 60 and tr["step_obs"][0][0][152] == 1.0
    tr = _load("healthmap_float")
    assert np.any((tr["step_obs"][0][:, 49:98] > 0) & (tr["step_obs"][0][:, 49:98] < 1))
    tr = _load("healthmap_int")
    hp = tr["step_obs"][0][:, 49:98]
    assert set(np.unique(hp)).issubset({-1.0, 0.0, 1.0})
    tr = _load("alone_reward_zero")
    assert tr["step_reward"][0][0] == 0.0
    tr = _load("old_age_death")
    assert (tr["step_flags"][0][:3] & 1).any() and np.any(tr["step_obs"][0][:, 98:147] == 1.0)
    tr = _load("best_agents_nonstatic")
    assert (tr["upd_best_uid"][0] >= 0).sum() == 1


def test_oracle_policy_forward_matches_reference_networks():
    """fp32 MLPs vs the reference's torch modules on the same weights: 1e-5 (north_star tolerance)."""
    m = np.load(golden_io.GOLDEN_DIR + "/models.npz")
    for name in ("DQN", "D3QN", "PERD3QN", "PPO"):
        out = orc.policy_forward(orc.KIND_BY_NAME[name], m[name + "_weights"], m["obs"])
        np.testing.assert_allclose(out, m[name + "_out"], rtol=0, atol=1e-5, err_msg=name)
        # greedy actions agree wherever the reference's top-2 gap is above the tolerance
        srt = np.sort(m[name + "_out"], axis=1)
        clear = (srt[:, -1] - srt[:, -2]) > 1e-4
        assert (out.argmax(1)[clear] == m[name + "_greedy"][clear]).all()


def test_oracle_philox_known_answer():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors): zero and 'pi' counters/keys."""
    # our wrapper maps (seed, epoch=0, world, tick, site, index) -> counter (index, site, tick, world), key (lo, hi)
    assert orc.philox(0, 0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    ctr = (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344)
    key = (0xa4093822, 0x299f31d0)
    got = orc.philox((key[1] << 32) | key[0], 0, ctr[3], ctr[2], ctr[1], ctr[0])
    assert got == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    got = orc.philox(0xffffffffffffffff, 0, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert got == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


def test_oracle_synthetic_reset_and_free_run():
    """Synthetic worlds: exactly n agents, reset-rule food statistics, deterministic in (seed, world)."""
    ow = orc.OracleWorlds(n_worlds=16, seed=42)
    ow.reset_synthetic(100)
    assert (ow.s["n_agents"] == 100).all()
    ct = ow.s["cell_type"]
    assert ((ct == orc.AGENT).sum(1) == 100).all() and ((ct == orc.SUPER).sum(1) == 1).all()
    assert 60 < (ct == orc.FOOD).sum(1).mean() < 120 and 25 < (ct == orc.POISON).sum(1).mean() < 65
    other = orc.OracleWorlds(n_worlds=16, seed=42)
    other.reset_synthetic(100)
    assert np.array_equal(ct, other.s["cell_type"])
    assert not np.array_equal(ct[0], ct[1])
    # row-major invariant + a few free-running Philox ticks keep it
    rng = np.random.RandomState(0)
    for _ in range(10):
        acts = rng.randint(0, 8, size=(16, ow.cap)).astype(np.int8)
        ow.step(acts)
        ow.update()
        for w in range(16):
            n = int(ow.s["n_agents"][w])
            cells = ow.s["a_i"][w, :n].astype(int) * 30 + ow.s["a_j"][w, :n]
            assert (np.diff(cells) > 0).all()
            assert (ow.s["cell_type"][w].reshape(-1)[cells] == orc.AGENT).all()
            assert (ow.s["cell_type"][w] == orc.AGENT).sum() == n
    assert ow.refill(70, 100) >= 0


def test_oracle_forward_matches_the_reference_on_its_pretrained_brains():
    """The reference's own trained brains (pretrained/All/*, loaded by the real reference through load_model= in
    oracle/gen_golden_pretrained.py): f32 forward within 1e-5 of the output scale (trained Q values reach ~30), and the
    batched numpy forward of the CPU baseline agrees too."""
    import json
    from oracle import cpu_bench
    p = np.load(golden_io.GOLDEN_DIR + "/pretrained.npz")
    meta = json.loads(bytes(p["meta"]).decode())
    keys = json.load(open(golden_io.GOLDEN_DIR + "/state_dict_keys.json"))
    for name in ("DQN", "D3QN", "PERD3QN", "PPO"):
        assert meta[name]["keys"] == keys[name], name   # the .pt files carry exactly the key names / shapes the brains expose
        want = p[name + "_out"]
        scale = max(1.0, float(np.abs(want).max()))
        out = orc.policy_forward(orc.KIND_BY_NAME[name], p[name + "_weights"], p["obs"])
        np.testing.assert_allclose(out / scale, want / scale, rtol=0, atol=1e-5, err_msg=name)
        out2 = cpu_bench.forward(name, cpu_bench.unpack(name, p[name + "_weights"]), p["obs"])
        np.testing.assert_allclose(out2 / scale, want / scale, rtol=0, atol=1e-5, err_msg=name + " (numpy)")
