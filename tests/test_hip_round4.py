"""GPU, round 4: parity AT THE BENCHED SIZE (256 worlds = 256 staggered workgroups) for the two instantiations of the multi-tick
kernel that round 3 only covered at 6-14 worlds -- k_run<512, fixed, kKindAll> (BASELINE configs[4] per GPU: PPO + PERD3QN,
static_families=False; PPO.py:101-106,164-169, PERD3QN.py:198-210, environment.py:521-547,728-739) and k_run<512, fixed, dueling,
TRAIN 1> (what trainer() / bench.py's api_trainer launch: per-tick epsilon schedule + Tracker, tracker.py:107-282) -- each against the
oracle fed the launch's actions and as ONE launch of 200 ticks against 200 launches of one; replica sharding through the API on the
GPU (the Tracker's collective over RCCL at world size 1)."""
import argparse
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from test_hip_round2 import _cmp_rows, _cmp_state, _same_device_state  # noqa: E402
from test_hip_round3 import _same_tracker  # noqa: E402

SEED, TICKS = 20260928, 200


def _worlds(workload, tracking=False, block=None):
    sys.path.insert(0, ROOT)
    import bench
    from test_hip_round2 import _need_run
    dw = bench.make_worlds(argparse.Namespace(worlds=256, workload=workload, seed=SEED), 0, "cuda:0")
    if tracking:
        dw.enable_tracking(True)
    _need_run(dw, block)
    return dw, bench.WORKLOADS[workload]


def _schedule(n_brains, t0, n):
    """A per-tick exploration schedule that changes every tick and differs between the brains (float32, as Environment.run uploads it)."""
    t = np.arange(t0, t0 + n, dtype=np.float64)[:, None]
    return (0.35 * 0.99 ** t * (1.0 + 0.5 * np.arange(n_brains)[None, :])).astype(np.float32)


def _against_the_oracle(workload, train):
    from oracle import oracle as orc
    dw, wl = _worlds(workload, tracking=train)
    nb = len(wl["brains"])
    ow = orc.OracleWorlds(n_worlds=256, seed=SEED, width=30, height=30, max_agents=100, n_brains=nb, static_families=wl["static_families"])
    ow.reset_synthetic(100)
    # round 5: the policy half at this size too (tests/policy_check.py) -- greedy brains of the TRAIN 0 launch only (an exploring
    # schedule changes per tick; its selection rule is covered at 16 worlds in test_hip_round2.py)
    import bench
    from policy_check import PolicyCheck
    pc = None if train else PolicyCheck(wl["brains"], [bench.brain_weights(n, 100 + k) for k, n in enumerate(wl["brains"])], [0.0] * nb)
    steps = 0
    for t in range(TICKS):
        n0 = ow.s["n_agents"].copy()
        with_q, with_a = pc is not None and t % 10 == 4, pc is not None and t % 10 == 9
        if with_q or with_a:
            pc.before(ow)
        if train:
            dw.run(1, 70, 100, eps_schedule=_schedule(nb, t, 1), trk_skip=1 if t == 0 else 0)
        else:
            dw.run(1, 70, 100, want_q=with_q)
        acts = dw.actions.cpu().numpy()
        if with_q or with_a:
            pc.after(acts, dw.out_q.cpu().numpy() if with_q else None, "tick %d" % t)
        if train and t == 0:
            keep = (ow.trk_sum.copy(), ow.trk_cnt.copy(), ow.trk_pop.copy())
        ow.step(acts)
        if train:
            if t == 0:   # (episode 0 stays out of the running sums)
                ow.trk_sum[:] = keep[0]; ow.trk_cnt[:] = keep[1]; ow.trk_pop[:, 1:] = keep[2][:, 1:]
            assert np.array_equal(dw.trk_tick.cpu().numpy(), ow.trk_tick, equal_nan=True), ("trk_tick", t)
        full = t % 10 == 9
        if full:
            n1 = ow.s["n_agents"].copy()
            _cmp_rows(dw.obs_state_prime().cpu().numpy(), ow.obs1, n1, "tick %d obs1" % t)
            _cmp_rows(dw.reward.cpu().numpy(), ow.reward, n1, "tick %d reward" % t)
            _cmp_rows(dw.done.cpu().numpy(), ow.done, n1, "tick %d done" % t)
            _cmp_rows(dw.src1.cpu().numpy(), ow.src1, n1, "tick %d src1" % t)
        ow.update(); ow.refill(70, 100)
        assert np.array_equal(dw.n_acted.cpu().numpy(), n0)
        steps += int(n0.sum())
        for key in ("cell_type", "n_agents", "tick", "epoch", "next_uid", "max_gene"):
            assert np.array_equal(dw.s[key].cpu().numpy().reshape(ow.s[key].shape), ow.s[key]), (t, key)
        if full:
            dw.check_error_flag()
            _cmp_state(dw, ow, "tick %d" % t)
            _cmp_rows(dw.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)
            if train:
                for name in ("trk_sum", "trk_cnt"):
                    assert np.array_equal(getattr(dw, name).cpu().numpy(), getattr(ow, name), equal_nan=True), (name, t)
                assert np.array_equal(dw.trk_pop.cpu().numpy()[:, 1:], ow.trk_pop[:, 1:]), ("trk_pop", t)
    assert steps > 4_000_000 and int(dw.acted_total.item()) == steps and int(dw.refill_count.item()) > 300
    if pc is not None:
        assert pc.rows > 800_000 and pc.q_rows > 400_000 and pc.max_dq < 1e-5 and pc.ppo_mismatches <= 3
    return dw, ow


def _one_launch_against_many(workload, train):
    a, wl = _worlds(workload, tracking=train)
    b, _ = _worlds(workload, tracking=train)
    nb = len(wl["brains"])
    if train:
        a.run(TICKS, 70, 100, eps_schedule=_schedule(nb, 0, TICKS), trk_skip=1)
        for t in range(TICKS):
            b.run(1, 70, 100, eps_schedule=_schedule(nb, t, 1), trk_skip=1 if t == 0 else 0)
    else:
        a.run(TICKS, 70, 100)
        for _ in range(TICKS):
            b.run(1, 70, 100)
    a.check_error_flag(); b.check_error_flag()
    _same_device_state(a, b, "%d ticks" % TICKS)
    assert int(a.acted_total.item()) == int(b.acted_total.item()) and int(a.refill_count.item()) == int(b.refill_count.item()) > 300
    _cmp_rows(a.actions.cpu().numpy(), b.actions.cpu().numpy(), a.n_acted.cpu().numpy(), "actions")
    for name in ("reward", "done", "src1", "src2"):
        assert np.array_equal(getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy()), name
    assert np.array_equal(a.obs_state_prime().cpu().numpy(), b.obs_state_prime().cpu().numpy())
    if train:
        _same_tracker(a, b, "%d ticks" % TICKS)


def test_c5_size_mixed_kind_launch_tracks_the_oracle_for_200_ticks():
    """k_run<512, fixed, kKindAll, TRAIN 0> on 256 workgroups (bench.py's `c5` leg): integer state every tick, both observation passes
    and every per-agent output every 10 ticks, non-static families (max_gene, best agents)."""
    _against_the_oracle("c5", train=False)


def test_c5_size_one_launch_of_200_ticks_equals_200_launches_of_one():
    _one_launch_against_many("c5", train=False)


def test_trainer_instantiation_at_the_benched_size_tracks_the_oracle_for_200_ticks():
    """k_run<512, fixed, dueling, TRAIN 1> on 256 workgroups (what bench.py's api_trainer / trainer() launch): exploring brains on a
    per-tick schedule, the Tracker's per-tick values every tick and its running sums every 10."""
    _against_the_oracle("c4", train=True)


def test_trainer_instantiation_one_launch_of_200_ticks_equals_200_launches_of_one():
    _one_launch_against_many("c4", train=True)


def test_c5_trainer_instantiation_one_launch_equals_many():
    """... and the mixed-kind TRAIN instantiation (k_run<512, fixed, kKindAll, TRAIN 1>: trainer() with configs[4]'s brains)."""
    _one_launch_against_many("c5", train=True)


# ---------------------------------------------------------------------------------------------------------------------
# one arithmetic on every path: a replica's trajectory does not depend on how many worlds share its GPU
# ---------------------------------------------------------------------------------------------------------------------
def test_sixteen_wave_workgroups_equal_eight_wave_workgroups_at_the_benched_size(hip_option):
    """k_run<1024> (four waves per tile, policy_quad: the same arithmetic per accumulator as the two-wave tiles of k_run<512>) against
    k_run<512> at 256 worlds: one launch of 120 ticks each, world state, both observation buffers, actions and the last tick's outputs bit
    for bit -- and the TRAIN instantiation with a per-tick epsilon schedule and the Tracker."""
    for train in (False, True):
        a, wl = _worlds("c4", tracking=train)
        hip_option("world_block", 1024)
        b, _ = _worlds("c4", tracking=train, block=1024)
        hip_option("world_block", None)
        kw = dict(eps_schedule=_schedule(2, 0, 120), trk_skip=1) if train else {}
        a.run(120, 70, 100, **kw); b.run(120, 70, 100, **kw)
        a.check_error_flag(); b.check_error_flag()
        _same_device_state(a, b, "120 ticks, train=%s" % train)
        assert int(a.acted_total.item()) == int(b.acted_total.item()) > 2_000_000 and int(a.refill_count.item()) == int(b.refill_count.item()) > 100
        _cmp_rows(a.actions.cpu().numpy(), b.actions.cpu().numpy(), a.n_acted.cpu().numpy(), "actions")
        assert np.array_equal(a.obs_state_prime().cpu().numpy(), b.obs_state_prime().cpu().numpy())
        if train:
            _same_tracker(a, b, "120 ticks")


def test_sixteen_wave_workgroups_with_more_than_four_tiles(hip_option):
    """Three dueling brains on 120 agents = six 32-row tiles per world: k_run<1024> takes them in two rounds of four-wave tiles with the rows
    read from memory (the tiles' exchange slices alias the LDS mirror); against the two-launch loop, every tick for 40 ticks."""
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    from test_hip_round2 import _weights
    hip_option("world_block", 1024)
    # (24 x 24: with three brains' constants next to a 30 x 30 world the exchange slices no longer fit into LDS -- rl_run says so)
    cfg = dict(width=24, height=24, max_agents=120, n_brains=3, static_families=True, limit_reproduction=False, incentivize_killing=True)
    pair = []
    for _ in range(2):
        dw = DeviceWorlds(n_worlds=12, seed=31, **cfg)
        dw.set_brains([(_lib.PERD3QN, 0.0, pack_brain_weights(_lib.PERD3QN, _weights("PERD3QN", 3))),
                       (_lib.D3QN, 0.1, pack_brain_weights(_lib.D3QN, _weights("D3QN", 4))),
                       (_lib.PERD3QN, 0.0, pack_brain_weights(_lib.PERD3QN, _weights("PERD3QN", 5)))])
        dw.reset_synthetic(120)
        pair.append(dw)
    fused, loop = pair
    from test_hip_round2 import _need_run
    _need_run(fused, 1024)
    for t in range(40):
        fused.run(1, 90, 120)
        loop.act(); loop.tick_refill(90, 120)
        fused.check_error_flag()
        _same_device_state(fused, loop, "tick %d" % t)
        _cmp_rows(fused.actions.cpu().numpy(), loop.actions.cpu().numpy(), fused.n_acted.cpu().numpy(), "tick %d actions" % t)
    assert int(fused.acted_total.item()) == int(loop.acted_total.item()) > 40_000


@pytest.mark.parametrize("workload", ["c4", "c5"])
def test_one_handle_of_1024_worlds_equals_four_handles_of_256(workload):
    """1 x 1,024 worlds (above 768 worlds per handle DeviceWorlds.run loops over rl_policy_act + rl_tick_refill: the stand-alone policy
    kernel, 256-thread world kernels) against 4 x 256 worlds with world_base 0 / 256 / 512 / 768 (the multi-tick launch): 60
    policy-driven ticks with refills, identical worlds, observations AND actions -- the stand-alone policy kernels run the tiles of
    rl_run's policy half (DQN.py:126-139, D3QN.py:161-173, PPO.py:164-169 in one summation order everywhere), no option set."""
    sys.path.insert(0, ROOT)
    import bench

    def make(n, base):
        return bench.make_worlds(argparse.Namespace(worlds=256, workload=workload, seed=SEED), 0, "cuda:0", n_worlds=n, world_base=base)
    big = make(1024, 0)
    parts = [make(256, 256 * q) for q in range(4)]
    for chunk in (1, 29, 30):
        big.run(chunk, 70, 100)
        for part in parts:
            part.run(chunk, 70, 100)
        big.check_error_flag()
        for q, part in enumerate(parts):
            part.check_error_flag()
            sl = slice(256 * q, 256 * (q + 1))
            n = part.s["n_agents"].cpu().numpy()
            for key in part.s:
                got, want = big.s[key][sl].cpu().numpy(), part.s[key].cpu().numpy()
                if key.startswith("a_"):
                    _cmp_rows(got, want, n, "%s part %d %s" % (workload, q, key))
                else:
                    assert np.array_equal(got, want), (workload, q, key)
            _cmp_rows(big.obs_state()[sl].cpu().numpy(), part.obs_state().cpu().numpy(), n, "obs2 part %d" % q)
            acted = part.n_acted.cpu().numpy()
            assert np.array_equal(big.n_acted[sl].cpu().numpy(), acted)
            _cmp_rows(big.actions[sl].cpu().numpy(), part.actions.cpu().numpy(), acted, "actions part %d" % q)
    assert not big._fused and all(p._fused for p in parts)
    assert int(big.acted_total.item()) == sum(int(p.acted_total.item()) for p in parts) > 4_000_000
    assert int(big.refill_count.item()) == sum(int(p.refill_count.item()) for p in parts) > 100


# ---------------------------------------------------------------------------------------------------------------------
# replica sharding through the API on the GPU: the Tracker's collective over RCCL (world size 1), world_base from the rank
# ---------------------------------------------------------------------------------------------------------------------
def test_trainer_under_a_process_group_shards_by_rank_and_reduces_its_tracker_over_rccl():
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = r'''
import json, os, sys, warnings
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from reinlife_amd import Models
from reinlife_amd.Helpers.trainer import trainer
use_dist = sys.argv[1] == "dist"
if use_dist:
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
torch.manual_seed(3)
brains = [Models.PERD3QN(), Models.D3QN()]
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    env = trainer(brains, n_episodes=120, update_interval=40, n_worlds=24, save=False, print_results=False, seed=77,
                  synthetic_agents=100, refill_below=70)
print(json.dumps({"results": env.tracker.results, "collectives": env.tracker.collectives_executed, "rank": env.rank, "base": env.world_base,
                  "device": str(env.device), "has_dist": env.dist is not None, "acted": int(env.worlds.acted_total.item())}))
if use_dist:
    dist.barrier(); dist.destroy_process_group()
''' % ROOT
    import json
    outs = {}
    for mode in ("plain", "dist"):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        out = subprocess.run([sys.executable, "-c", code, mode], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        outs[mode] = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])   # (RCCL prints its banner at exit)
    p, d = outs["plain"], outs["dist"]
    assert not p["has_dist"] and p["collectives"] == 0
    assert d["has_dist"] and d["collectives"] == 3 and d["rank"] == 0 and d["base"] == 0 and d["device"] == "cuda:0"
    assert p["acted"] == d["acted"] > 100_000
    assert json.dumps(p["results"], sort_keys=True) == json.dumps(d["results"], sort_keys=True)
    assert len(p["results"]["Avg Number of Populations"]) == 3


def test_random_configurations_through_the_launch_64_cases():
    """tools/fuzz_parity.py, 64 cases (round 3: 16): random world shapes, capacities, brain lists of any kinds, family modes and launch
    modes (plain / Tracker + schedule / capture) against the oracle and against tick-by-tick launches."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "64", "11"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "fuzz ok: 64 cases" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_policy_act_with_more_brains_than_one_launch_holds():
    """Twelve brains of all four kinds (more than the eight a policy launch serves: the canonical tiles in two batches; rl_run does not cover it,
    DeviceWorlds.run loops over the two launches): Q values / probabilities against the oracle's f32 forward (1e-5), the selected actions
    against the oracle's rule where they are clear of a tie, and 20 ticks of worlds against the oracle fed the actions."""
    import torch
    from oracle import oracle as orc
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    from test_hip_round2 import _weights
    names = ["DQN", "PPO", "D3QN", "PERD3QN"] * 3
    eps = [0.0, 0.0, 0.0, 0.2] * 3
    wts = [_weights(n, 900 + k) for k, n in enumerate(names)]
    cfg = dict(width=30, height=30, max_agents=100, n_brains=len(names), static_families=True, limit_reproduction=False, incentivize_killing=True)
    dw = DeviceWorlds(n_worlds=10, seed=1212, world_base=3, **cfg)
    ow = orc.OracleWorlds(n_worlds=10, seed=1212, world_base=3, **cfg)
    dw.set_brains([(_lib.KIND_BY_METHOD[n], e, pack_brain_weights(_lib.KIND_BY_METHOD[n], w)) for n, e, w in zip(names, eps, wts)])
    dw.reset_synthetic(100); ow.reset_synthetic(100)
    assert not dw.run_supported()
    checked = 0
    for t in range(20):
        n = ow.s["n_agents"].copy()
        dw.act(want_q=True)
        torch.cuda.synchronize()
        acts, q = dw.actions.cpu().numpy().copy(), dw.out_q.cpu().numpy()
        for b, name in enumerate(names):
            ws, ks = np.nonzero((np.arange(dw.cap)[None, :] < n[:, None]) & (ow.s["a_brain"] == b))
            if not len(ws):
                continue
            want = orc.policy_forward(orc.KIND_BY_NAME[name], wts[b], ow.obs2[ws, ks])
            np.testing.assert_allclose(q[ws, ks], want, rtol=0, atol=1e-5, err_msg="%s (brain %d) tick %d" % (name, b, t))
            if eps[b] == 0.0 and name != "PPO":
                srt = np.sort(want, axis=1)
                clear = srt[:, -1] - srt[:, -2] > 1e-5
                assert np.array_equal(acts[ws, ks][clear], want.argmax(1)[clear]), (name, b, t)
                checked += int(clear.sum())
        ow.step(acts); ow.update(); ow.refill(70, 100)
        dw.tick_refill(70, 100)
        dw.check_error_flag()
        _cmp_state(dw, ow, "tick %d" % t)
        _cmp_rows(dw.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)
    assert checked > 3000
    # ... and DeviceWorlds.run() (the two-launch loop for this configuration) continues the same worlds
    dw.run(5, 70, 100)
    dw.check_error_flag()
    assert int(dw.s["tick"].max().item()) >= 5
