"""GPU, round 2: the kernel instantiations and workloads the first suite did not reach -- 256/512-thread world kernels,
> 768 worlds per launch, the configs[4] workload (PPO + PERD3QN, non-static, policy-driven, refills), action agreement with
an f32 forward over a million free-running rows, within-row dynamic range of the split-precision policy, the RCCL path of
bench.py at world size 1, and replica-layout independence (world_base)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_io

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _weights(kind, seed):
    sys.path.insert(0, ROOT)
    import bench
    return bench.brain_weights(kind, seed)


def _cmp_state(dw, ow, tag):
    n = ow.s["n_agents"]
    for key in dw.s:
        got, want = dw.s[key].cpu().numpy(), ow.s[key]
        if key.startswith("a_"):
            for w in range(ow.R):
                assert np.array_equal(got[w, : n[w]], want[w, : n[w]]), (tag, key, w)
        else:
            assert np.array_equal(got.reshape(want.shape), want), (tag, key)


def _cmp_rows(got, want, n, tag):
    for w in range(len(n)):
        assert np.array_equal(got[w, : n[w]], want[w, : n[w]]), (tag, w)


# ---------------------------------------------------------------------------------------------------------------------
# k_world<256> / <512> / <1024> (+ k_reset<T>): every block size, specialised and generic code, all outputs, every tick
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("block", [256, 512, 1024])
@pytest.mark.parametrize("generic", [False, True], ids=["specialised", "generic"])
@pytest.mark.parametrize("static", [True, False], ids=["static", "nonstatic"])
def test_lean_tick_every_block_size(block, generic, static, hip_option):
    from oracle import oracle as orc
    from reinlife_amd.worlds import DeviceWorlds
    hip_option("world_block", block)   # read by the library at every launch
    if generic:
        hip_option("world_generic", 1)
    else:
        hip_option("world_generic", 0)
    R = 24
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=static, limit_reproduction=False, incentivize_killing=True)
    dw = DeviceWorlds(n_worlds=R, seed=4711, world_base=7, **cfg)
    ow = orc.OracleWorlds(n_worlds=R, seed=4711, world_base=7, **cfg)
    dw.reset_synthetic(100); ow.reset_synthetic(100)       # k_reset<block>
    _cmp_state(dw, ow, "reset")
    _cmp_rows(dw.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "reset obs")
    rng = np.random.RandomState(block)
    refills = 0
    for t in range(60):
        acts = rng.randint(0, 8, size=(R, dw.cap)).astype(np.int8)
        n0 = ow.s["n_agents"].copy()
        ow.step(acts)
        n1 = ow.s["n_agents"].copy()
        want = {k: getattr(ow, k).copy() for k in ("reward", "done", "src1", "obs1")}
        ow.update()
        refills += ow.refill(70, 100)
        dw.set_actions(acts)
        dw.tick_refill(70, 100)
        dw.check_error_flag()
        assert np.array_equal(dw.n_acted.cpu().numpy(), n0)
        _cmp_rows(dw.reward.cpu().numpy(), want["reward"], n1, "tick %d reward" % t)
        _cmp_rows(dw.done.cpu().numpy(), want["done"], n1, "tick %d done" % t)
        _cmp_rows(dw.src1.cpu().numpy(), want["src1"], n1, "tick %d src1" % t)
        _cmp_rows(dw.obs_state_prime().cpu().numpy(), want["obs1"], n1, "tick %d obs1" % t)
        _cmp_state(dw, ow, "tick %d" % t)
        _cmp_rows(dw.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)
    assert refills > 3
    dw.refill(101, 100); ow.refill(101, 100)                # stand-alone refill of every world: k_reset<block> again
    _cmp_state(dw, ow, "refill all")


@pytest.mark.parametrize("block", [256, 512])
def test_split_step_update_every_block_size(block, hip_option):
    """The non-lean kernels (step / update with tracker outputs) at the small block sizes."""
    from hip_backend import HipBackend
    from oracle import oracle as orc
    hip_option("world_block", block)
    R = 12
    cfg = dict(width=30, height=30, max_agents=100, n_brains=3, static_families=True, limit_reproduction=True, incentivize_killing=True)
    hb = HipBackend(R, seed=99, **cfg)
    ow = orc.OracleWorlds(n_worlds=R, seed=99, **cfg)
    hb.dw.reset_synthetic(100); ow.reset_synthetic(100)
    rng = np.random.RandomState(1)
    for t in range(40):
        acts = rng.randint(0, 8, size=(R, hb.cap)).astype(np.int8)
        ow.step(acts); hb.step(acts)
        _cmp_state(hb.dw, ow, "tick %d step" % t)
        _cmp_rows(hb.obs1, ow.obs1, ow.s["n_agents"], "tick %d obs1" % t)
        assert np.array_equal(hb.trk_tick, ow.trk_tick)
        ow.update(); hb.update()
        _cmp_state(hb.dw, ow, "tick %d update" % t)
        _cmp_rows(hb.obs2, ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)


def test_many_worlds_launch_picks_the_256_thread_kernels():
    """> 768 worlds per handle: the host picks k_world<256> / k_reset<256> by itself (no environment override) -- the
    instantiation behind the 1024- and 4096-world lines of DESIGN.md 6."""
    from oracle import oracle as orc
    from reinlife_amd.worlds import DeviceWorlds
    assert "RL_WORLD_BLOCK" not in os.environ
    R = 832
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=True, limit_reproduction=False, incentivize_killing=True)
    dw = DeviceWorlds(n_worlds=R, seed=5, **cfg)
    ow = orc.OracleWorlds(n_worlds=R, seed=5, **cfg)
    dw.reset_synthetic(100); ow.reset_synthetic(100)
    rng = np.random.RandomState(2)
    for t in range(14):
        acts = rng.randint(0, 8, size=(R, dw.cap)).astype(np.int8)
        ow.step(acts); ow.update(); ow.refill(88, 100)   # a high threshold: refills from the first ticks on
        dw.set_actions(acts)
        dw.tick_refill(88, 100)
        dw.check_error_flag()
        if t % 3 == 2 or t == 13:
            _cmp_state(dw, ow, "tick %d" % t)
            _cmp_rows(dw.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)
    assert int(dw.refill_count.item()) > 50


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] as a workload: PPO + PERD3QN, static_families=False, policy-driven, refills
# ---------------------------------------------------------------------------------------------------------------------
def test_c5_workload_soak_policy_driven_nonstatic_mixed_brains():
    """16 worlds x 200 ticks of what `bench.py --workload c5` runs (one mixed-kind policy launch + fused tick + refill):
    the policy's actions are checked against the oracle's selection rule on the GPU's own outputs (PPO: inverse-CDF sample,
    PERD3QN: epsilon-greedy) and its outputs against the oracle's f32 forward; the world the actions drive must stay
    bit-identical to the oracle's -- state, both observation passes."""
    import torch
    from oracle import oracle as orc
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    R, ticks = 16, 200
    names, eps = ["PPO", "PERD3QN"], [0.0, 0.05]
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=False, limit_reproduction=False, incentivize_killing=True)
    dw = DeviceWorlds(n_worlds=R, seed=20260928, world_base=256, **cfg)
    ow = orc.OracleWorlds(n_worlds=R, seed=20260928, world_base=256, **cfg)
    wts = [_weights(n, 100 + k) for k, n in enumerate(names)]
    dw.set_brains([(_lib.KIND_BY_METHOD[n], e, pack_brain_weights(_lib.KIND_BY_METHOD[n], w)) for n, e, w in zip(names, eps, wts)])
    dw.reset_synthetic(100); ow.reset_synthetic(100)
    steps = refills = 0
    for t in range(ticks):
        check = t % 10 == 0
        dw.act(want_q=check)
        acts = dw.actions.cpu().numpy().copy()
        n = ow.s["n_agents"].copy()
        steps += int(n.sum())
        if check:
            q = dw.out_q.cpu().numpy()
            for b, name in enumerate(names):
                ws, ks = np.nonzero((np.arange(dw.cap)[None, :] < n[:, None]) & (ow.s["a_brain"] == b))
                want_q = orc.policy_forward(orc.KIND_BY_NAME[name], wts[b], ow.obs2[ws, ks])
                np.testing.assert_allclose(q[ws, ks], want_q, rtol=0, atol=1e-5, err_msg="tick %d %s" % (t, name))
                want_a = orc.select_actions(ow.cfg, orc.KIND_BY_NAME[name], q[ws, ks], ws, ks, ow.s["tick"], ow.s["epoch"], eps[b])
                assert np.array_equal(acts[ws, ks], want_a), "tick %d %s actions" % (t, name)
        dw.tick_refill(70, 100)
        ow.step(acts)
        n1 = ow.s["n_agents"].copy()
        obs1 = ow.obs1.copy() if check else None
        ow.update()
        refills += ow.refill(70, 100)
        if check or t == ticks - 1:
            torch.cuda.synchronize(); dw.check_error_flag()
            _cmp_state(dw, ow, "tick %d" % t)
            _cmp_rows(dw.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)
            if obs1 is not None:
                _cmp_rows(dw.obs_state_prime().cpu().numpy(), obs1, n1, "tick %d obs1" % t)
    assert refills > 10 and int(dw.refill_count.item()) == refills and steps > 200_000
    assert int(ow.s["max_gene"].max()) > 2   # non-static: _produce created new genes


# ---------------------------------------------------------------------------------------------------------------------
# greedy action agreement with an f32 forward, measured (not argued) over >= 1e6 free-running rows
# ---------------------------------------------------------------------------------------------------------------------
def test_greedy_actions_agree_with_an_f32_forward_on_a_million_rows(capsys):
    """256 free-running worlds x 50 ticks (~1.1e6 agent rows): the HIP kernel's greedy action vs argmax of an f32 forward of
    the same observation rows (batched sgemm of oracle/cpu_bench.py; the C oracle's scalar forward on a subset).  A flip is
    allowed only where the f32 top-2 gap is below 1e-5 -- north_star's tolerance on the Q values; the count is printed."""
    import torch
    from oracle import cpu_bench, oracle as orc
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    R, ticks = 256, 50
    names = ["PERD3QN", "DQN", "D3QN"]
    cfg = dict(width=30, height=30, max_agents=100, n_brains=3, static_families=True, limit_reproduction=False, incentivize_killing=True)
    dw = DeviceWorlds(n_worlds=R, seed=77, **cfg)
    wts = [_weights(n, 300 + k) for k, n in enumerate(names)]
    layers = [cpu_bench.unpack(n, w) for n, w in zip(names, wts)]
    dw.set_brains([(_lib.KIND_BY_METHOD[n], 0.0, pack_brain_weights(_lib.KIND_BY_METHOD[n], w)) for n, w in zip(names, wts)])
    dw.reset_synthetic(100)
    rows = flips = near = 0
    worst_gap = 0.0
    maxdq = 0.0
    for t in range(ticks):
        dw.act(want_q=True)
        torch.cuda.synchronize()
        n = dw.s["n_agents"].cpu().numpy()
        br = dw.s["a_brain"].cpu().numpy()
        obs = dw.obs_state().cpu().numpy()
        acts = dw.actions.cpu().numpy()
        q = dw.out_q.cpu().numpy()
        live = np.arange(dw.cap)[None, :] < n[:, None]
        for b, name in enumerate(names):
            ws, ks = np.nonzero(live & (br == b))
            ref = cpu_bench.forward(name, layers[b], obs[ws, ks])
            if t == 0:   # the sgemm forward is itself pinned to the oracle's scalar forward
                sub = slice(0, 4000)
                np.testing.assert_allclose(ref[sub], orc.policy_forward(orc.KIND_BY_NAME[name], wts[b], obs[ws[sub], ks[sub]]), rtol=0, atol=2e-6)
            maxdq = max(maxdq, float(np.abs(q[ws, ks] - ref).max()))
            want = ref.argmax(1)
            got = acts[ws, ks]
            srt = np.sort(ref, axis=1)
            gap = srt[:, -1] - srt[:, -2]
            bad = got != want
            rows += len(ws); flips += int(bad.sum()); near += int((gap < 1e-5).sum())
            if bad.any():
                worst_gap = max(worst_gap, float(gap[bad].max()))
        dw.tick_refill(70, 100)
    with capsys.disabled():
        print("\n[argmax agreement] %d rows, %d greedy actions differ from the f32 argmax (largest f32 top-2 gap among them %.3g); "
              "%d rows have an f32 top-2 gap < 1e-5; max |Q_hip - Q_f32| = %.3g" % (rows, flips, worst_gap, near, maxdq))
    assert rows >= 1_000_000
    assert maxdq <= 1e-5
    assert worst_gap < 1e-5, "a greedy action flipped where the f32 forward separates the top two actions by %.3g" % worst_gap


# ---------------------------------------------------------------------------------------------------------------------
# within-row dynamic range of the block-scaled 2 x f16 split
# ---------------------------------------------------------------------------------------------------------------------
def test_policy_split_precision_within_row_dynamic_range(capsys):
    """Rows whose elements span 2^0 .. 2^-20 against weight rows with the same spread, arranged adversarially (large weights
    on tiny inputs and the reverse, so that every product is ~2^-20 of max|x| max|w|).  The scheme scales each ROW by one
    power of two, so what it keeps of an element is measured against its row's maximum.  Bound it must meet, per output o of a
    layer with inputs x and weight row w_o (n terms):

        |y_hip - y_exact|  <=  2^-19 * sum_k |x_k w_ok|  +  n * 2^-31 * max|x| * max|w_o|

    (first term: the dropped lo.lo product and the two truncations toward zero, 2^-22 relative each and all of one sign, plus f32 accumulation; second: both
    operands' low parts bottom out at the f16 subnormal step, 2^-24 of a scaled row maximum in [2^10, 2^11)).  For rows without
    such a spread the second term is far below f32's own rounding; with it, it is what separates the scheme from true f32 --
    an ABSOLUTE error of ~1e-7 * max|x| max|w| on outputs that are themselves that small.  Checked on the input layer (the
    only one fed arbitrary data) through a DQN whose later layers pass relu(+y), relu(-y) through and recombine them."""
    import torch
    from reinlife_amd import _lib
    from reinlife_amd.worlds import pack_brain_weights, policy_forward
    rng = np.random.RandomState(11)
    n_rows, n_in = 512, 153
    expo = rng.randint(0, 21, size=(n_rows, n_in))
    expo[:, 0] = 0                                           # every row has a full-scale element
    sign = rng.choice([-1.0, 1.0], size=(n_rows, n_in))
    x = (rng.uniform(1.0, 2.0, size=(n_rows, n_in)) * np.exp2(-expo.astype(np.float64)) * sign).astype(np.float32)
    w = np.zeros((8, n_in), np.float64)
    wexp = rng.randint(0, 21, size=(4, n_in)); wexp[:, 0] = 0
    w[:4] = rng.uniform(1.0, 2.0, size=(4, n_in)) * np.exp2(-wexp.astype(np.float64)) * rng.choice([-1.0, 1.0], size=(4, n_in))  # independent spread
    for o, r in ((4, 1), (5, 2), (6, 3), (7, 4)):           # adversarial against row r: |w_k| = 2^-20 / 2^-expo_k, signs matched
        w[o] = np.exp2((expo[r] - 20).astype(np.float64)) * rng.uniform(1.0, 1.25, size=n_in) * sign[r]
    w = w.astype(np.float32)
    W1 = np.zeros((128, n_in), np.float32); W1[:8] = w; W1[8:16] = -w
    W2 = np.zeros((64, 128), np.float32); W2[np.arange(16), np.arange(16)] = 1.0
    W3 = np.zeros((8, 64), np.float32); W3[np.arange(8), np.arange(8)] = 1.0; W3[np.arange(8), 8 + np.arange(8)] = -1.0
    flat = np.concatenate([W1.reshape(-1), np.zeros(128, np.float32), W2.reshape(-1), np.zeros(64, np.float32), W3.reshape(-1), np.zeros(8, np.float32)])
    out = policy_forward(_lib.DQN, pack_brain_weights(_lib.DQN, flat), torch.as_tensor(x, device="cuda:0")).cpu().numpy().astype(np.float64)
    xd, wd = x.astype(np.float64), w.astype(np.float64)
    exact = xd @ wd.T
    mag = np.abs(xd) @ np.abs(wd).T
    xmax = np.abs(xd).max(1, keepdims=True)
    bound_l1 = 2.0 ** -19 * mag + n_in * 2.0 ** -31 * xmax * np.abs(wd).max(1)[None, :]
    # the two pass-through layers split the activation row again: 2^-21 of the value itself + the subnormal floor against the
    # row's largest feature
    bound = bound_l1 + 2 * (2.0 ** -21 * np.abs(exact) + 2.0 ** -31 * np.abs(exact).max(1, keepdims=True))
    err = np.abs(out - exact)
    f32 = np.abs((x @ w.T).astype(np.float64) - exact)
    adv = err[[1, 2, 3, 4], [4, 5, 6, 7]]                    # the adversarial (row, output) pairs
    with capsys.disabled():
        print("\n[within-row range] max err / bound = %.3f over %d outputs; adversarial pairs: |y| = %s, err = %s (f32 sgemm: %s); "
              "max err / sum|x w| elsewhere = %.3g"
              % (float((err / bound).max()), err.size, np.abs(exact[[1, 2, 3, 4], [4, 5, 6, 7]]).round(7).tolist(), adv.tolist(),
                 f32[[1, 2, 3, 4], [4, 5, 6, 7]].tolist(), float((err[:, :4] / mag[:, :4]).max())))
    assert (err <= bound).all(), float((err / bound).max())
    # the plain statement for rows without an adversarial partner: f32-grade relative to the magnitude of the terms
    assert float((err[:, :4] / mag[:, :4]).max()) < 3e-6


# ---------------------------------------------------------------------------------------------------------------------
# multi-GPU plumbing on one GPU: the RCCL path of bench.py at world size 1, and replica-layout independence
# ---------------------------------------------------------------------------------------------------------------------
def _bench(extra_env, *args):
    env = dict(os.environ, **extra_env)
    env.pop("RL_WORLD_BLOCK", None); env.pop("RL_WORLD_GENERIC", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--burnin", "40", "--worlds", "64",
                          "--no-cpu-baseline", "--no-kernel-timing", *args], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_rccl_path_at_world_size_one_reduces_the_same_counters():
    """RL_FORCE_DIST=1: bench.py initialises the nccl (= RCCL) process group with one rank and sends its counters through the
    same all-gather the 8-GPU run uses.  The reduced counters must equal the plain single-process run's (the worlds are
    deterministic in (seed, global replica id))."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    plain = _bench({})
    dist = _bench({"RL_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert plain["rccl_ranks"] == 1 and dist["rccl_ranks"] == 1 and dist["n_gpus"] == 1
    # the metric's reduction really ran over RCCL, once -- and never without a process group
    assert dist["rccl_collectives_executed"] == 1 and plain["rccl_collectives_executed"] == 0
    # trainer() under the process group: rank -> world_base, ONE collective per closed Tracker interval (4 in the warm-up call with
    # update_interval=10, 4 in the 2000-episode call, none in the 30-episode window), and the pooled statistics are the plain run's
    ta, tb = plain["api_trainer"], dist["api_trainer"]
    assert tb["ranks"] == 1 and tb["world_base"] == 0 and ta["tracker_rccl_collectives"] == 0
    assert tb["tracker_rccl_collectives"] == 8 and tb["tracker_intervals_closed"] == 4
    assert ta["tracker_last_interval"] == tb["tracker_last_interval"] and ta["launches"] == tb["launches"] == 4
    for key in ("agent_steps", "mean_agents_per_world", "world_refills", "worlds_total"):
        assert plain["config"][key] == dist["config"][key], key
    assert plain["config"]["agent_steps"] > 30 * 64 * 60


def test_bench_refuses_more_gpus_than_there_are():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "GPUs are visible" in (out.stderr + out.stdout)


def test_replica_trajectories_do_not_depend_on_the_layout():
    """world_base keys every Philox stream by the GLOBAL replica id: 1 x 64 worlds and 2 x 32 worlds (two handles, as two
    ranks would hold them) must leave replica r in the same state after K policy-driven ticks with refills."""
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=False, limit_reproduction=False, incentivize_killing=True)
    names, eps = ["PPO", "PERD3QN"], [0.0, 0.1]
    wts = [_weights(n, 100 + k) for k, n in enumerate(names)]

    def make(n, base):
        dw = DeviceWorlds(n_worlds=n, seed=31337, world_base=base, **cfg)
        dw.set_brains([(_lib.KIND_BY_METHOD[nm], e, pack_brain_weights(_lib.KIND_BY_METHOD[nm], w)) for nm, e, w in zip(names, eps, wts)])
        dw.reset_synthetic(100)
        return dw

    whole, halves = make(64, 1000), [make(32, 1000), make(32, 1032)]
    for _ in range(90):
        for dw in [whole] + halves:
            dw.act()
            dw.tick_refill(70, 100)
    for dw in [whole] + halves:
        dw.check_error_flag()
    assert int(whole.refill_count.item()) == sum(int(h.refill_count.item()) for h in halves) > 0
    assert int(whole.acted_total.item()) == sum(int(h.acted_total.item()) for h in halves)
    for key in whole.s:
        got = np.concatenate([h.s[key].cpu().numpy() for h in halves])
        want = whole.s[key].cpu().numpy()
        if key.startswith("a_"):
            n = whole.s["n_agents"].cpu().numpy()
            for w in range(64):
                assert np.array_equal(got[w, : n[w]], want[w, : n[w]]), (key, w)
        else:
            assert np.array_equal(got, want), key


# ---------------------------------------------------------------------------------------------------------------------
# rl_run: n ticks of policy + step + update_env (+ refill) in ONE launch, worlds resident in LDS
# ---------------------------------------------------------------------------------------------------------------------
def _run_pair(R, static, seed, block=None):
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=static, limit_reproduction=False, incentivize_killing=True)
    names, eps = ["PERD3QN", "D3QN"], [0.0, 0.15]
    wts = [_weights(n, 100 + k) for k, n in enumerate(names)]
    out = []
    for _ in range(2):
        dw = DeviceWorlds(n_worlds=R, seed=seed, world_base=3, **cfg)
        dw.set_brains([(_lib.KIND_BY_METHOD[n], e, pack_brain_weights(_lib.KIND_BY_METHOD[n], w)) for n, e, w in zip(names, eps, wts)])
        dw.reset_synthetic(100)
        out.append(dw)
    return out, wts, names, eps, cfg


def _need_run(dw, block=None):
    """The multi-tick launch must cover this handle -- except at a FORCED workgroup size of 256 / 1024 threads: k_run<256> and k_run<1024>
    are slower than what the product runs in their place and live in the tuning build only (RL_TUNE=1 python reinlife_amd/build.py;
    REINLIFE_HIP_LIB=reinlife_amd/lib/libreinlife_hip_tune.so runs these cases), DESIGN.md 5.10."""
    if dw.run_supported():
        return
    assert block in (256, 1024), "rl_run should cover this configuration"
    pytest.skip("k_run<%d> is a tuning-build instantiation (not in the product library)" % block)


def _same_device_state(a, b, tag):
    n = a.s["n_agents"].cpu().numpy()
    assert np.array_equal(n, b.s["n_agents"].cpu().numpy()), tag
    for key in a.s:
        x, y = a.s[key].cpu().numpy(), b.s[key].cpu().numpy()
        if key.startswith("a_"):
            for w in range(a.R):
                assert np.array_equal(x[w, : n[w]], y[w, : n[w]]), (tag, key, w)
        else:
            assert np.array_equal(x, y), (tag, key)
    _cmp_rows(a.obs_state().cpu().numpy(), b.obs_state().cpu().numpy(), n, tag + " obs2")


@pytest.mark.parametrize("static", [True, False], ids=["static", "nonstatic"])
@pytest.mark.parametrize("block", [None, 256, 1024], ids=["T512", "T256", "T1024"])
def test_multi_tick_launch_equals_the_two_launch_loop(static, block, hip_option):
    """rl_run(n) == n x (rl_policy_act + rl_tick_refill): world state, both observation buffers, the last tick's outputs and
    actions, the counters -- for chunks of 1, 2, 7 and 30 ticks (odd and even: the Agent.state ping-pong), with refills."""
    if block:
        hip_option("world_block", block)
    # rl_run's workgroups run the one-wave policy tile (policy_tile1); the stand-alone launch uses it too when asked to, and
    # then the two paths must agree bit for bit (with its default 4-wave tile they agree to ~1e-7 in Q, like any two float32
    # summation orders)
    (fused, loop), *_ = _run_pair(20, static, 555)
    _need_run(fused, block)
    done = 0
    for chunk in (1, 2, 7, 30, 1, 30, 30):
        fused.run(chunk, 70, 100)
        for _ in range(chunk):
            loop.act(); loop.tick_refill(70, 100)
        done += chunk
        fused.check_error_flag(); loop.check_error_flag()
        tag = "after %d ticks" % done
        _same_device_state(fused, loop, tag)
        n = fused.s["n_agents"].cpu().numpy()
        n1 = None
        assert np.array_equal(fused.n_acted.cpu().numpy(), loop.n_acted.cpu().numpy()), tag
        acted = fused.n_acted.cpu().numpy()
        _cmp_rows(fused.actions.cpu().numpy(), loop.actions.cpu().numpy(), acted, tag + " actions")
        _cmp_rows(fused.prev_state().cpu().numpy(), loop.prev_state().cpu().numpy(), acted, tag + " policy input of the last tick")
        assert int(fused.acted_total.item()) == int(loop.acted_total.item()) and int(fused.refill_count.item()) == int(loop.refill_count.item())
        # post-step outputs of the last tick: the post-step list length is not stored; compare over the rows both paths wrote
        for name in ("reward", "done", "src1"):
            x, y = getattr(fused, name).cpu().numpy(), getattr(loop, name).cpu().numpy()
            assert np.array_equal(x, y), (tag, name)
        assert np.array_equal(fused.obs_state_prime().cpu().numpy(), loop.obs_state_prime().cpu().numpy()), tag + " obs1"
        assert np.array_equal(fused.src2.cpu().numpy(), loop.src2.cpu().numpy()), tag + " src2"
    assert int(fused.refill_count.item()) > 0


def test_multi_tick_launch_tracks_the_oracle_tick_by_tick():
    """run(1) per tick against the oracle fed the chosen actions: rl_run's own policy + tick, not just its agreement with the
    other path."""
    from oracle import oracle as orc
    (fused, _), wts, names, eps, cfg = _run_pair(12, True, 808)
    ow = orc.OracleWorlds(n_worlds=12, seed=808, world_base=3, **cfg)
    ow.reset_synthetic(100)
    for t in range(40):
        n = ow.s["n_agents"].copy()
        fused.run(1, 70, 100)
        acts = fused.actions.cpu().numpy().copy()
        if t % 8 == 0:   # the actions are the oracle's selection rule applied to an f32 forward (greedy brain: where the top two are apart)
            ws, ks = np.nonzero((np.arange(fused.cap)[None, :] < n[:, None]) & (ow.s["a_brain"] == 0))
            q = orc.policy_forward(orc.PERD3QN, wts[0], ow.obs2[ws, ks])
            srt = np.sort(q, axis=1)
            clear = srt[:, -1] - srt[:, -2] > 1e-5
            assert np.array_equal(acts[ws, ks][clear], q.argmax(1)[clear])
        ow.step(acts); ow.update(); ow.refill(70, 100)
        fused.check_error_flag()
        _cmp_state(fused, ow, "tick %d" % t)
        _cmp_rows(fused.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "tick %d obs2" % t)


def test_multi_tick_launch_falls_back_when_unsupported(hip_option):
    """Mixed-kind brains at a workgroup size other than 512 threads (or capture outputs) are outside rl_run's scope: DeviceWorlds.run
    loops over the two launches instead."""
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    hip_option("world_block", 256)
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=True)
    pair = []
    for _ in range(2):
        dw = DeviceWorlds(n_worlds=6, seed=4, **cfg)
        dw.set_brains([(_lib.PPO, 0.0, pack_brain_weights(_lib.PPO, _weights("PPO", 1))), (_lib.PERD3QN, 0.0, pack_brain_weights(_lib.PERD3QN, _weights("PERD3QN", 2)))])
        dw.reset_synthetic(100)
        pair.append(dw)
    assert not pair[0].run_supported()
    pair[0].run(9, 70, 100)
    for _ in range(9):
        pair[1].act(); pair[1].tick_refill(70, 100)
    _same_device_state(pair[0], pair[1], "fallback")


@pytest.mark.parametrize("width,height,max_agents,n_new,thr,limit", [(20, 15, 60, 50, 40, False), (30, 30, 100, 100, 70, True), (12, 9, 20, 18, 12, False),
                                                                      (30, 30, 150, 140, 100, False)],
                         ids=["20x15", "30x30-limit_reproduction", "12x9", "30x30-140agents-six-tiles"])
@pytest.mark.parametrize("block", [None, 256, 1024], ids=["T512", "T256", "T1024"])
def test_multi_tick_launch_other_shapes_and_the_sequential_update(width, height, max_agents, n_new, thr, limit, block, hip_option):
    """k_run's generic (not shape-specialised) instantiations and the limit_reproduction path (update_env not overlapped with the
    observation pass), at every workgroup size, against the two-launch loop AND the oracle."""
    from oracle import oracle as orc
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    if block:
        hip_option("world_block", block)
    R = 10
    cfg = dict(width=width, height=height, max_agents=max_agents, n_brains=2, static_families=True, limit_reproduction=limit, incentivize_killing=True)
    wts = [_weights("PERD3QN", 7), _weights("D3QN", 8)]
    pair = []
    for _ in range(2):
        dw = DeviceWorlds(n_worlds=R, seed=99, **cfg)
        dw.set_brains([(_lib.PERD3QN, 0.0, pack_brain_weights(_lib.PERD3QN, wts[0])), (_lib.D3QN, 0.2, pack_brain_weights(_lib.D3QN, wts[1]))])
        dw.reset_synthetic(n_new)
        pair.append(dw)
    fused, loop = pair
    if not fused.run_supported():
        pytest.skip("slot_cap above this workgroup size" if not block or fused.cap > block else "k_run<%d> is a tuning-build instantiation" % block)
    ow = orc.OracleWorlds(n_worlds=R, seed=99, **cfg)
    ow.reset_synthetic(n_new)
    for t in range(45):
        fused.run(1, thr, n_new)
        loop.act(); loop.tick_refill(thr, n_new)
        acts = fused.actions.cpu().numpy().copy()
        ow.step(acts); ow.update(); ow.refill(thr, n_new)
        fused.check_error_flag()
        _same_device_state(fused, loop, "tick %d" % t)
        _cmp_state(fused, ow, "tick %d vs oracle" % t)
        _cmp_rows(fused.obs_state().cpu().numpy(), ow.obs2, ow.s["n_agents"], "tick %d obs2 vs oracle" % t)
    fused.run(20, thr, n_new)
    for _ in range(20):
        loop.act(); loop.tick_refill(thr, n_new)
    _same_device_state(fused, loop, "after the 20-tick launch")
    assert int(fused.refill_count.item()) == int(loop.refill_count.item())


# ---------------------------------------------------------------------------------------------------------------------
# dense policy launches: four one-wave tiles per workgroup, weights through LDS (k_policy_dense)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind_name", ["PERD3QN", "D3QN"])
def test_dense_policy_kernel_is_the_one_wave_tile_bit_for_bit(kind_name, hip_option):
    """policy_variant "dense" against "wave" and the default (the same arithmetic per row) for ragged row counts, the automatic choice above
    1,536 tiles, and the f32 oracle forward (1e-5) on a sample."""
    import torch
    from oracle import oracle as orc
    from reinlife_amd import _lib
    from reinlife_amd.worlds import pack_brain_weights, policy_forward
    kind = _lib.KIND_BY_METHOD[kind_name]
    w = _weights(kind_name, 11)
    packed = pack_brain_weights(kind, w)
    g = torch.Generator(device="cuda:0").manual_seed(5)
    for n in (1, 33, 257, 4095, 8191, 60000):
        obs = torch.randn(n + 1, 153, device="cuda:0", generator=g)[:n].contiguous()
        outs = {}
        for v in ("wave", "dense", None):
            if v is None:
                hip_option("policy_variant", "auto")
            else:
                hip_option("policy_variant", v)
            out = torch.full((n, 8), float("nan"), device="cuda:0")
            policy_forward(kind, packed, obs, out)
            torch.cuda.synchronize()
            outs[v] = out.cpu().numpy()
        assert np.array_equal(outs["wave"], outs["dense"]), n
        assert np.array_equal(outs[None], outs["wave"]), "the default launch (pair tiles, or dense from 1,536 tiles on) is the same arithmetic"
        if n >= 1536 * 32:
            assert np.array_equal(outs[None], outs["dense"]), "the dense kernel is the default from 1,536 tiles on"
        sub = slice(max(0, n - 300), n)
        want = orc.policy_forward(orc.KIND_BY_NAME[kind_name], w, obs[sub].cpu().numpy())
        np.testing.assert_allclose(outs["dense"][sub], want, rtol=0, atol=1e-5)


def test_dense_policy_kernel_through_the_row_lists_with_draws(hip_option):
    """rl_policy_act over 832 worlds of two PERD3QN brains (one of them epsilon-greedy: the Philox draw per row): the launch picks the
    dense kernel by itself; actions and Q values equal the one-wave tile's."""
    import torch
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=True)
    res = {}
    for v in ("wave", None):
        if v is None:
            hip_option("policy_variant", "auto")
        else:
            hip_option("policy_variant", v)
        dw = DeviceWorlds(n_worlds=832, seed=21, **cfg)
        dw.set_brains([(_lib.PERD3QN, eps, pack_brain_weights(_lib.PERD3QN, _weights("PERD3QN", 40 + k))) for k, eps in enumerate((0.0, 0.3))])
        dw.reset_synthetic(100)
        seq = []
        for t in range(8):
            dw.act(want_q=True)
            torch.cuda.synchronize()
            n = dw.s["n_agents"].cpu().numpy()
            seq.append((n.copy(), dw.actions.cpu().numpy().copy(), dw.out_q.cpu().numpy().copy()))
            dw.tick_refill(70, 100)
        res[v] = seq
    for (n0, a0, q0), (n1, a1, q1) in zip(res["wave"], res[None]):
        assert np.array_equal(n0, n1)
        live = np.arange(a0.shape[1])[None, :] < n0[:, None]
        assert np.array_equal(a0[live], a1[live])
        assert np.array_equal(q0[live], q1[live])


@pytest.mark.parametrize("n_brains,max_agents,n_new,thr", [(1, 100, 100, 70), (3, 100, 100, 70), (8, 100, 100, 70), (2, 200, 190, 120)],
                         ids=["1brain", "3brains", "8brains", "crowded-200"])
def test_multi_tick_launch_brain_counts_and_crowded_worlds(n_brains, max_agents, n_new, thr):
    """rl_run == the two-launch loop for 1 / 3 / 8 brains (1-8 tiles per world: two waves per tile up to four tiles, one wave per tile
    above) and for worlds of up to 200+ agents (slot capacity 448, more than 128 rows per world), half of the brains epsilon-greedy."""
    import torch
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    cfg = dict(width=30, height=30, max_agents=max_agents, n_brains=n_brains, static_families=True, limit_reproduction=False, incentivize_killing=True)
    names = ["PERD3QN" if b % 2 == 0 else "D3QN" for b in range(n_brains)]
    wts = [_weights(n, 700 + k) for k, n in enumerate(names)]
    pair = []
    for _ in range(2):
        dw = DeviceWorlds(n_worlds=12, seed=99, world_base=5, **cfg)
        dw.set_brains([(_lib.KIND_BY_METHOD[n], 0.2 * (k % 2), pack_brain_weights(_lib.KIND_BY_METHOD[n], w)) for k, (n, w) in enumerate(zip(names, wts))])
        dw.reset_synthetic(n_new)
        pair.append(dw)
    fused, loop = pair
    _need_run(fused)
    most = 0
    for chunk in (1, 6, 25, 40):
        fused.run(chunk, thr, n_new)
        for _ in range(chunk):
            loop.act(); loop.tick_refill(thr, n_new)
        fused.check_error_flag(); loop.check_error_flag()
        _same_device_state(fused, loop, "%d brains, chunk %d" % (n_brains, chunk))
        acted = fused.n_acted.cpu().numpy()
        most = max(most, int(acted.max()))
        _cmp_rows(fused.actions.cpu().numpy(), loop.actions.cpu().numpy(), acted, "actions")
        assert int(fused.acted_total.item()) == int(loop.acted_total.item())
    if max_agents >= 200:
        assert most > 128, "the crowded case is meant to exceed four 32-row tiles of one brain"
