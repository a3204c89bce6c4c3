"""GPU, round 6: what the round changed on the device side, held to the oracle / to the other path.
  * `run_always` beyond 768 worlds takes the 512-thread multi-tick kernel (several workgroups per CU in turn) instead of silently the two-launch
    loop -- and gives the same bits as that loop;
  * the seam between two ticks of one launch is open (recycle_world's closing barrier only where store_world or a drain follows):
    launches cut at every length, the launch's last tick, refills and the TRAIN instantiation against tick-by-tick launches;
  * the forced 256- / 1024-thread instantiations the product refuses run in the TUNING library, in a process of its own (18
    parity cases had left the driver's suite when they moved there).
Reference loop being fused: Helpers/trainer.py:85-99."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_always_beyond_768_worlds_runs_the_multi_tick_kernel_and_equals_the_two_launch_loop(hip_option):
    import torch
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    from test_hip_round2 import _cmp_rows, _same_device_state, _weights
    names = ["PERD3QN", "D3QN"]
    cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=True)

    def make():
        dw = DeviceWorlds(n_worlds=1024, seed=31, **cfg)
        dw.set_brains([(_lib.KIND_BY_METHOD[n], 0.02 * k, pack_brain_weights(_lib.KIND_BY_METHOD[n], _weights(n, 70 + k))) for k, n in enumerate(names)])
        dw.reset_synthetic(100)
        return dw
    loop = make()
    assert not loop.run_supported() and "tuning-build" in _lib.lib().rl_last_error().decode()      # the product holds no k_run<256>
    hip_option("run_always", 1)
    fused = make()
    hip_option("run_always", None)
    assert fused.run_supported() and fused.lib.rl_get_option(fused.handle, b"run_always") == 1
    before = fused.launches
    for n in (17, 1, 22):
        fused.run(n, 70, 100)
        for _ in range(n):
            loop.act(); loop.tick_refill(70, 100)
    torch.cuda.synchronize()
    fused.check_error_flag(); loop.check_error_flag()
    assert fused.launches - before == 3                                       # three launches, not 2 x 40
    _same_device_state(fused, loop, "1,024 worlds, 40 ticks")
    assert int(fused.acted_total.item()) == int(loop.acted_total.item()) > 40 * 1024 * 60
    _cmp_rows(fused.actions.cpu().numpy(), loop.actions.cpu().numpy(), loop.n_acted.cpu().numpy(), "actions")
    assert np.array_equal(fused.obs_state().cpu().numpy(), loop.obs_state().cpu().numpy())


@pytest.mark.parametrize("kinds,static,train", [(["PERD3QN", "PERD3QN"], True, False), (["PPO", "PERD3QN"], False, False), (["DQN", "PPO", "D3QN"], True, True)])
def test_open_seam_launches_of_every_length_equal_tick_by_tick_launches(kinds, static, train):
    """A launch of n ticks leaves the barrier behind recycle_world out between its ticks and keeps it behind the last one: cut the same 61 ticks
    into launches of 1 (never open), 2, 3, 5, 7, 11, 13, 19 ticks -- state, both Agent.state buffers, state_prime, actions and the Tracker sums
    after every launch equal those of one-tick launches; worlds start at 100 agents with a refill threshold, so refills fall inside launches."""
    import torch
    import bench
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    from test_hip_round2 import _cmp_rows, _same_device_state
    pair = []
    for _ in range(2):
        dw = DeviceWorlds(n_worlds=40, seed=5, width=30, height=30, max_agents=100, n_brains=len(kinds), static_families=static)
        dw.set_brains([(_lib.KIND_BY_METHOD[n], 0.04 * k, pack_brain_weights(_lib.KIND_BY_METHOD[n], bench.brain_weights(n, 100 + k))) for k, n in enumerate(kinds)])
        if train:
            dw.enable_tracking(True)
        dw.reset_synthetic(100)
        pair.append(dw)
    a, b = pair
    assert a.run_supported()
    rng = np.random.RandomState(2)
    done = 0
    for n in (1, 2, 3, 5, 7, 11, 13, 19):
        sched = rng.uniform(0, 0.3, size=(n, len(kinds))).astype(np.float32) if train else None
        a.run(n, 95, 100, eps_schedule=sched)
        for t in range(n):
            b.run(1, 95, 100, eps_schedule=None if sched is None else sched[t:t + 1])
        done += n
        torch.cuda.synchronize(); a.check_error_flag(); b.check_error_flag()
        _same_device_state(a, b, "after %d ticks (launch of %d)" % (done, n))
        _cmp_rows(a.actions.cpu().numpy(), b.actions.cpu().numpy(), b.n_acted.cpu().numpy(), "actions")
        assert np.array_equal(a.obs_state().cpu().numpy(), b.obs_state().cpu().numpy()) and np.array_equal(a.prev_state().cpu().numpy(), b.prev_state().cpu().numpy())
        assert np.array_equal(a.obs_state_prime().cpu().numpy(), b.obs_state_prime().cpu().numpy())
        if train:
            assert np.array_equal(a.trk_sum.cpu().numpy(), b.trk_sum.cpu().numpy()) and np.array_equal(a.trk_cnt.cpu().numpy(), b.trk_cnt.cpu().numpy())
    assert int(a.refill_count.item()) == int(b.refill_count.item()) > 0 and int(a.acted_total.item()) == int(b.acted_total.item())


def test_the_tuning_library_runs_the_forced_workgroup_sizes_the_product_refuses():
    """k_run<256> / k_run<1024> live in lib/libreinlife_hip_tune.so only; under the product their parity cases skip.  Here they RUN -- the same test
    functions, in a pytest of their own that loads the tuning library (REINLIFE_HIP_LIB) -- and none of them may skip for that reason."""
    from reinlife_amd import build
    tune = build.TUNE_LIB_PATH
    if not os.path.exists(tune):
        pytest.skip("the tuning library is not built (RL_TUNE=1 python reinlife_amd/build.py; __graft_entry__.build() builds it)")
    env = dict(os.environ, REINLIFE_HIP_LIB=tune)
    for k in ("RL_WORLD_BLOCK", "RL_TUNE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_round2.py"), os.path.join(ROOT, "tests", "test_hip_round3.py"),
                          os.path.join(ROOT, "tests", "test_hip_round4.py"), "-m", "gpu", "-q", "-rs", "-k", "256 or 1024 or sixteen", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = out.stdout[-3000:]
    assert out.returncode == 0, (tail, out.stderr[-2000:])
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 18, tail
    # nothing skipped for want of the instantiations (_need_run's message); the two known skips are cases whose world does not fit the forced
    # workgroup at all (slot_cap above it / no room for the four-wave tiles' exchange slices: tests/test_hip_round2.py:518)
    assert "not in the product library" not in out.stdout, tail
    m = re.search(r"(\d+) skipped", tail)
    assert m is None or int(m.group(1)) <= 2, tail


def test_mfma_16x16x32_chains_give_the_bits_of_32x32x16_chains(tmp_path):
    """The hardware fact DESIGN.md 5.12 / section 9 rest on: a chain of v_mfma_f32_16x16x32_f16 over the same k slots gives
    the same f32 bits as the chain of v_mfma_f32_32x32x16_f16 every policy tile runs -- tools/ubench/mfma_shapes.hip, 400 trials x 1,024 dot
    products over 96 k from a non-zero accumulator -- at 18 instead of 32 matrix-pipe counts per instruction."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "mfma_shapes")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "tools", "ubench", "mfma_shapes.hip")], check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, check=True).stdout
    m = re.search(r"results_differing_in_bits (\d+) of (\d+)", out)
    assert m and int(m.group(1)) == 0 and int(m.group(2)) == 409600, out
    p = re.search(r"pipe_counts_per_mfma 32x32x16 ([0-9.]+)\s+16x16x32 ([0-9.]+)", out)
    assert p and 30.0 < float(p.group(1)) < 34.0 and 16.0 < float(p.group(2)) < 20.0, out
