"""GPU, round 5: bench.py's N > 1 path EXECUTED -- two torchrun ranks sharing the one GPU of the box (`--dist-backend gloo --share-gpu`:
every rank on cuda:0, the collectives hop through the host and gloo), so that `python bench.py --gpus 8` on an 8-GPU node differs from
what ran here only in backend="nccl" and the device index.  Checked: ONE JSON line, the ranks and their world_base, whole-job sums equal
to a 1-rank run holding all the worlds (the worlds are keyed by global replica id), the Tracker pooled through trainer() on every rank,
configs[4] (`c5`) on every rank and reduced, nobody waiting at the final barrier; and the single-world figures of the default line.
Two ranks on one GPU cannot show a speed-up: structure is asserted, not speed (Helpers/trainer.py:79-83 of the reference)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "30", "--warmup", "5", "--burnin", "40", "--no-cpu-baseline"]


def _bench(*args, timeout=900, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RL_WORLD_BLOCK", "RL_WORLD_GENERIC", "RL_FORCE_DIST")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *[a for a in COMMON if a not in drop], *args], env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "expected ONE JSON line, got %d" % len(lines)
    assert out.stdout.strip().splitlines()[-1] == lines[0]      # ... and it is the last thing on stdout
    return json.loads(lines[0])


def test_two_rank_dry_run_of_the_multi_gpu_bench_path():
    two = _bench("--gpus", "2", "--dist-backend", "gloo", "--share-gpu", "--worlds", "64")
    one = _bench("--gpus", "1", "--worlds", "128", "--no-single-world")
    # what ran: two ranks, stated as a gloo dry run -- never to be mistaken for RCCL evidence
    assert two["n_gpus"] == 2 and two["ranks"] == 2 and two["dist_backend"] == "gloo" and two["rccl_ranks"] == 0
    assert one["ranks"] == 1 and one["dist_backend"] is None and one["rccl_ranks"] == 1
    assert two["scaling"] == "weak" and two["config"]["worlds_per_gpu"] == 64 and two["config"]["worlds_total"] == 128
    assert "x2" in two["config"]["parallelism"]
    # the shards: global replica ids 0.. and 64.., every rank's counters in the one collective, whole job = their sum = the 1-rank job
    pr = two["per_rank"]
    assert pr["world_base"] == [0, 64] and len(pr["agent_steps"]) == len(pr["elapsed_ms"]) == 2
    assert sum(pr["agent_steps"]) == two["config"]["agent_steps"] == one["config"]["agent_steps"] > 30 * 128 * 60
    assert min(pr["agent_steps"]) > 30 * 64 * 60
    assert two["config"]["world_refills"] == one["config"]["world_refills"]
    assert two["rccl_collectives_executed"] == 1 and one["rccl_collectives_executed"] == 0      # the metric's reduction: ONE collective
    assert abs(two["value"] - two["config"]["agent_steps"] / (two["ms_per_step"] * 1e-3 * 30)) < 1e-3 * two["value"]
    assert max(pr["elapsed_ms"]) * 1e-3 <= two["ms_per_step"] * 1e-3 * 30 + 1e-6 and pr["value_min"] <= pr["value_max"]
    # the job's time = latest rank's end - earliest rank's start on the node's shared clock (round 6): never shorter than any rank's own K
    # steps, longer by the start skew; the bracket incl. the closing barrier (milliseconds under gloo) and the own-clock figure are beside it
    assert all(b >= e for b, e in zip(pr["elapsed_incl_closing_barrier_ms"], pr["elapsed_ms"]))
    assert pr["start_skew_us"][0] == 0.0 and pr["end_skew_us"][0] == 0.0 and len(pr["start_skew_us"]) == 2
    job_ms = two["ms_per_step"] * 30
    spread = max(e + s * 1e-3 for e, s in zip(pr["elapsed_ms"], pr["start_skew_us"])) - min(s * 1e-3 for s in pr["start_skew_us"])
    assert abs(job_ms - spread) < 2e-2, (job_ms, spread)          # ... reproducible from the per-rank table
    assert two["value_slowest_rank_own_clock"] >= two["value"] * (1 - 1e-9) and two["value_incl_closing_barrier"] <= two["value_slowest_rank_own_clock"]
    assert abs(one["value_slowest_rank_own_clock"] - one["value"]) < 1e-6 * one["value"]      # one rank: one interval
    # nobody is left waiting behind a leg only rank 0 runs
    assert len(pr["final_barrier_wait_s"]) == 2 and max(pr["final_barrier_wait_s"]) < 5.0
    # the dominant kernel's roofline stays in a multi-rank line (rank 0's kernel)
    assert two["roofline"] is not None and two["roofline"]["bound"] == "hbm" and 0 < two["roofline"]["frac"] < 1
    assert two["cpu_baseline"] is None and two["single_world"] is None          # N = 1 only (the contract)
    # trainer() on every rank under the process group: rank -> world_base, the Tracker pooled with one collective per closed interval,
    # and the pooled statistics are EXACTLY the 1-rank job's (bit-identical: per-world rows summed in global replica order)
    ta, tb = one["api_trainer"], two["api_trainer"]
    assert tb["ranks"] == 2 and ta["ranks"] == 1 and tb["tracker_rccl_collectives"] == 8 and ta["tracker_rccl_collectives"] == 0
    assert tb["tracker_intervals_closed"] == ta["tracker_intervals_closed"] == 4
    assert tb["tracker_last_interval"] == ta["tracker_last_interval"]
    # configs[4] on every rank, reduced like the main line
    c2, c1 = two["c5"], one["c5"]
    assert c2["ranks"] == 2 and c1["ranks"] == 1 and c2["worlds_total"] == 128
    assert c2["per_rank"]["world_base"] == [0, 64] and c1["per_rank"]["world_base"] == [0]
    assert sum(c2["per_rank"]["agent_steps"]) == c2["agent_steps"] == c1["agent_steps"] > 1000 * 128 * 60
    assert len(c2["per_rank"]["us_per_tick"]) == 2 and c2["us_per_tick"] == max(c2["per_rank"]["us_per_tick"])
    assert c2["value_min_rank"] <= c2["value_max_rank"] and c2["roofline"]["rank"] == 0 and 0 < c2["roofline"]["mfma_frac"] < 1
    assert abs(c2["value"] - c2["agent_steps"] / (c2["us_per_tick"] * 1e-6 * 1000)) < 2e-3 * c2["value"]


def test_eight_rank_dry_run_with_the_cpu_baseline_in_a_multi_rank_line():
    """What the driver's ONE 8-GPU run executes, rehearsed as eight gloo ranks on the one GPU of the box: rendezvous of
    eight, eight concurrent imports of the (current) library, 8-row tables, world_base = rank x worlds, configs[4] on every rank, the
    Tracker pooled over eight ranks, `cpu_baseline` on rank 0 AFTER every timed leg with the other ranks parked (north_star: the CPU path
    beside the 1/2/4/8 figures, in the same run), nobody waiting long at the legs' barrier."""
    line = _bench("--gpus", "8", "--dist-backend", "gloo", "--share-gpu", "--worlds", "32", "--no-single-world",
                  drop=("--no-cpu-baseline",), timeout=1500)
    assert line["n_gpus"] == 8 and line["ranks"] == 8 and line["dist_backend"] == "gloo" and line["rccl_ranks"] == 0
    pr = line["per_rank"]
    assert pr["world_base"] == [32 * r for r in range(8)] and line["config"]["worlds_total"] == 256
    assert "32 worlds/GPU" in line["config"]["workload"] and "instead of 256" in line["config"]["workload"]
    for key in ("agent_steps", "elapsed_ms", "elapsed_incl_closing_barrier_ms", "start_skew_us", "end_skew_us", "final_barrier_wait_s", "cpu_baseline_wait_s"):
        assert len(pr[key]) == 8, key
    assert sum(pr["agent_steps"]) == line["config"]["agent_steps"] > 30 * 256 * 60 and min(pr["agent_steps"]) > 30 * 32 * 60
    assert line["rccl_collectives_executed"] == 1
    # the legs' barrier: rank 0 carries the kernel probes (~1 s); nobody is parked for long.  The CPU baseline's barrier is the one that waits.
    assert max(pr["final_barrier_wait_s"]) < 30.0
    cpu = line["cpu_baseline"]
    assert cpu is not None and cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["unit"] == "agent-steps/s"
    assert max(pr["cpu_baseline_wait_s"][1:]) > 1.0 and pr["cpu_baseline_wait_s"][0] < max(pr["cpu_baseline_wait_s"][1:])
    c5, api = line["c5"], line["api_trainer"]
    assert c5["ranks"] == 8 and c5["per_rank"]["world_base"] == [32 * r for r in range(8)] and c5["worlds_total"] == 256
    assert api["ranks"] == 8 and api["tracker_intervals_closed"] == 4 and api["tracker_rccl_collectives"] == 8
    assert line["value"] <= line["value_slowest_rank_own_clock"] * (1 + 1e-9)


def test_single_world_figures_of_the_default_line():
    """BASELINE configs[1] / configs[2] and the literal drop-in defaults as throughput figures (VERDICT r04 missing #3)."""
    line = _bench("--gpus", "1", "--worlds", "64", "--no-api-trainer", "--no-c5", "--no-kernel-timing")
    sw = line["single_world"]
    assert sw["c2"]["step_only"]["value"] > 0 and sw["c2"]["host_loop"]["value"] > 0 and 60 < sw["c2"]["mean_agents"] <= 100
    assert sw["c3"]["value"] > 0 and sw["c3"]["two_launch"]["value"] > 0 and 60 < sw["c3"]["mean_agents"] <= 100
    for key in ("trainer_default", "trainer_configs0"):
        r = sw[key]
        assert r["rng"] == "reference" and r["n_worlds"] == 1 and r["episodes"] == 301 and r["value"] > 0 and r["mean_agents"] > 0
    assert "configs[0]" in sw["trainer_configs0"]["call"] and "DQN(max_epi=300)" in sw["trainer_default"]["call"]
    assert sw["reference_cpu_survey_time"]["C3 100 agents, DQN forward + step"] == 7100


# ---------------------------------------------------------------------------------------------------------------------
# the input layer's use of what an observation row IS (rl_policy_dev.h in_chunk_class) -- where it must NOT be used
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["health-1000", "crowded-beyond-2n", "plain"])
def test_known_row_structure_is_only_used_where_it_holds(case):
    """The mixed-kind multi-tick kernel skips the input layer's row-maximum pass and the exactly-zero W_hi . x_lo products when a world's rows
    are KNOWN to lie in [1, 2) with an integer health plane (run_obs_flags).  Worlds where that does not hold -- an agent whose health a
    caller set beyond +-400 (the row's health feature then exceeds 2), more than 2 x max_agents agents (n / max_agents >= 2), an agent on
    cell (0,0) (float health plane: happens by itself in a few worlds per tick) -- must take the general code: outputs (Q / probabilities),
    actions and worlds BIT-identical to the stand-alone kernels, which assume nothing, tick by tick."""
    import numpy as np
    import torch
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    from test_hip_round2 import _cmp_rows, _same_device_state, _weights
    names, eps = ["PPO", "PERD3QN", "DQN"], [0.0, 0.05, 0.1]
    crowded = case == "crowded-beyond-2n"
    cfg = dict(width=12, height=10, max_agents=4, n_brains=3) if crowded else dict(width=30, height=30, max_agents=100, n_brains=3)
    n_new, thr = (11, 3) if crowded else (100, 70)
    pair = []
    for _ in range(2):
        dw = DeviceWorlds(n_worlds=24, seed=17, static_families=False, **cfg)
        dw.set_brains([(_lib.KIND_BY_METHOD[n], e, pack_brain_weights(_lib.KIND_BY_METHOD[n], _weights(n, 40 + k))) for k, (n, e) in enumerate(zip(names, eps))])
        dw.reset_synthetic(n_new)
        if case == "health-1000":   # a state only a caller can write: the row scale of these worlds' rows is not 2^10
            dw.s["a_health"][::3, 0] = 1000
            dw.s["a_health"][1::3, 1] = -900
            dw.s["a_flags"][1::3, 1] = 0
            _lib.check(dw.lib.rl_bind_state(dw.handle, __import__("ctypes").byref(dw._state)), "rl_bind_state")
            dw.observe()
        pair.append(dw)
    fused, loop = pair
    assert fused.run_supported()
    float_rows = big_rows = 0
    for t in range(25):
        obs = fused.obs_state().cpu().numpy()
        n = fused.s["n_agents"].cpu().numpy()
        live = np.arange(fused.cap)[None, :] < n[:, None]
        big_rows += int((np.abs(obs).max(axis=2)[live] >= 2.0).sum())
        float_rows += int(((obs[:, :, 49:98] != np.trunc(obs[:, :, 49:98])).any(axis=2) & live).sum())
        fused.run(1, thr, n_new, want_q=True)
        loop.act(want_q=True); loop.tick_refill(thr, n_new)
        fused.check_error_flag(); loop.check_error_flag()
        acted = fused.n_acted.cpu().numpy()
        _cmp_rows(fused.out_q.cpu().numpy(), loop.out_q.cpu().numpy(), acted, "tick %d outputs" % t)
        _cmp_rows(fused.actions.cpu().numpy(), loop.actions.cpu().numpy(), acted, "tick %d actions" % t)
        _same_device_state(fused, loop, "tick %d" % t)
    assert float_rows > 0                                  # some worlds had an agent on cell (0,0): the float health plane was exercised
    assert (big_rows > 0) == (case != "plain"), big_rows   # rows whose largest magnitude is >= 2 exist exactly in the two adversarial cases
