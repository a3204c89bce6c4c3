#!/usr/bin/env python
"""bench.py -- ReinLife hot-path throughput on MI355X: agent-steps/s (policy forward + env.step + update_env).

    python bench.py --gpus N --steps K --warmup W
    N > 1: one process per GPU over RCCL.  Under torchrun (WORLD_SIZE set: python -m torch.distributed.run --nnodes=1
    --nproc-per-node N ... bench.py --gpus N ...) this process is one rank; started plainly with --gpus N > 1 it re-executes
    itself under torch.distributed.run with N ranks.  The JSON carries `rccl_ranks` = the world size RCCL saw; a mismatch with
    --gpus is an error, never a silent single-rank run.
    Dry run of the N > 1 path on ONE GPU (tests/test_hip_round5.py): --dist-backend gloo --share-gpu puts every rank on cuda:0 and
    sends the (host-hopped) collectives through gloo; the line then says "dist_backend": "gloo" and "rccl_ranks": 0 -- it exercises
    every line of the multi-rank code except RCCL itself and the device index, and is no evidence about scaling.

One "step" = one trainer-loop tick (Helpers/trainer.py:85-99 minus learn) of EVERY world on the GPU:
    policy forward + action selection for all agents  ->  step + update_env  ->  re-generation of worlds below 70 agents
executed either by ONE multi-tick launch (rl_run: every world stays in its workgroup's LDS between ticks; --path fused, the
default where it is faster: dueling brains, <= 768 worlds per GPU) or by two launches per tick (rl_policy_act + rl_tick_refill;
--path two-launch).  Both write every per-tick output of the reference's loop every tick; same Philox streams, same worlds.
Workload (BASELINE.json configs[3]): 256 independent 30x30 worlds per GPU x 100 agents, two
PERD3QN brains (random-init weights of the reference architecture, greedy), static families, synthetic worlds from
the Philox generator, a world is re-generated when its population drops below 70.  Worlds are sharded over GPUs
with NO data-path collective (weak scaling); RCCL is used once, for the final counter reduction.
An agent-step = one live agent receiving an action and being advanced by one step().

Before the warm-up the worlds are brought to their steady regime in untimed set-up (`--burnin`, 2000 ticks: every world
starts with one cohort of 100 agents, so without it the first refills come in synchronized waves; and the chip needs tens of
milliseconds of sustained load to reach its working clocks -- behind 300 ticks = 7 ms of load a 20-step window reads 7.6e8,
behind 600 ... 5000 ticks 8.0e8, `tools/window_probe.py`), so any --steps window measures the state a
training loop runs in.

Prints ONE JSON line (rank 0).  `value` counts the full tick (update_env included: more work than the metric's
literal "env.step + policy fwd", never less); `variant_policy_plus_step` is the literal variant (i) of BASELINE.md 3
(get_action + step, update_env untimed), measured with HIP events.  `cpu_baseline` = oracle/cpu_bench.py.
"""
import argparse
import json
import os
import sys
import subprocess
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from reinlife_amd import _lib  # noqa: E402
from reinlife_amd import distributed as rl_dist  # noqa: E402
from reinlife_amd.distributed import reduce_counters  # noqa: E402
from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights  # noqa: E402

# Algorithmic HBM bytes per agent-step (DESIGN.md "Roofline accounting"; SURVEY.md 8d gives 2,082 B for the whole
# tick of the reference's data flow, of which the policy's 612 B observation read belongs to the policy kernel):
#   tick kernel : state_prime row 612 + state row 612 + agent record read 36 + write 36 + action 1 + reward 4 + done 1
#                 + src 2+2 + type grid (900 B read + 900 B write) / 100 agents = 18   -> 1,324 B
#   policy      : observation row 612 + action 1 + brain id 4                          -> 617 B, 107,008 FLOP (PERD3QN)
TICK_BYTES_PER_AGENT_STEP = 612 + 612 + 36 + 36 + 1 + 4 + 1 + 4 + 18
POLICY_BYTES_PER_AGENT = 612 + 1 + 4
POLICY_FLOP_PER_AGENT = {"DQN": 56576, "D3QN": 107008, "PERD3QN": 107008, "PPO": 213504}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2516.6   # MI355X_MICROARCH.md: ~2.5 PF dense f16 / bf16 (16 x the 157.3 TF f32 matrix rate)
# The policy kernel gets f32-grade results from the f16 pipe: every f32 product is three f16 partial products
# (2 x f16 block-scaled split), so the ceiling for ALGORITHMIC (f32-equivalent) FLOP/s is the f16 dense peak / 3.
MFMA_SPLIT_PRODUCTS = 3
MFMA_F32_EQUIV_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / MFMA_SPLIT_PRODUCTS


def brain_weights(kind_name, seed):
    """Random-init weights of the reference architecture, nn.Linear default scale (uniform +-1/sqrt(fan_in))."""
    shapes = {"DQN": [(128, 153), (64, 128), (8, 64)],
              "D3QN": [(128, 153), (128, 128), (8, 128), (128, 128), (1, 128)],
              "PPO": [(256, 153), (256, 256), (8, 256), (1, 256)]}["D3QN" if kind_name == "PERD3QN" else kind_name]
    rng = np.random.RandomState(seed)
    parts = []
    for n_out, n_in in shapes:
        b = 1.0 / np.sqrt(n_in)
        parts.append(rng.uniform(-b, b, size=(n_out, n_in)).astype(np.float32).reshape(-1))
        parts.append(rng.uniform(-b, b, size=(n_out,)).astype(np.float32))
    return np.concatenate(parts)


WORKLOADS = {
    "c4": dict(brains=["PERD3QN", "PERD3QN"], static_families=True,
               name="%d worlds/GPU x 30x30 x 100 agents, 2xPERD3QN greedy inference, static families (BASELINE configs[3]%s)"),
    "c5": dict(brains=["PPO", "PERD3QN"], static_families=False,
               name="%d worlds/GPU x 30x30 x 100 agents, PPO + PERD3QN mixed brains, static_families=False (BASELINE configs[4]%s)"),
}


def workload_name(key, worlds):
    """config.workload: the BASELINE configuration at the replica count that actually ran (256 per GPU is the configuration's own)."""
    return WORKLOADS[key]["name"] % (worlds, "" if worlds == 256 else " at %d instead of 256 worlds per GPU" % worlds)


def make_worlds(args, rank, device, n_worlds=None, world_base=None):
    wl = WORKLOADS[args.workload]
    dw = DeviceWorlds(n_worlds=n_worlds or args.worlds, width=30, height=30, max_agents=100, n_brains=len(wl["brains"]),
                      static_families=wl["static_families"], seed=args.seed, device=device,
                      world_base=rank * args.worlds if world_base is None else world_base)
    dw.set_brains([(_lib.KIND_BY_METHOD[n], 0.0, pack_brain_weights(_lib.KIND_BY_METHOD[n], brain_weights(n, 100 + k), device))
                   for k, n in enumerate(wl["brains"])])
    dw.reset_synthetic(100)
    return dw


def one_step(dw):
    dw.act()                    # k_bucket + k_policy: Agent.get_action for every agent of every world
    dw.tick_refill(70, 100)     # k_world<TICK>: step + update_env (+ re-generation of worlds below 70 agents)


FUSED_CHUNK = 2000  # ticks per rl_run launch (a launch of 2000 ticks lasts ~50 ms; each launch costs ~40 us of start-up)


def run_ticks(dw, n, fused, events=None):
    """n trainer-loop ticks of every world of `dw`.  events: a list that receives (start, end, ticks) HIP-event pairs recorded on the
    launch stream around every multi-tick launch (the kernel's duration over the timed region)."""
    if fused:
        while n > 0:
            k = min(n, FUSED_CHUNK)
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            dw.run(k, 70, 100)      # k_run: policy + step + update_env + refill, k ticks in one launch
            if events is not None:
                e1.record()
                events.append((e0, e1, k))
            n -= k
    else:
        for _ in range(n):
            one_step(dw)


class StreamGroups:
    """--groups G > 1: the GPU's worlds as G independent DeviceWorlds, each on its own HIP stream, so that one group's
    policy launch can overlap another group's tick (both kernels are latency-bound at 256 worlds and leave issue slots
    and HBM idle).  Worlds keep their global ids (world_base), so results do not depend on G."""

    def __init__(self, args, rank, device):
        G = args.groups
        if args.worlds % G:
            raise SystemExit("bench.py: --worlds must be a multiple of --groups")
        per = args.worlds // G
        self.streams = [torch.cuda.Stream(device) for _ in range(G)]
        self.parts = []
        for g, st in enumerate(self.streams):
            with torch.cuda.stream(st):
                self.parts.append(make_worlds(args, rank, device, per, rank * args.worlds + g * per))
        torch.cuda.synchronize()

    def step(self):
        for dw, st in zip(self.parts, self.streams):
            with torch.cuda.stream(st):
                one_step(dw)

    def counters(self):
        return (sum(float(dw.acted_total.item()) for dw in self.parts), sum(float(dw.refill_count.item()) for dw in self.parts))

    def zero_counters(self):
        for dw, st in zip(self.parts, self.streams):
            with torch.cuda.stream(st):
                dw.acted_total.zero_()
                dw.refill_count.zero_()

    def check_error_flag(self):
        for dw in self.parts:
            dw.check_error_flag()


def cpu_baseline(args):
    """The CPU port of the same path on the host cores (oracle/cpu_bench.py): one process per core, C world tick + batched
    sgemm policy; world_only / policy_only / full_tick, single-thread and all-core, ~15 s in all."""
    from oracle import cpu_bench
    wl = WORKLOADS[args.workload]
    return cpu_bench.run(wl["brains"], wl["static_families"], args.seed, [brain_weights(n, 100 + k) for k, n in enumerate(wl["brains"])],
                         cores=max(1, min(os.cpu_count() or 1, 64)))


def api_trainer(args, device, dist=None, rank=0):
    """The same loop through the reference's API surface (Helpers/trainer.py:7-107): trainer(brains, n_episodes=K, n_worlds=...,
    save=False, print_results=False) with the reference's default training=True -- the brains' epsilon decays per episode and the
    Tracker closes an interval every 500 episodes -- on the benchmark's synthetic worlds (100 agents per world, re-generated below
    70: the keyword-only extras synthetic_agents / refill_below).  Wall-clocked around the loop inside trainer() (env.loop_seconds:
    construction of the Environment and the reset launch are set-up, like the untimed set-up of the main line); agent-steps from the
    device counter.  With several ranks (--gpus N) EVERY rank makes the same call: Environment takes rank -> world_base = rank *
    n_worlds and cuda:LOCAL_RANK from the process group and the Tracker pools every closed interval over all ranks with one RCCL
    collective (Helpers/tracker.py); the figures below are then whole-job sums over the slowest rank's time."""
    import warnings
    from reinlife_amd import Models
    from reinlife_amd.Helpers.trainer import trainer
    wl = WORKLOADS[args.workload]
    cls = {"DQN": Models.DQN, "D3QN": Models.D3QN, "PERD3QN": Models.PERD3QN, "PPO": Models.PPO}

    def brains():
        out = []
        for k, n in enumerate(wl["brains"]):
            b = cls[n](max_epi=10_000) if n == "DQN" else cls[n]()
            flat, o = torch.from_numpy(brain_weights(n, 100 + k)), 0
            sd = b._net().state_dict()
            for key, v in sd.items():   # the same random-init weights as the main line, through load_state_dict
                sd[key] = flat[o:o + v.numel()].reshape(v.shape).clone(); o += v.numel()
            b._net().load_state_dict(sd)
            out.append(b)
        return out

    tracker_collectives = [0]

    def one(k, **kw):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            env = trainer(brains(), n_episodes=k, n_worlds=args.worlds, save=False, print_results=False, static_families=wl["static_families"],
                          device=device, seed=args.seed, synthetic_agents=100, refill_below=70, dist=dist, **kw)
        steps = int(env.worlds.acted_total.item())
        env.worlds.check_error_flag()
        tracker_collectives[0] += env.tracker.collectives_executed
        if dist is not None:   # whole job: the ranks' agent-steps over the slowest rank's loop time (one more small all-gather)
            tot, tmax, _ = reduce_counters(torch.tensor([float(steps)], dtype=torch.float64, device=device), env.loop_seconds, dist)
            return int(tot[0].item()), tmax, env
        return steps, env.loop_seconds, env
    one(40, update_interval=10)                # warm-up: the TRAIN kernel's first launch, the Tracker's torch reductions (first use loads their code objects: ~15 ms)
    k_long = max(2000, args.steps)
    s_long, t_long, env = one(k_long)
    s_win, t_win, env_win = one(args.steps)    # the same window as the main line's --steps
    return {"value": round(s_long / t_long, 1), "unit": "agent-steps/s", "episodes": k_long + 1, "us_per_tick": round(t_long / (k_long + 1) * 1e6, 2),
            "launches": env.worlds.launches, "tracker_intervals_closed": len(env.tracker.results["Avg Number of Populations"]),
            "final_epsilon": [round(float(getattr(b, "epsilon", 0.0)), 4) for b in env.brains],
            "tracker_last_interval": {"Avg Population Size": [v[-1] for v in env.tracker.results["Avg Population Size"].values()],
                                      "Avg Number of Populations": env.tracker.results["Avg Number of Populations"][-1]},
            "value_at_steps": round(s_win / t_win, 1), "steps_window": args.steps + 1, "window_ms": round(t_win * 1e3, 4),
            "ranks": 1 if dist is None else dist.get_world_size(), "world_base": env.world_base,
            "tracker_rccl_collectives": tracker_collectives[0] if dist is not None else 0,
            "call": "trainer(brains, n_episodes=K, n_worlds=%d, save=False, print_results=False, synthetic_agents=100, refill_below=70) "
                    "[training=True, update_interval=500: the reference's defaults]%s" % (args.worlds, "" if dist is None else " on every rank, torch.distributed initialised"),
            "timed": "the loop inside trainer() (env.loop_seconds), device idle before and after; agent-steps from the device counter"}


# the reference's own CPU path, measured at survey time by importing it (Python 3.10, 1 thread of 8 host cores of the
# survey container; it cannot travel to the GPU box): agent-steps/s.  Context for `single_world`, not measured in this run.
REFERENCE_SURVEY_FIGURES = {"C1 trainer([PERD3QN(),PERD3QN()]) natural population, full loop incl. learn": 1300, "C1 get_action + step only": 2600,
                            "C2 100 agents, random actions, env.step only": 10400, "C2 step + update_env": 5900,
                            "C3 100 agents, DQN forward + step": 7100, "C3 with update_env": 4700,
                            "source": "BASELINE.md section 2 (survey-time, reference imported read-only, 1 core)"}


def single_world(args, device):
    """BASELINE configs[1] / configs[2] and the literal drop-in default as throughput figures (parity for them lives in tests/): ONE 30x30
    world on one MI355X -- one workgroup on one of 256 CUs, so these are latency figures of a single world, not a use of the chip.
      c2: 100 agents (re-generated below 70), uniform random actions, Environment.step only (rl_step between HIP events; update_env + refill
          run untimed between the pairs) -- SURVEY.md 8d C2;
      c3: the same world, two greedy DQN brains (153 -> 128 -> 64 -> 8), policy forward + step + update_env in the multi-tick launch (rl_run) -- C3;
      trainer_default: trainer([DQN(max_epi=300), DQN(max_epi=300)], n_episodes=300, save=False, print_results=False) with NO extra
          keyword -- n_worlds=1 -> rng="reference": the host makes the reference's draws from `random` / np.random tick by tick
          (Helpers/trainer.py, reference loop trainer.py:85-99), natural population (two families, one agent each at reset);
      trainer_configs0: BASELINE configs[0]'s literal call, trainer([PERD3QN(), PERD3QN()], width=30, height=30, max_agents=100,
          static_families=True, n_episodes=300, save=False, print_results=False).
    ~0.1 s of GPU time + the two host-driven loops (~1 s)."""
    import random
    import warnings
    from reinlife_amd import Models
    from reinlife_amd.Helpers.trainer import trainer
    out = {"reference_cpu_survey_time": REFERENCE_SURVEY_FIGURES}
    one = argparse.Namespace(**dict(vars(args), worlds=1))

    # ---- c2: random actions, step only --------------------------------------------------------------------------------------
    dw = DeviceWorlds(n_worlds=1, width=30, height=30, max_agents=100, n_brains=2, static_families=True, seed=args.seed, device=device)
    dw.reset_synthetic(100)
    n = 400
    acts = torch.randint(0, 8, (n + 20, 1, dw.cap), dtype=torch.int8, device=device, generator=torch.Generator(device).manual_seed(args.seed))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(n)]
    acted, t0 = [], None
    for t in range(n + 20):
        if t == 20:
            torch.cuda.synchronize()
            dw.acted_total.zero_()
            t0 = time.perf_counter()
        dw.actions.copy_(acts[t])
        if t >= 20:
            ev[t - 20][0].record()
        dw.step()
        if t >= 20:
            ev[t - 20][1].record()
        dw.update(); dw.refill(70, 100)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dw.check_error_flag()
    steps = int(dw.acted_total.item())
    t_step = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e-3
    out["c2"] = {"workload": "1 world x 30x30 x 100 agents (refill below 70), uniform random actions (BASELINE configs[1])",
                 "step_only": {"value": round(steps / n / t_step, 1), "unit": "agent-steps/s", "us_per_step": round(t_step * 1e6, 2),
                               "timed": "rl_step (Environment.step) between HIP events, median of %d; update_env + refill untimed between the pairs" % n},
                 "host_loop": {"value": round(steps / wall, 1), "unit": "agent-steps/s", "us_per_tick": round(wall / n * 1e6, 2),
                               "timed": "wall clock over %d ticks of copy-actions + rl_step + rl_update + rl_refill launched from Python (4 launches per tick, no synchronise inside)" % n},
                 "mean_agents": round(steps / n, 1)}
    del dw

    # ---- c3: two greedy DQN brains, the multi-tick launch ----------------------------------------------------------------------
    dw = DeviceWorlds(n_worlds=1, width=30, height=30, max_agents=100, n_brains=2, static_families=True, seed=args.seed, device=device)
    dw.set_brains([(_lib.DQN, 0.0, pack_brain_weights(_lib.DQN, brain_weights("DQN", 100 + k), device)) for k in range(2)])
    dw.reset_synthetic(100)
    n = 2000
    res = {"workload": "1 world x 30x30 x 100 agents (refill below 70), 2 x DQN 153->128->64->8 greedy, policy forward + step + update_env (BASELINE configs[2])"}
    if dw.run_supported():
        dw.run(n, 70, 100)
        torch.cuda.synchronize()
        dw.acted_total.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(); dw.run(n, 70, 100); e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        steps = int(dw.acted_total.item())
        res.update({"value": round(steps / wall, 1), "unit": "agent-steps/s", "us_per_tick": round(wall / n * 1e6, 2),
                    "kernel_us_per_tick": round(e0.elapsed_time(e1) * 1e-3 / n * 1e6, 2), "mean_agents": round(steps / n, 1),
                    "timed": "wall clock around ONE rl_run launch of %d ticks + synchronise (one workgroup, the world resident in its LDS)" % n})
    dw.acted_total.zero_()
    m = 300
    for _ in range(20):
        one_step(dw)
    torch.cuda.synchronize()
    dw.acted_total.zero_()
    t0 = time.perf_counter()
    for _ in range(m):
        one_step(dw)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dw.check_error_flag()
    steps = int(dw.acted_total.item())
    res["two_launch"] = {"value": round(steps / wall, 1), "unit": "agent-steps/s", "us_per_tick": round(wall / m * 1e6, 2),
                         "timed": "wall clock over %d ticks of rl_policy_act + rl_tick_refill launched from Python (what Environment.act/step/update_env cost per tick without the fused loop)" % m}
    out["c3"] = res
    del dw

    # ---- the drop-in defaults: trainer(brains, ...) exactly as a user of the reference calls it -----------------------------------------
    def call(make, n_epi, **kw):
        random.seed(args.seed); np.random.seed(args.seed % (2 ** 32)); torch.manual_seed(args.seed)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            env = trainer(make(), n_episodes=n_epi, save=False, print_results=False, **kw)
        steps = int(env.worlds.acted_total.item())
        return {"value": round(steps / env.loop_seconds, 1), "unit": "agent-steps/s", "episodes": n_epi + 1,
                "us_per_tick": round(env.loop_seconds / (n_epi + 1) * 1e6, 1), "mean_agents": round(steps / (n_epi + 1), 2),
                "rng": env.rng, "n_worlds": env.n_worlds, "device": str(env.device),
                "timed": "the loop inside trainer() (env.loop_seconds); agent-steps from the device counter"}
    dqn = lambda: [Models.DQN(max_epi=300), Models.DQN(max_epi=300)]  # noqa: E731
    call(dqn, 30)   # warm-up: first launches of the tick-by-tick kernels, the brains' first forward
    r = call(dqn, 300)
    r["call"] = "trainer([DQN(max_epi=300), DQN(max_epi=300)], n_episodes=300, save=False, print_results=False)"
    out["trainer_default"] = r
    r = call(lambda: [Models.PERD3QN(), Models.PERD3QN()], 300, width=30, height=30, max_agents=100, static_families=True)
    r["call"] = "trainer([PERD3QN(), PERD3QN()], width=30, height=30, max_agents=100, static_families=True, n_episodes=300, save=False, print_results=False)  [BASELINE configs[0]]"
    out["trainer_configs0"] = r
    return out


def c5_leg(args, rank, device, dist=None):
    """BASELINE configs[4] next to the main line -- on EVERY rank: PPO + PERD3QN mixed brains, static_families=False (the mixed-kind
    multi-tick kernel, k_run<512, fixed, kKindAll>; PPO.py:101-106,164-169, PERD3QN.py:198-210, environment.py:521-547,728-739), the
    same synthetic worlds and refill rule, 256 worlds per GPU at global replica ids rank * 256 ... (2048 replicas on 8
    GPUs).  ~60 ms of GPU time per rank: 1000 untimed ticks, then one 1000-tick launch between HIP events and a wall clock, the ranks
    released together by a barrier.  With a process group the figures are whole-job -- every rank's agent-steps over the SLOWEST rank's
    wall time, reduced by the same one-collective row gather as the main line -- with the per-rank table beside them; `roofline` is
    rank 0's kernel (HIP events on its launch stream)."""
    import argparse
    a5 = argparse.Namespace(**dict(vars(args), workload="c5"))
    dw = make_worlds(a5, rank, device)
    if not dw.run_supported():
        return {"error": "rl_run does not cover this configuration"}
    n = 1000
    dw.run(n, 70, 100)
    torch.cuda.synchronize()
    dw.acted_total.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    e0.record(); dw.run(n, 70, 100); e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dw.check_error_flag()
    steps = int(dw.acted_total.item())
    t_k = e0.elapsed_time(e1) * 1e-3
    # whole job: [agent-steps, kernel seconds, world_base] of every rank + the slowest rank's wall time, ONE collective
    row = torch.tensor([float(steps), t_k, float(dw.world_base)], dtype=torch.float64, device=device)
    tot, wall_max, table = reduce_counters(row, wall, dist)
    table = table.numpy()
    steps_all = float(tot[0].item())
    wl = WORKLOADS["c5"]
    flop = float(np.mean([POLICY_FLOP_PER_AGENT[k] for k in wl["brains"]]))
    by = TICK_BYTES_PER_AGENT_STEP + POLICY_BYTES_PER_AGENT
    n_ranks = len(table)
    return {"workload": workload_name("c5", args.worlds), "value": round(steps_all / wall_max, 1), "unit": "agent-steps/s", "ranks": n_ranks,
            "worlds_total": args.worlds * n_ranks, "agent_steps": int(round(steps_all)), "ticks": n,
            "us_per_tick": round(wall_max / n * 1e6, 2), "kernel_us_per_tick": round(float(table[:, 1].max()) / n * 1e6, 2),
            "agent_steps_per_tick": round(steps_all / n, 1),
            "per_rank": {"world_base": [int(x) for x in table[:, 2]], "agent_steps": [int(x) for x in table[:, 0]],
                         "value": [round(float(a / w), 1) for a, w in zip(table[:, 0], table[:, 3])],
                         "us_per_tick": [round(float(w) / n * 1e6, 2) for w in table[:, 3]],
                         "kernel_us_per_tick": [round(float(k) / n * 1e6, 2) for k in table[:, 1]]},
            "value_min_rank": round(float((table[:, 0] / table[:, 3]).min()), 1), "value_max_rank": round(float((table[:, 0] / table[:, 3]).max()), 1),
            "roofline": {"kernel": "k_run<512, fixed, kKindAll> (rl_run, mixed brain kinds)", "rank": 0, "bound": "hbm", "achieved": round(steps * by / t_k / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(steps * by / t_k / 1e9 / HBM_PEAK_GBS, 5), "bytes_per_agent_step": by,
                         "flop_per_agent": flop, "mfma_tflops": round(steps * flop / t_k / 1e12, 2),
                         "mfma_frac": round(steps * flop / t_k / 1e12 / MFMA_F32_EQUIV_PEAK_TFLOPS, 5),
                         "how": "rank 0's launch: HIP events on the launch stream around ONE launch of %d ticks, queued behind %d untimed ticks" % (n, n),
                         **latency_bound("c5", args.worlds, t_k / n)},
            "timed": "wall clock around the launch + synchronize on every rank, released together by a barrier; value = all ranks' agent-steps / "
                     "the slowest rank's wall time; inputs resident in HBM"}


def latency_bound(workload, worlds, tick_s):
    """roofline.latency_bound_us / frac_of_latency_bound (tools/latency_bound.py): the part of a k_run tick that is the policy
    half's floor (matrix-pipe issue of the slowest SIMD, or the world's weight stream through the CU's load path, whichever is longer) + the
    tick's barriers + the sections one wave executes alone -- what remains if every data-parallel section cost
    nothing -- as a fraction of the tick (taken in a stamped build, profiles/latency_model.json, valid for the kernel sources it is stamped
    with) times the tick measured HERE.  At one world per CU this, not the HBM or MFMA peak, is the floor the structure can approach."""
    path = os.path.join(ROOT, "profiles", "latency_model.json")
    out = {"latency_bound_us": None, "frac_of_latency_bound": None}
    try:
        m = json.load(open(path))
        from reinlife_amd import build as _build
        w = m["workloads"].get(workload)
        if w and w.get("frac_of_latency_bound") and worlds <= 256 and m.get("kernel_src_sha16") == _build.source_hash():
            out = {"latency_bound_us": round(w["frac_of_latency_bound"] * tick_s * 1e6, 2), "frac_of_latency_bound": w["frac_of_latency_bound"],
                   "latency_model": {"file": "profiles/latency_model.json", "mfma_slowest_simd_counts": w["mfma_slowest_simd"]["counts"],
                                     "weight_stream_counts": w["weight_stream"]["counts"], "weight_stream_kb_per_world_tick": w["weight_stream"]["kb_per_world_tick"],
                                     "cu_load_bytes_per_clock": w["weight_stream"]["cu_load_bytes_per_clock"], "policy_floor_counts": w["policy_floor_counts"],
                                     "barriers_counts": w["barriers"]["counts"], "one_wave_sections_counts": w["one_wave_sections_counts"],
                                     "bound_counts": w["bound_counts"], "stamped_tick_counts": w["stamped_tick_counts"],
                                     "how": "bound counts / the stamped build's tick counts x avg_tick_us; one world per CU: max(matrix-pipe issue of the slowest "
                                            "SIMD, the world's weight stream through the CU's 64 B/clk load path) + workgroup barriers + one-wave sections (DESIGN.md 6.2)"}}
    except Exception:  # noqa: BLE001
        pass
    return out


def tuning_halves(args):
    """tools/run_halves.py --json in a subprocess that loads lib/libreinlife_hip_tune.so (RL_TUNE=1 python reinlife_amd/build.py); an
    {"error": ...} when that library is not built for the current sources -- the benchmark never builds or loads it itself."""
    env = dict(os.environ, RL_TUNE="1")
    env.pop("REINLIFE_HIP_LIB", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_halves.py"), "--json", "--no-build", "--worlds", str(args.worlds),
                              "--workload", args.workload, "--seed", str(args.seed)], env=env, capture_output=True, text=True, timeout=300)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": "tools/run_halves.py failed: %r" % (e,)}


def torchrun_command(args, argv, n_visible):
    """(cmd, env) of `python bench.py --gpus N` with N > 1 and no torchrun environment: N ranks of this script, one per GPU, through
    torch.distributed.run on 127.0.0.1 with a free port -- the launch line the driver uses.  Refuses more ranks than visible GPUs
    unless the ranks share one GPU (--share-gpu, gloo dry run)."""
    if not args.share_gpu and n_visible < args.gpus:
        raise SystemExit("bench.py: --gpus %d but only %d GPUs are visible" % (args.gpus, n_visible))
    if args.share_gpu and n_visible < 1:
        raise SystemExit("bench.py: --share-gpu needs one visible GPU, there is none")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return cmd, env


def respawn_under_torchrun(args):
    cmd, env = torchrun_command(args, sys.argv[1:], torch.cuda.device_count())
    if os.environ.get("RL_BENCH_PRINT_SPAWN"):   # (tests/test_bench_cpu.py: the launch line, without launching)
        print(json.dumps({"cmd": cmd, "HSA_ENABLE_IPC_MODE_LEGACY": env["HSA_ENABLE_IPC_MODE_LEGACY"]}))
        raise SystemExit(0)
    # the library is brought up to date HERE, once, before N ranks exist: they then find it current (a few file digests each, no lock, no
    # compiler).  Under the driver's own torchrun line there is no parent: build() is serialised by an flock and moves its files into
    # place atomically (reinlife_amd/build.py), so N ranks on a stale tree compile once and never map a half-written library.
    _lib.lib()
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def check_rank_environment(args, n_visible):
    """(world_size, rank, local_rank, device index) of this process, or SystemExit with the mismatch: WORLD_SIZE against --gpus, LOCAL_RANK
    against the visible GPUs (every rank on cuda:0 with --share-gpu)."""
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world_size))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu and args.dist_backend != "gloo":
        raise SystemExit("bench.py: --share-gpu is the gloo dry run (RCCL refuses two ranks on one device): add --dist-backend gloo")
    dev_index = 0 if args.share_gpu else local_rank
    if dev_index >= n_visible:
        raise SystemExit("bench.py: LOCAL_RANK %d but only %d GPUs visible" % (local_rank, n_visible))
    return world_size, rank, local_rank, dev_index


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough to be past the start-up transient (every world starts with one cohort of 100 agents, so the first
    # refills come in waves; the worlds drift apart within a few hundred ticks); 2,300 steps are ~80 ms of GPU time
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--worlds", type=int, default=256, help="worlds per GPU")
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--seed", type=int, default=20260928)
    ap.add_argument("--groups", type=int, default=1, help="split the GPU's worlds into this many stream groups (overlap)")
    ap.add_argument("--path", default="auto", choices=["auto", "fused", "two-launch"],
                    help="fused = one multi-tick launch (rl_run); two-launch = rl_policy_act + rl_tick_refill per tick")
    ap.add_argument("--burnin", type=int, default=2000, help="untimed set-up ticks: de-synchronise the worlds' cohorts, bring the chip to its working clocks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api-trainer", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the configs[4] per-GPU leg of the default (c4) line")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-single-world", action="store_true", help="skip the configs[1] / configs[2] single-world figures (N = 1 only)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend: nccl = RCCL over xGMI (the product); gloo = dry run of the N > 1 path (collectives hop through the host)")
    ap.add_argument("--share-gpu", action="store_true", help="dry run: every rank on cuda:0 (needs --dist-backend gloo)")
    ap.add_argument("--dist-timeout", type=float, default=180.0, help="process-group timeout in seconds (rendezvous and every collective)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)
    world_size, rank, local_rank, dev_index = check_rank_environment(args, torch.cuda.device_count())
    dist = None
    if world_size > 1 or os.environ.get("RL_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world_size))
        torch.cuda.set_device(dev_index)
        # a rank that dies in set-up must not leave the others in a collective for the backend's default 10 / 30 minutes
        from datetime import timedelta
        pg_timeout = timedelta(seconds=args.dist_timeout)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=pg_timeout)
        else:
            dist.init_process_group("gloo", timeout=pg_timeout)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
    n_ranks = dist.get_world_size() if dist is not None else 1
    backend = rl_dist.backend_of(dist) if dist is not None else None
    rccl_ranks = n_ranks if backend in (None, "nccl") else 0   # the world size RCCL saw (1 without a process group; 0 = a gloo dry run)
    device = "cuda:%d" % dev_index
    torch.cuda.set_device(dev_index)

    if args.groups > 1:
        grp = StreamGroups(args, rank, device)
        dw = grp.parts[0]
        step_all = grp.step
    else:
        grp = None
        dw = make_worlds(args, rank, device)
        step_all = lambda: one_step(dw)  # noqa: E731
    fused = grp is None and args.path != "two-launch" and dw.run_supported() and (args.worlds <= 768 or args.path == "fused")
    if args.path == "fused" and not fused:
        raise SystemExit("bench.py: --path fused is not available for this workload (brain kinds) / --groups")

    timed_events = []

    def advance(n, events=None):
        if grp:
            for _ in range(n):
                step_all()
        else:
            run_ticks(dw, n, fused, events)

    advance(args.burnin)   # set-up, untimed: past the start-up transient (see the module docstring)
    torch.cuda.synchronize()
    advance(args.warmup)
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps, inputs resident in HBM ------------------------------------------------------
    if grp:
        grp.zero_counters()
    else:
        dw.acted_total.zero_()
        dw.refill_count.zero_()
    if dist is not None:
        dist.barrier()
        try:  # RCCL prints its banner through C stdio: push it out now so that the JSON line is the LAST line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t0_shared = time.monotonic_ns()   # CLOCK_MONOTONIC: one clock for every process of the node -- the ranks' starts and ends are comparable
    # (HIP events around the launches of the timed region itself when it is long; a short region -- the driver's 20 steps are ONE launch
    # of ~0.5 ms -- is not burdened with the two event records (~1 % of it): its launch is replayed right afterwards, see below)
    events_in_region = fused and not args.no_kernel_timing and args.steps >= 200
    advance(args.steps, timed_events if events_in_region else None)
    torch.cuda.synchronize()
    # This rank's K steps end when ITS device is idle.  The JOB's time is taken on the node's shared clock: from the EARLIEST rank's start
    # (its exit from the opening barrier + synchronise) to the LATEST rank's end -- start skew between the ranks is inside the figure, the
    # closing barrier's own latency (a collective, not work of the K steps) is not; the bracketed figure (closing barrier + synchronise
    # included: the contract's literal bracket) is reported beside it as value_incl_closing_barrier.  One rank: all three are the same interval.
    t1_shared = time.monotonic_ns()
    elapsed = (t1_shared - t0_shared) * 1e-9   # (this rank's own K steps, on the same clock as the job's figure: own <= job by construction)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed_bracket = time.perf_counter() - t0
    (grp or dw).check_error_flag()

    # the only collective of the job: ONE RCCL all-gather of every rank's [agent-steps, refills, world_base, bracket, start, end, own elapsed]
    # row over xGMI (reinlife_amd/distributed.py); executed whenever a process group exists, also with one rank.  (Nanoseconds of
    # CLOCK_MONOTONIC = time since boot: < 2^53 for 104 days of uptime, exact in float64.)
    counted = grp.counters() if grp else (float(dw.acted_total.item()), float(dw.refill_count.item()))
    stats = torch.tensor(list(counted) + [float(dw.world_base), elapsed_bracket, float(t0_shared), float(t1_shared)], dtype=torch.float64, device=device)
    stats, elapsed_own_max, rank_table = reduce_counters(stats, elapsed, dist)
    total_agent_steps, refills = stats.tolist()[:2]
    rank_rates = (rank_table[:, 0] / rank_table[:, -1]).tolist()   # each rank's own agent-steps/s over its own clock: stragglers show here
    starts_ns, ends_ns = rank_table[:, 4], rank_table[:, 5]
    elapsed = float(ends_ns.max() - starts_ns.min()) * 1e-9 if n_ranks > 1 else elapsed_own_max   # latest end - earliest start
    elapsed_bracket_max = float(rank_table[:, 3].max())

    # ---- per-kernel durations with HIP events on the launch stream (untimed extra steps) ------------------------------
    roofline, extra = None, {}
    fused_roof = None
    if rank == 0 and not args.no_kernel_timing and fused:
        # the multi-tick launch, HIP events around a launch of N ticks
        def timed_run(n):
            # launches back to back in stream order, the last one between the events: the ones before it bring the chip to its
            # working clocks (a launch that follows a host round trip starts on a chip that has begun to clock down, and behind a
            # 0.5 ms launch it still is at its idle clocks: tools/launch_cost.py, tools/window_probe.py), and nothing but the
            # kernel lies between the events
            if n < 1000:
                dw.run(1000, 70, 100)
            dw.run(n, 70, 100)
            before = dw.acted_total.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dw.run(n, 70, 100); e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / n, (int(dw.acted_total.item()) - int(before.item())) / n
        # the dominant kernel over the TIMED REGION: HIP events around its launches there (recorded on the launch stream); for a short
        # region, around a replay of its launch (same number of ticks) queued directly behind a warm launch, so that the events see
        # the kernel and not the host's launch latency in front of it
        if events_in_region:
            t_all = sum(e0.elapsed_time(e1) for e0, e1, _ in timed_events) * 1e-3 / sum(k for _, _, k in timed_events)
            per_tick = float(rank_table[rank, 0]) / args.steps
            launches = [k for _, _, k in timed_events]
        else:
            t_all, per_tick = timed_run(min(args.steps, FUSED_CHUNK))
            launches = [min(args.steps, FUSED_CHUNK)]
        wl = WORKLOADS[args.workload]
        flop = float(np.mean([POLICY_FLOP_PER_AGENT[n] for n in wl["brains"]]))
        by = TICK_BYTES_PER_AGENT_STEP + POLICY_BYTES_PER_AGENT
        fused_roof = {"kernel": "k_run (rl_run: policy + step + update_env + refill, worlds resident in LDS)", "bound": "hbm",
                      "achieved": round(per_tick * by / t_all / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(per_tick * by / t_all / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                      "avg_tick_us": round(t_all * 1e6, 2), "ticks_per_launch": launches, "agent_steps_per_tick": round(per_tick, 1),
                      "how": "HIP events on the launch stream around " + ("the launches of the timed region" if events_in_region else
                             "a replay of the timed region's launch, queued directly behind a warm launch (the region itself is too short to carry event records)"),
                      "bytes_per_agent_step": by,
                      "mfma_tflops": round(per_tick * flop / t_all / 1e12, 2), "mfma_frac": round(per_tick * flop / t_all / 1e12 / MFMA_F32_EQUIV_PEAK_TFLOPS, 5)}
        # PMC bytes need rocprofv3 (tools/pmc_traffic.sh), so they come from a tracked file -- stamped with the hash of the kernel
        # sources they were measured on: when the sources have changed since, the figure is dropped, not silently carried along
        tpath = os.path.join(ROOT, "profiles", "run_traffic.json")
        if os.path.exists(tpath) and args.worlds == 256 and args.workload == "c4":
            try:
                tj = json.load(open(tpath))
                from reinlife_amd import build as _build
                now = _build.source_hash()
                fused_roof["traffic"] = tj.get("hbm_bytes_per_tick") if tj.get("kernel_src_sha16") == now else None
                fused_roof["traffic_source"] = {"file": "profiles/run_traffic.json", "measured_on_kernel_src_sha16": tj.get("kernel_src_sha16"),
                                                "current_kernel_src_sha16": now}
            except Exception:  # noqa: BLE001
                pass
        fused_roof.update(latency_bound(args.workload, args.worlds, t_all))
        # the launch's two halves alone: measured by the TUNING library in a process of its own (tools/run_halves.py: the switch that
        # skips half of every tick -- results WRONG by design -- is not in the product library this process has loaded)
        halves = tuning_halves(args)
        if "error" in halves:
            fused_roof["halves"] = halves["error"]
        else:
            t_tick_half, n_tick_half = halves["tick_half_us"] * 1e-6, halves["tick_half_agent_steps_per_tick"]
            t_pol_in = max(halves["full_us"] * 1e-6 - t_tick_half, 1e-9)
            pt = halves["agent_steps_per_tick"]
            fused_roof["tick_half"] = {"us_per_tick": round(t_tick_half * 1e6, 2), "hbm_GBs": round(n_tick_half * TICK_BYTES_PER_AGENT_STEP / t_tick_half / 1e9, 1),
                                       "hbm_frac": round(n_tick_half * TICK_BYTES_PER_AGENT_STEP / t_tick_half / 1e9 / HBM_PEAK_GBS, 4),
                                       "how": "tools/run_halves.py on %s: launches with the policy half skipped (run mask 1)" % halves["library"]}
            # the policy half inside a full tick = the tick minus the tick half alone (the policy alone, with the tick half skipped, reads its
            # rows from memory instead of the LDS mirror the tick half fills, and is slower than in place)
            fused_roof["policy_half"] = {"us_per_tick": round(t_pol_in * 1e6, 2), "mfma_tflops": round(pt * flop / t_pol_in / 1e12, 1),
                                         "mfma_frac": round(pt * flop / t_pol_in / 1e12 / MFMA_F32_EQUIV_PEAK_TFLOPS, 4),
                                         "alone_us_per_tick": round(halves["policy_alone_us"], 2),
                                         "how": "per-tick time of a %d-tick launch of the same library (%.2f us) - tick_half.us_per_tick; alone_us_per_tick = launches with the tick half skipped (run mask 2: rows from memory, not from the LDS mirror)" % (halves["ticks"], halves["full_us"])}
    if rank == 0 and not args.no_kernel_timing:
        # (with --groups G the probe runs group 0 alone: its launches cover worlds/G worlds each)
        # back-to-back launches, no host sync inside the probe (a launch from an idle stream costs ~8 us extra): the event
        # pairs then agree with rocprofv3's per-kernel average (profiles/r01c_kernel_stats.txt)
        n_probe = 60
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_probe)]
        acted_before = int(dw.acted_total.item())
        for _ in range(5):
            one_step(dw)
        acted_before = int(dw.acted_total.item())
        for i in range(n_probe):
            ev[i][0].record(); dw.act(); ev[i][1].record(); dw.tick_refill(70, 100); ev[i][2].record()
        torch.cuda.synchronize()
        probe_acted = int(dw.acted_total.item()) - acted_before
        t_act = float(np.median([e[0].elapsed_time(e[1]) for e in ev])) * 1e-3
        t_tick = float(np.median([e[1].elapsed_time(e[2]) for e in ev])) * 1e-3
        per_launch = probe_acted / n_probe
        wl = WORKLOADS[args.workload]
        flop = np.mean([POLICY_FLOP_PER_AGENT[n] for n in wl["brains"]])
        tick_gbs = per_launch * TICK_BYTES_PER_AGENT_STEP / t_tick / 1e9
        pol_tflops = per_launch * flop / t_act / 1e12
        traffic = pol_traffic = None  # PMC HBM bytes per launch, measured separately (tools/pmc_traffic.sh, 256 worlds/GPU)
        tpath = os.path.join(ROOT, "profiles", "tick_traffic.json")
        if os.path.exists(tpath) and args.worlds == 256 and args.workload == "c4" and args.groups == 1:
            try:
                tj = json.load(open(tpath))
                from reinlife_amd import build as _build
                if tj.get("kernel_src_sha16") == _build.source_hash():   # (stamped like run_traffic.json)
                    traffic, pol_traffic = tj.get("hbm_bytes_per_launch"), tj.get("policy_hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = pol_traffic = None
        tick_roof = {"kernel": "k_world<TICK> (rl_tick_refill)", "bound": "hbm", "achieved": round(tick_gbs, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(tick_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "avg_launch_us": round(t_tick * 1e6, 2), "agent_steps_per_launch": round(per_launch, 1),
                     "bytes_per_agent_step": TICK_BYTES_PER_AGENT_STEP}
        pol_roof = {"kernel": "k_policy_pair / k_policy_dense (rl_policy_act: the tiles of rl_run's policy half as a stand-alone launch)", "bound": "mfma", "achieved": round(pol_tflops, 3),
                    "peak": round(MFMA_F32_EQUIV_PEAK_TFLOPS, 1), "unit": "TFLOP/s",
                    "frac": round(pol_tflops / MFMA_F32_EQUIV_PEAK_TFLOPS, 5),
                    "peak_note": "algorithmic f32-equivalent FLOP/s; peak = f16 dense %.1f TF / %d partial products of the "
                                 "block-scaled 2 x f16 split (the f32-input MFMA peak would be 157.3)" % (MFMA_BF16_PEAK_TFLOPS, MFMA_SPLIT_PRODUCTS),
                    "traffic": pol_traffic, "avg_launch_us": round(t_act * 1e6, 2), "flop_per_agent": flop,
                    "agent_steps_per_launch": round(per_launch, 1)}
        roofline = tick_roof if t_tick >= t_act else pol_roof
        extra = {"roofline_tick": tick_roof, "roofline_policy": pol_roof}
        if fused_roof is not None:   # the timed region ran the multi-tick launch: that is the dominant kernel of the line
            roofline = fused_roof
            extra["two_launch_step_us"] = round((t_act + t_tick) * 1e6, 2)
        # variant (i) of BASELINE.md 3 / SURVEY.md 8d: "get_action + step" only.  The loop still runs update_env + refill (the
        # world must go on), but only policy + rl_step lie between the event pairs.
        if args.groups == 1:
            n_var = 40
            ev2 = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(n_var)]
            for i in range(3 + n_var):
                k = i - 3
                if k >= 0:
                    ev2[k][0].record()
                dw.act(); dw.step()
                if k >= 0:
                    ev2[k][1].record()
                dw.update(); dw.refill(70, 100)
            torch.cuda.synchronize()
            n_act = dw.n_acted.sum().item()  # agents of the last step (steady regime: the per-step count varies by < 1 %)
            t_ps = float(np.median([e[0].elapsed_time(e[1]) for e in ev2])) * 1e-3
            extra["variant_policy_plus_step"] = {
                "value": round(n_act / t_ps, 1), "unit": "agent-steps/s", "us_per_step": round(t_ps * 1e6, 2),
                "what": "rl_policy_act + rl_step (Environment.step, un-fused kernel) between HIP events; update_env + refill run untimed "
                        "between the pairs; 1 GPU (rank 0)"}

    main_collectives = rl_dist.collectives_executed   # (the metric's own reduction: exactly one when a process group exists)
    api = None
    if not args.no_api_trainer and args.groups == 1:   # every rank: trainer() shards by rank and reduces its Tracker over RCCL
        api = api_trainer(args, device, dist, rank)

    c5 = None
    if args.workload == "c4" and args.groups == 1 and not args.no_c5:   # every rank: configs[4] IS the weak-scaling configuration
        c5 = c5_leg(args, rank, device, dist)

    single = None
    if rank == 0 and args.gpus == 1 and args.groups == 1 and not args.no_single_world:
        single = single_world(args, device)

    # every rank is done with its timed legs: how long each one waits here for the slowest (rank 0 carries the rank-0-only kernel probes)
    # is part of the line -- a rank stuck behind a leg the others skipped would show as seconds
    barrier_waits = [0.0]
    if dist is not None:
        tb = time.perf_counter()
        dist.barrier()
        w = torch.tensor([time.perf_counter() - tb], dtype=torch.float64, device=device)
        barrier_waits = [round(float(x), 4) for x in rl_dist.gather_rows(w, dist).reshape(-1).tolist()]

    # the CPU port on the box's host cores, in the same run, at EVERY N (north_star: "next to the reference numpy/CPU path timed on the
    # same box's host cores ... in the same run"): rank 0, AFTER every timed leg of every rank (the barrier above), the other ranks parked
    # in the barrier below -- its ~15 s of all-core work never overlap a GPU measurement.
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)
    cpu_wait = [0.0]
    if dist is not None:
        tb = time.perf_counter()
        dist.barrier()
        w = torch.tensor([time.perf_counter() - tb], dtype=torch.float64, device=device)
        cpu_wait = [round(float(x), 3) for x in rl_dist.gather_rows(w, dist).reshape(-1).tolist()]

    if rank == 0:
        wl = WORKLOADS[args.workload]
        out = {
            "metric": "agent-steps/sec (env.step + policy fwd), 30x30 grid, 100 agents",
            "value": round(total_agent_steps / elapsed, 1),
            "unit": "agent-steps/s",
            "n_gpus": args.gpus,
            "rccl_ranks": rccl_ranks,
            "rccl_collectives_executed": main_collectives,
            "ranks": n_ranks,
            "dist_backend": {None: None, "nccl": "nccl (RCCL)"}.get(backend, backend),
            "value_incl_closing_barrier": round(total_agent_steps / elapsed_bracket_max, 1),
            "value_slowest_rank_own_clock": round(total_agent_steps / elapsed_own_max, 1),
            "timed": "N > 1: all ranks' agent-steps / (latest rank's end - earliest rank's start) on the node's CLOCK_MONOTONIC, a rank's start = its exit from "
                     "the opening barrier + synchronise, its end = its own synchronise after the K steps; value_incl_closing_barrier = over the slowest "
                     "rank's bracket incl. the closing barrier + synchronise; value_slowest_rank_own_clock = over the longest of the ranks' own "
                     "start -> end intervals (start skew invisible).  N = 1: one interval, three equal figures up to the closing synchronise",
            "per_rank": {"value_min": round(min(rank_rates), 1), "value_max": round(max(rank_rates), 1),
                         "elapsed_ms": [round(float(x) * 1e3, 3) for x in rank_table[:, -1].tolist()],
                         "elapsed_incl_closing_barrier_ms": [round(float(x) * 1e3, 3) for x in rank_table[:, 3].tolist()],
                         "start_skew_us": [round(float(x - starts_ns[0]) * 1e-3, 1) for x in starts_ns.tolist()],
                         "end_skew_us": [round(float(x - ends_ns[0]) * 1e-3, 1) for x in ends_ns.tolist()],
                         "timed": "per rank: its exit from the opening barrier + synchronise -> its own synchronise after the K steps; start_skew_us / end_skew_us = "
                                  "that rank's start / end against rank 0's on the node's shared clock",
                         "agent_steps": [int(x) for x in rank_table[:, 0].tolist()], "world_base": [int(x) for x in rank_table[:, 2].tolist()],
                         "device": device if not args.share_gpu else "cuda:0 shared by every rank (dry run)",
                         "final_barrier_wait_s": barrier_waits, "cpu_baseline_wait_s": cpu_wait},
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32-grade policy via block-scaled 2 x f16 split on v_mfma_f32_32x32x16_f16 (f32 accumulate) / int32+u8 world state / f64 rewards",
            "data": "synthetic",
            "config": {"workload": workload_name(args.workload, args.worlds), "worlds_per_gpu": args.worlds, "worlds_total": args.worlds * max(1, world_size),
                       "stream_groups": args.groups, "loop": "one multi-tick launch (rl_run)" if fused else "two launches per tick (rl_policy_act + rl_tick_refill)", "grid": "30x30", "max_agents": 100, "brains": wl["brains"], "static_families": wl["static_families"],
                       "refill_below": 70, "includes_update_env": True, "burn_in_ticks": args.burnin,
                       "mean_agents_per_world": round(total_agent_steps / (args.steps * args.worlds * max(1, world_size)), 2),
                       "world_refills": int(refills), "agent_steps": int(round(total_agent_steps)), "parallelism": "replica-sharded x%d, no data-path collective" % max(1, world_size)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "api_trainer": api,
            "c5": c5,
            "single_world": single,
        }
        out.update(extra)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
