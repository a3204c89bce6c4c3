"""trainer(): same signature, defaults and return value as ReinLife/Helpers/trainer.py:7-107.

The loop is the reference's (get_action -> step -> learn -> update_env, trainer.py:85-99) with get_action batched on the
GPU through env.act().  A single world (rng="reference", the default for n_worlds == 1 outside a multi-rank job) makes every random draw of the
loop exactly as the reference does, so the same seeds give the same run; replicated worlds draw in-kernel.

INFERENCE ONLY: training=True (the reference's default) keeps the loop, the epsilon schedules and the Tracker, but the brains
of this build do not learn -- Environment warns about it, and save=True writes the weights as they were loaded / initialised
(settings.json says so).  Training is outside this build's scope (BASELINE.json north_star)."""
import time

import torch

from ..World.environment import Environment


def trainer(brains, n_episodes=10_000, width=30, height=30, visualize_results=False, google_colab=False, update_interval=500,
            print_results=True, max_agents=100, render=False, static_families=True, training=True, save=True,
            limit_reproduction=False, incentivize_killing=True, *, n_worlds=1, device=None, seed=0, rng=None, per_agent_api=False,
            fused=None, synthetic_agents=None, refill_below=None, dist=None, world_base=None):
    """Extra keyword-only arguments: n_worlds / device / seed / rng / synthetic_agents / refill_below (Environment); per_agent_api=True makes the reference's literal
    per-agent get_action / learn calls; fused (default: True for rng="philox" without per_agent_api) runs the loop through
    Environment.run -- whole chunks of ticks per launch, ending where the Tracker closes an interval -- instead of three launches
    and a host round trip per tick; fused=False keeps the tick-by-tick loop (same results, tests/test_hip_round3.py).
    The wall time of the loop itself is left in env.loop_seconds.
    Several GPUs: start one process per GPU (torchrun) and initialise torch.distributed (backend "nccl" = RCCL) before the
    call -- or pass the process-group module as `dist`.  Every rank then owns `n_worlds` replicas (global ids rank * n_worlds ...: the
    worlds are the same whatever the number of ranks), runs on cuda:LOCAL_RANK unless `device` says otherwise, and the Tracker's
    per-interval statistics (tracker.py:107-121) are pooled over ALL ranks' worlds by one all-gather of the ranks' per-world sums per closed interval --
    the only collective of the loop; every rank returns the same `env.tracker.results`."""
    env = Environment(width=width, height=height, max_agents=max_agents, brains=brains, grid_size=24,
                      static_families=static_families, update_interval=update_interval, print_results=print_results,
                      interactive_results=visualize_results, google_colab=google_colab, training=training,
                      limit_reproduction=limit_reproduction, incentivize_killing=incentivize_killing, n_worlds=n_worlds,
                      device=device, seed=seed, rng=rng, synthetic_agents=synthetic_agents, refill_below=refill_below, dist=dist,
                      world_base=world_base)
    env.reset()
    if fused is None:
        fused = env.rng == "philox" and not per_agent_api
    if fused and (env.rng != "philox" or per_agent_api):
        raise ValueError("trainer(fused=True) needs rng='philox' and per_agent_api=False")
    if env.rng == "philox":
        env._bind_brains()   # set-up like reset(): the brains' weights are packed for the matrix cores and uploaded once
    env._sync()
    t0 = time.perf_counter()
    if fused and not render:
        env.run(0, n_episodes + 1)      # trainer.py:85-99, n_episodes + 1 iterations
    else:
        for n_epi in range(n_episodes + 1):
            if fused:
                env.run(n_epi, 1)
            elif per_agent_api:  # the reference's literal per-agent calls (slow; API compatibility)
                for agent in env.agents:
                    agent.get_action(n_epi)
                env.step()
                if training:
                    for agent in env.agents:
                        agent.learn(n_epi=n_epi)
                env.update_env(n_epi)
            else:
                env.act(n_epi)
                env.step()
                env.update_env(n_epi)
            if render:  # trainer.py:101-102
                env.render(fps=120)
    torch.cuda.synchronize(env.worlds.device)      # the loop is over when the device is
    env.loop_seconds = time.perf_counter() - t0
    env.worlds.check_error_flag()                  # (a read-back of its own: after the clock)
    if save:
        env.save_results()
    return env
