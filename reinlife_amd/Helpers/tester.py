"""tester(): same signature as ReinLife/Helpers/tester.py:6-13.  The reference loops forever and renders to a pygame
window; here every iteration paints `env.frame` (Helpers/render.py), an optional keyword-only `n_steps` bounds the loop,
`on_frame(env)` (keyword-only) is called after each render, and the environment is returned.  Under an initialised torch.distributed
process group (or `dist=`) every rank shows its own block of replicas (Environment: world_base = rank * n_worlds, cuda:LOCAL_RANK)."""
from ..World.environment import Environment


def tester(brains, width=30, height=30, max_agents=100, pastel_colors=False, static_families=True, limit_reproduction=False,
           fps=10, *, n_steps=None, n_worlds=1, device=None, seed=0, rng=None, on_frame=None, dist=None, world_base=None):
    env = Environment(width=width, height=height, grid_size=24, max_agents=max_agents, pastel_colors=pastel_colors,
                      brains=brains, training=False, static_families=static_families, limit_reproduction=limit_reproduction,
                      n_worlds=n_worlds, device=device, seed=seed, rng=rng, dist=dist, world_base=world_base)
    env.reset()
    env.render(fps=fps)  # tester.py:55
    step = 0
    while n_steps is None or step < n_steps:
        if env.rng == "philox":
            env.run(0, 1)   # one launch per frame: get_action (n_epi = 0, tester.py:57-68) + step + update_env
        else:
            env.act(0)  # tester.py:57-68: every brain is asked with n_epi = 0
            env.step()
            env.update_env()
        env.render(fps=fps)
        if on_frame is not None:
            on_frame(env)
        step += 1
    return env
