"""rl_run's per-tick time with one half of every tick switched off (rl_debug_set_run_mask -- the TUNING library only; the product library
does not carry the switch), per workgroup size (tuning aid; GPU).
    python tools/run_halves.py [512 256 1024]                       human-readable, per workgroup size
    python tools/run_halves.py --json --worlds 256 --workload c4    ONE JSON object for bench.py (which runs this in a subprocess, so that
                                                                    the benchmarking process itself only ever loads the product library)
A launch under a mask leaves WRONG worlds behind by design; only its duration is used.  Every figure: a launch of N ticks between HIP
events, queued directly behind a warm launch of the same kind."""
import argparse
import json
import os
import sys

os.environ.setdefault("RL_TUNE", "1")   # reinlife_amd.build: TAG "_tune" -> lib/libreinlife_hip_tune.so
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("blocks", nargs="*")
ap.add_argument("--json", action="store_true")
ap.add_argument("--no-build", action="store_true", help="refuse (JSON error) instead of building when the tuning library is absent or stale")
ap.add_argument("--worlds", type=int, default=256)
ap.add_argument("--workload", default="c4")
ap.add_argument("--seed", type=int, default=20260928)
ap.add_argument("--ticks", type=int, default=200)
a = ap.parse_args()

from reinlife_amd import build as _build  # noqa: E402
if a.no_build and not _build.library_is_current():
    print(json.dumps({"error": "the tuning library is not built for these sources: RL_TUNE=1 python reinlife_amd/build.py"}))
    raise SystemExit(0)

import torch  # noqa: E402
import bench  # noqa: E402
from reinlife_amd import _lib  # noqa: E402

lib = _lib.lib()
lib.rl_debug_set_run_mask.argtypes = [__import__("ctypes").c_int]


def timed(dw, n, mask):
    """(seconds per tick, agent-steps per tick) of an n-tick launch under `mask`, behind a warm launch."""
    lib.rl_debug_set_run_mask(0)
    dw.run(1000, 70, 100)                      # the chip at its working clocks, the worlds in their steady regime
    keep = {k: v.clone() for k, v in dw.s.items()}
    obs = [o.clone() for o in dw._obs2]
    lib.rl_debug_set_run_mask(mask)
    try:
        dw.run(n, 70, 100)
        before = dw.acted_total.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dw.run(n, 70, 100); e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / n, (int(dw.acted_total.item()) - int(before.item())) / n
    finally:
        lib.rl_debug_set_run_mask(0)
        for k, v in keep.items():              # a masked launch leaves wrong worlds behind: restore
            dw.s[k].copy_(v)
        for o, k in zip(dw._obs2, obs):
            o.copy_(k)


args = argparse.Namespace(worlds=a.worlds, workload=a.workload, seed=a.seed)
if a.json:
    dw = bench.make_worlds(args, 0, "cuda:0")
    t_full, per_tick = timed(dw, a.ticks, 0)
    t_tick, n_tick = timed(dw, a.ticks, 1)
    t_pol, _ = timed(dw, a.ticks, 2)
    print(json.dumps({"ticks": a.ticks, "full_us": t_full * 1e6, "agent_steps_per_tick": per_tick, "tick_half_us": t_tick * 1e6,
                      "tick_half_agent_steps_per_tick": n_tick, "policy_alone_us": t_pol * 1e6, "library": os.path.basename(_build.LIB_PATH)}))
else:
    for blk in a.blocks or ("512", "256", "1024"):
        _lib.set_option("world_block", blk)
        dw = bench.make_worlds(args, 0, "cuda:0")
        for mask in (0, 1, 2):
            t, n = timed(dw, 300, mask)
            print("block %s run mask %d (1 = no policy, 2 = no tick): %.2f us/tick  %.3e agent-steps/s" % (blk, mask, t * 1e6, n / t), flush=True)
