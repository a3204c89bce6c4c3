"""Long free run: the HIP path (policy-driven actions, fused tick + refill) against the oracle fed the same actions
(one-off confidence check; GPU + oracle; ~1-2 minutes).   python tools/soak_parity.py [worlds] [ticks] [static|nonstatic] [fused] [kinds]
("fused": every tick is one rl_run(1) launch -- policy + tick + refill in the multi-tick kernel -- instead of the two launches;
kinds: comma-separated brain kinds, default PERD3QN,PERD3QN -- e.g. PPO,PERD3QN runs the mixed-kind kernel; "train" as a sixth argument
adds the Tracker accumulators and a per-tick epsilon schedule, i.e. the TRAIN instantiation)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from reinlife_amd import _lib  # noqa: E402
from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 600
static = (sys.argv[3] != "nonstatic") if len(sys.argv) > 3 else True
fused = len(sys.argv) > 4 and sys.argv[4] == "fused"
kinds = sys.argv[5].split(",") if len(sys.argv) > 5 else ["PERD3QN", "PERD3QN"]
train = len(sys.argv) > 6 and sys.argv[6] == "train"
cfg = dict(width=30, height=30, max_agents=100, n_brains=len(kinds), static_families=static, limit_reproduction=False, incentivize_killing=True)
dw = DeviceWorlds(n_worlds=R, seed=4242, **cfg)
ow = orc.OracleWorlds(n_worlds=R, seed=4242, **cfg)
dw.set_brains([(_lib.KIND_BY_METHOD[n], 0.1 * k, pack_brain_weights(_lib.KIND_BY_METHOD[n], bench.brain_weights(n, 100 + k)))
               for k, n in enumerate(kinds)])
if train:
    dw.enable_tracking(True)
rng = np.random.RandomState(1)
dw.reset_synthetic(100); ow.reset_synthetic(100)
steps = 0
for t in range(ticks):
    if fused:
        dw.run(1, 70, 100, eps_schedule=rng.uniform(0, 0.3, size=(1, len(kinds))).astype(np.float32) if train else None)
    else:
        dw.act()
    acts = dw.actions.cpu().numpy().copy()
    n = ow.s["n_agents"].copy()
    steps += int(n.sum())
    if not fused:
        dw.tick_refill(70, 100)
    ow.step(acts)
    if train and t % 25 == 24:
        assert np.array_equal(dw.trk_sum.cpu().numpy(), ow.trk_sum) and np.array_equal(dw.trk_cnt.cpu().numpy(), ow.trk_cnt), (t, "tracker")
    ow.update(); ow.refill(70, 100)
    if t % 25 == 24 or t == ticks - 1:
        torch.cuda.synchronize(); dw.check_error_flag()
        for key in dw.s:
            got, want = dw.s[key].cpu().numpy(), ow.s[key]
            if key.startswith("a_"):
                nn = ow.s["n_agents"]
                for w in range(R):
                    assert np.array_equal(got[w, :nn[w]], want[w, :nn[w]]), (t, key, w)
            else:
                assert np.array_equal(got.reshape(want.shape), want), (t, key)
        o = dw.obs_state().cpu().numpy()
        for w in range(R):
            assert np.array_equal(o[w, :ow.s["n_agents"][w]], ow.obs2[w, :ow.s["n_agents"][w]]), (t, "obs2", w)
print("soak ok (%s, %s families, brains %s%s): %d worlds x %d ticks, %d agent-steps, %d refills, state and observations bit-identical to the oracle"
      % ("rl_run" if fused else "two launches", "static" if static else "non-static", "+".join(kinds), ", TRAIN launch" if train else "", R, ticks, steps, int(dw.refill_count.item())))
