"""Long free run: the HIP path (policy-driven actions, fused tick + refill) against the oracle fed the same actions
(one-off confidence check; GPU + oracle; ~1-2 minutes).   python tools/soak_parity.py [worlds] [ticks] [static|nonstatic] [fused]
("fused": every tick is one rl_run(1) launch -- policy + tick + refill in the multi-tick kernel -- instead of the two launches)"""
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
cfg = dict(width=30, height=30, max_agents=100, n_brains=2, static_families=static, limit_reproduction=False, incentivize_killing=True)
dw = DeviceWorlds(n_worlds=R, seed=4242, **cfg)
ow = orc.OracleWorlds(n_worlds=R, seed=4242, **cfg)
dw.set_brains([(_lib.KIND_BY_METHOD["PERD3QN"], 0.1 * k, pack_brain_weights(_lib.KIND_BY_METHOD["PERD3QN"], bench.brain_weights("PERD3QN", 100 + k)))
               for k in range(2)])
dw.reset_synthetic(100); ow.reset_synthetic(100)
steps = 0
for t in range(ticks):
    if fused:
        dw.run(1, 70, 100)
    else:
        dw.act()
    acts = dw.actions.cpu().numpy().copy()
    n = ow.s["n_agents"].copy()
    steps += int(n.sum())
    if not fused:
        dw.tick_refill(70, 100)
    ow.step(acts); ow.update(); ow.refill(70, 100)
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
print("soak ok (%s, %s families): %d worlds x %d ticks, %d agent-steps, %d refills, state and observations bit-identical to the oracle"
      % ("rl_run" if fused else "two launches", "static" if static else "non-static", R, ticks, steps, int(dw.refill_count.item())))
