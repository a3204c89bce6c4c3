"""Quick probe of DeviceWorlds.run (multi-tick launch) against the two-launch loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from reinlife_amd import _lib
from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
args = __import__("argparse").Namespace(worlds=R, workload="c4", seed=1)
a = bench.make_worlds(args, 0, "cuda:0"); b = bench.make_worlds(args, 0, "cuda:0")
print("supported", a.run_supported(), flush=True)
a.run(K, 70, 100)
torch.cuda.synchronize()
print("ran", flush=True)
for _ in range(K):
    b.act(); b.tick_refill(70, 100)
torch.cuda.synchronize()
a.check_error_flag()
for key in a.s:
    x, y = a.s[key].cpu().numpy(), b.s[key].cpu().numpy()
    n = b.s["n_agents"].cpu().numpy()
    if key.startswith("a_"):
        bad = [w for w in range(R) if not np.array_equal(x[w, :n[w]], y[w, :n[w]])]
    else:
        bad = [] if np.array_equal(x, y) else ["all"]
    print(key, "ok" if not bad else "DIFF %s" % bad[:5])
print("actions equal", np.array_equal(a.actions.cpu().numpy()[:, :60], b.actions.cpu().numpy()[:, :60]))
# ---- timing: one launch of N ticks against the two-launch loop
for N in (50, 200):
    for dw, name in ((a, "rl_run"), (b, "two launches")):
        torch.cuda.synchronize(); t0 = time.perf_counter(); before = int(dw.acted_total.item())
        if name == "rl_run":
            dw.run(N, 70, 100)
        else:
            for _ in range(N):
                dw.act(); dw.tick_refill(70, 100)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        steps = int(dw.acted_total.item()) - before
        print("%-14s %4d ticks: %.2f us/tick, %.3e agent-steps/s" % (name, N, dt / N * 1e6, steps / dt), flush=True)
