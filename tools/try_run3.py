"""Timing of the multi-tick launch at several workgroup sizes (RL_WORLD_BLOCK) against the two-launch loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args = __import__("argparse").Namespace(worlds=R, workload="c4", seed=1)
for blk in (sys.argv[2:] or ["1024", "512", "256"]):
    os.environ["RL_WORLD_BLOCK"] = blk
    for dbg in ("0", "1", "2"):
        os.environ["RL_RUN_DEBUG"] = dbg
        a = bench.make_worlds(args, 0, "cuda:0")
        a.run(300 if dbg == "0" else 5, 70, 100)
        torch.cuda.synchronize()
        N = 300
        before = int(a.acted_total.item())
        t0 = time.perf_counter(); a.run(N, 70, 100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("block %s RL_RUN_DEBUG=%s (1 = no policy, 2 = no tick): %.2f us/tick  %.3e agent-steps/s" % (blk, dbg, dt / N * 1e6, (int(a.acted_total.item()) - before) / dt), flush=True)
os.environ.pop("RL_RUN_DEBUG"); os.environ.pop("RL_WORLD_BLOCK")
b = bench.make_worlds(args, 0, "cuda:0")
for _ in range(300):
    b.act(); b.tick_refill(70, 100)
torch.cuda.synchronize(); before = int(b.acted_total.item()); t0 = time.perf_counter()
for _ in range(300):
    b.act(); b.tick_refill(70, 100)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("two launches: %.2f us/tick  %.3e agent-steps/s" % (dt / 300 * 1e6, (int(b.acted_total.item()) - before) / dt))
