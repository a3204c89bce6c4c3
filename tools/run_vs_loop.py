"""rl_run (one multi-tick launch) against the two-launch loop at several world counts (tuning aid; GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
for R in (256, 512, 1024, 4096):
    args = __import__("argparse").Namespace(worlds=R, workload="c4", seed=1)
    a = bench.make_worlds(args, 0, "cuda:0"); b = bench.make_worlds(args, 0, "cuda:0")
    N = 300 if R <= 1024 else 100
    a.run(N, 70, 100); torch.cuda.synchronize()
    before = int(a.acted_total.item()); t0 = time.perf_counter(); a.run(N, 70, 100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r1 = (int(a.acted_total.item()) - before) / dt
    for _ in range(N):
        b.act(); b.tick_refill(70, 100)
    torch.cuda.synchronize(); before = int(b.acted_total.item()); t0 = time.perf_counter()
    for _ in range(N):
        b.act(); b.tick_refill(70, 100)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print("%5d worlds: rl_run %.2f us/tick %.3e /s | two launches %.2f us/tick %.3e /s" % (R, dt / N * 1e6, r1, dt2 / N * 1e6, (int(b.acted_total.item()) - before) / dt2), flush=True)
