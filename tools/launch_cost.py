"""Fixed cost of one rl_run launch: wall time of run(n) + synchronize for n = 1, 2, 5, 20, 100 (tuning; GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=1)
a = bench.make_worlds(args, 0, "cuda:0")
a.run(300, 70, 100); torch.cuda.synchronize()
for n in (1, 2, 5, 20, 100, 500):
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); a.run(n, 70, 100); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts.sort(key=lambda x: x[1])
    h, tot = ts[len(ts) // 2]
    print("n=%4d: host call %.1f us, call + sync %.1f us  (%.2f us per tick)" % (n, h * 1e6, tot * 1e6, tot * 1e6 / n), flush=True)
