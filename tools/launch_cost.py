"""Fixed cost of one rl_run launch: wall time of run(n) + synchronize for n = 1 .. 500, and the driver's 20-step window after different
kinds of gaps (tuning; GPU)."""
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
def shot(prep):
    ts = []
    for _ in range(9):
        a.run(5, 70, 100); torch.cuda.synchronize()
        prep()
        torch.cuda.synchronize(); t0 = time.perf_counter(); a.run(20, 70, 100); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e6, ts[0] * 1e6, ts[-1] * 1e6
print("back to back:            median %.0f us (min %.0f, max %.0f)" % shot(lambda: None))
print("after two zero_() fills: median %.0f us (min %.0f, max %.0f)" % shot(lambda: (a.acted_total.zero_(), a.refill_count.zero_())))
print("after a 2 ms sleep:      median %.0f us (min %.0f, max %.0f)" % shot(lambda: time.sleep(0.002)))
print("after a 20 ms sleep:     median %.0f us (min %.0f, max %.0f)" % shot(lambda: time.sleep(0.02)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); a.run(20, 70, 100); e1.record(); torch.cuda.synchronize()
print("HIP events around the same launch: %.0f us" % (e0.elapsed_time(e1) * 1e3))
