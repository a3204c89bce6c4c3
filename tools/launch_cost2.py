"""Where the driver's 20-step window loses time: one rl_run(20) + synchronize after different kinds of gaps (tuning; GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=1)
a = bench.make_worlds(args, 0, "cuda:0")
a.run(300, 70, 100); torch.cuda.synchronize()
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
