"""Why are the ticks of a short launch slower?  (a) cycles per tick do not depend on the launch length (tools/run_first_tick.py), so it is
time per cycle; (b) here: 60 launches of 20 ticks queued back to back without a host round trip, HIP events around every one -- if the
chip's clock is what differs, the per-tick time converges to the long launch's once the load is sustained (tuning; GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=1)
a = bench.make_worlds(args, 0, "cuda:0")
a.run(600, 70, 100); torch.cuda.synchronize()
for gap_ms in (0.0, 5.0):
    time.sleep(gap_ms * 1e-3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    ev[0].record()
    for i in range(60):
        a.run(20, 70, 100); ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(60)]
    print("after a %.0f ms idle gap, 60 x 20-tick launches back to back, us per launch: first %s ... 10th %.0f, 30th %.0f, last %.0f (a 2000-tick launch: %.0f per 20 ticks)" % (
        gap_ms, [round(x) for x in d[:5]], d[9], d[29], d[59], 20 * 23.6), flush=True)
torch.cuda.synchronize(); time.sleep(0.005)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); a.run(2000, 70, 100); e1.record(); torch.cuda.synchronize()
print("2000-tick launch after a 5 ms gap: %.2f us per tick" % (e0.elapsed_time(e1) * 1e3 / 2000))
