"""The driver's window (`bench.py --steps 20 --warmup 5`) taken apart on the host side (tuning; GPU): the same sequence of calls as
bench.py's timed region, repeated, with the wall time of every piece."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = argparse.Namespace(worlds=256, workload="c4", seed=20260928)
dw = bench.make_worlds(args, 0, "cuda:0")
pc = time.perf_counter
for rep in range(4):
    if rep == 0:
        for _ in range(int(os.environ.get("BURNIN_LAUNCHES", "1"))):
            dw.run(int(os.environ.get("BURNIN", "300")), 70, 100)
        torch.cuda.synchronize()
    if os.environ.get("ZERO_EARLY"):
        dw.acted_total.zero_(); dw.refill_count.zero_()
    dw.run(5, 70, 100)
    torch.cuda.synchronize()
    dw.acted_total.zero_(); dw.refill_count.zero_()
    torch.cuda.synchronize()
    ev = os.environ.get("EVENTS")
    if ev:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = pc()
    if ev:
        e0.record()
    dw.run(20, 70, 100)
    if ev:
        e1.record()
    t1 = pc()
    torch.cuda.synchronize()
    t2 = pc()
    torch.cuda.synchronize()
    t3 = pc()
    print("window %d: call %.1f us  first synchronise %.1f us  second %.1f us  total %.1f us  (%.3e agent-steps/s)" % (
        rep, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6, float(dw.acted_total.item()) / (t3 - t0)), flush=True)
    if ev:
        print("          events around the launch: %.1f us" % (e0.elapsed_time(e1) * 1e3))
        dw.run(20, 70, 100)
        e0.record(); dw.run(20, 70, 100); e1.record(); torch.cuda.synchronize()
        print("          events around a replay behind a warm launch: %.1f us" % (e0.elapsed_time(e1) * 1e3))
