"""Kernel time (HIP events) of rl_run launches of 1 .. 500 ticks with and without the staggered workgroup starts (rl_debug_set_run_mask & 32 switches
them off; tuning; GPU)."""
import os, sys, time
os.environ.setdefault("RL_TUNE", "1")   # the tuning library (libreinlife_hip_tune.so) carries rl_debug_set_run_mask; the product does not
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=1)
for dbg in ("32", "0"):
    __import__("reinlife_amd._lib", fromlist=["lib"]).lib().rl_debug_set_run_mask(int(dbg))
    a = bench.make_worlds(args, 0, "cuda:0")
    a.run(300, 70, 100); torch.cuda.synchronize()
    for n in (1, 5, 20, 100, 500):
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record(); a.run(n, 70, 100); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print("%s n=%d: %.1f us (%.2f per tick)" % ("no stagger" if dbg == "32" else "staggered ", n, ts[len(ts)//2], ts[len(ts)//2] / n), flush=True)
