"""Timing of the multi-tick launch with halves switched off (RL_RUN_DEBUG: tuning only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args = __import__("argparse").Namespace(worlds=R, workload="c4", seed=1)
for dbg in ("0", "1", "2", "3"):
    os.environ["RL_RUN_DEBUG"] = dbg
    a = bench.make_worlds(args, 0, "cuda:0")
    a.run(100, 70, 100) if dbg == "0" else a.run(5, 70, 100)
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter(); a.run(N, 70, 100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("RL_RUN_DEBUG=%s (1 = no policy, 2 = no tick): %.2f us/tick" % (dbg, dt / N * 1e6), flush=True)
