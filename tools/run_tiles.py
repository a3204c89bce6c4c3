"""rl_run per-tick time with the policy limited to 1 / 2 / 3 tiles per world (rl_debug_set_run_mask bits 8 / 4 / 16; tuning only; GPU)."""
import os, sys, time
os.environ.setdefault("RL_TUNE", "1")   # the tuning library (libreinlife_hip_tune.so) carries rl_debug_set_run_mask; the product does not
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=1)
for dbg, what in (("0", "all tiles"), ("16", "<= 3 tiles"), ("4", "<= 2 tiles"), ("8", "1 tile"), ("1", "no policy")):
    __import__("reinlife_amd._lib", fromlist=["lib"]).lib().rl_debug_set_run_mask(int(dbg))
    a = bench.make_worlds(args, 0, "cuda:0")
    a.run(300, 70, 100); torch.cuda.synchronize()
    t0 = time.perf_counter(); a.run(300, 70, 100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-12s %.2f us/tick" % (what, dt / 300 * 1e6), flush=True)
