"""rl_run per-tick time with the scheduled policy tile cut short after each stage (RL_RUN_DEBUG bits 32 / 64 / 128; results WRONG; tuning; GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=1)
for dbg, what in (("1", "no policy"), ("32", "row read, scale, first split"), ("64", "+ input layer, epilogue, first B2 split"), ("128", "+ hidden layer, epilogue"), ("0", "whole tile")):
    os.environ["RL_RUN_DEBUG"] = dbg
    a = bench.make_worlds(args, 0, "cuda:0")
    a.run(300, 70, 100); torch.cuda.synchronize()
    t0 = time.perf_counter(); a.run(300, 70, 100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-45s %.2f us/tick" % (what, dt / 300 * 1e6), flush=True)
