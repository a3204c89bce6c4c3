"""rl_run per-tick time with one half switched off (rl_debug_set_run_mask, tuning only), per workgroup size (tuning aid; GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=1)
for blk in sys.argv[1:] or ("512", "256", "1024"):
    os.environ["RL_WORLD_BLOCK"] = blk
    for dbg in ("0", "1", "2"):
        __import__("reinlife_amd._lib", fromlist=["lib"]).lib().rl_debug_set_run_mask(int(dbg))
        a = bench.make_worlds(args, 0, "cuda:0")
        a.run(300 if dbg == "0" else 5, 70, 100); torch.cuda.synchronize()
        N = 300
        before = int(a.acted_total.item())
        t0 = time.perf_counter(); a.run(N, 70, 100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("block %s rl_debug_set_run_mask=%s (1 = no policy, 2 = no tick): %.2f us/tick  %.3e agent-steps/s" % (blk, dbg, dt / N * 1e6, (int(a.acted_total.item()) - before) / dt), flush=True)
