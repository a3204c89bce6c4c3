"""Host-side issue cost of one bench step against its GPU time (tuning aid, GPU only): the Python loop enqueues a step in
~13 us, the GPU takes ~35 us, so the loop stays ahead of the device."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
args = argparse.Namespace(worlds=256, workload="c4", seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
for _ in range(50): bench.one_step(dw)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(300): bench.one_step(dw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host issue %.1f us/step, total %.1f us/step" % ((t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
