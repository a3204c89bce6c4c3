"""A/B of two builds of the library on the same box: rl_run per-tick time at 256 worlds, alternating, N rounds (tuning; GPU).
   python tools/run_ab.py libA.so libB.so [rounds]"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, bench
args = __import__("argparse").Namespace(worlds=256, workload=os.environ.get("RL_AB_WORKLOAD", "c4"), seed=1)
a = bench.make_worlds(args, 0, "cuda:0")
kw = {}
if os.environ.get("RL_AB_TRAIN"):   # the TRAIN 1 instantiation (what trainer() launches): Tracker accumulators + a per-tick epsilon schedule
    import numpy as np
    a.enable_tracking(True)
    kw = dict(eps_schedule=np.full((2000, len(bench.WORKLOADS[args.workload]["brains"])), 0.05, np.float32), trk_skip=1)
a.run(600, 70, 100); torch.cuda.synchronize()
before = int(a.acted_total.item()); t0 = time.perf_counter(); a.run(2000, 70, 100, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%%.2f us/tick %%.3e agent-steps/s" %% (dt / 2000 * 1e6, (int(a.acted_total.item()) - before) / dt))
''' % root
libs = sys.argv[1:3]
for r in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    for lib in libs:
        env = dict(os.environ, REINLIFE_HIP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("%-40s %s" % (os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]), flush=True)
