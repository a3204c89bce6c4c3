"""A/B of library builds on the driver's window: wall time of run(20) + synchronize after run(5) + synchronize (what `bench.py --steps 20
--warmup 5` times), alternating builds, plus the steady per-tick time of a 2000-tick launch (tuning; GPU).
   python tools/short_ab.py libA.so libB.so ... [rounds=3]"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, bench
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=20260928)
a = bench.make_worlds(args, 0, "cuda:0")
a.run(300, 70, 100); torch.cuda.synchronize()
ts = []
for _ in range(41):
    a.run(5, 70, 100); torch.cuda.synchronize()
    a.acted_total.zero_(); a.refill_count.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter(); a.run(20, 70, 100); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
a.run(600, 70, 100); torch.cuda.synchronize()
t0 = time.perf_counter(); a.run(2000, 70, 100); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("20-tick window: median %%.1f us (min %%.1f, p90 %%.1f)   steady %%.2f us/tick" %% (ts[20] * 1e6, ts[0] * 1e6, ts[36] * 1e6, dt / 2000 * 1e6))
''' % root
libs = [x for x in sys.argv[1:] if x.endswith(".so")]
rounds = [int(x) for x in sys.argv[1:] if x.isdigit()]
for r in range(rounds[0] if rounds else 3):
    for lib in libs:
        env = dict(os.environ, REINLIFE_HIP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("%-40s %s" % (os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]), flush=True)
