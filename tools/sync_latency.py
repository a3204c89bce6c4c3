"""Host-side latency around one multi-tick launch (tuning; GPU): wall time from the call of DeviceWorlds.run(n) to the return of the
synchronise behind it, for three ways of waiting -- torch.cuda.synchronize(), a busy poll of a HIP event, hipStreamSynchronize -- and
whatever runtime environment the caller set (HSA_ENABLE_INTERRUPT=0 makes the runtime poll its completion signals).
   python tools/sync_latency.py [n_ticks ...]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ticks = [int(x) for x in sys.argv[1:]] or [1, 20]
args = argparse.Namespace(worlds=256, workload="c4", seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
dw.run(600, 70, 100)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
stream = torch.cuda.current_stream()


def wait_sync():
    torch.cuda.synchronize()


def wait_poll():
    ev1.record()
    while not ev1.query():
        pass
    torch.cuda.synchronize()


def wait_stream():
    stream.synchronize()
    torch.cuda.synchronize()


print("env: HSA_ENABLE_INTERRUPT=%s" % os.environ.get("HSA_ENABLE_INTERRUPT"))
for n in ticks:
    for name, wait in (("synchronize", wait_sync), ("event poll", wait_poll), ("stream sync", wait_stream)):
        best, tot, dev = 1e9, 0.0, 0.0
        reps = 30
        for r in range(reps + 5):
            torch.cuda.synchronize()
            time.sleep(0.0005)            # the idle gap a host loop leaves
            t0 = time.perf_counter()
            ev0.record()
            dw.run(n, 70, 100)
            if wait is not wait_poll:
                ev1.record()
            wait()
            dt = time.perf_counter() - t0
            if r >= 5:
                best = min(best, dt); tot += dt; dev += ev0.elapsed_time(ev1) * 1e-3
        print("n_ticks %4d  %-12s wall mean %7.1f us  min %7.1f us   events %7.1f us" % (n, name, tot / reps * 1e6, best * 1e6, dev / reps * 1e6), flush=True)
