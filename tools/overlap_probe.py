"""Experiment: G groups of worlds on G streams, out of phase, so one group's policy launch overlaps another's tick.
   python tools/overlap_probe.py [total_worlds] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from types import SimpleNamespace

def run(total, G, steps, phase_shift=True):
    dev = torch.device("cuda:0")
    groups, streams = [], []
    for g in range(G):
        args = SimpleNamespace(worlds=total // G, workload="c4", seed=20260928)
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            dw = bench.make_worlds(args, g, dev)
        groups.append(dw); streams.append(s)
    torch.cuda.synchronize()
    def step():
        if G == 1:
            with torch.cuda.stream(streams[0]):
                groups[0].act(); groups[0].tick_refill(70, 100)
            return
        if phase_shift:   # slot 1: act(even) | tick(odd) ; slot 2: tick(even) | act(odd)
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    (groups[g].act() if g % 2 == 0 else groups[g].tick_refill(70, 100))
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    (groups[g].tick_refill(70, 100) if g % 2 == 0 else groups[g].act())
        else:
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    groups[g].act(); groups[g].tick_refill(70, 100)
    for g in range(G):   # odd groups need actions before their first tick
        with torch.cuda.stream(streams[g]):
            groups[g].act()
    for _ in range(30): step()
    torch.cuda.synchronize()
    a0 = sum(int(d.acted_total.item()) for d in groups)
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    a1 = sum(int(d.acted_total.item()) for d in groups)
    print("worlds %5d groups %d shift %d : %.1f us/step  %.3e agent-steps/s" % (total, G, phase_shift, dt / steps * 1e6, (a1 - a0) / dt), flush=True)

total = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for G, sh in ((1, True), (2, True), (2, False), (4, True), (4, False)):
    run(total, G, steps, sh)
