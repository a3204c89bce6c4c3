#!/usr/bin/env python
"""Shader-clock stamps of the TICK half of the multi-tick launch (rl_run), workgroup `world`, thread 0, last tick of a launch (tuning; GPU; prof build)."""
import ctypes as C, os, sys
os.environ["RL_PHASE_PROFILE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reinlife_amd import _lib
NAMES = ["policy half end -> tick entry (params, carve)", "Environment.step: act, attack, conflicts, eat/move/death, rewards, food", "order 1 + planes 1 (+ barrier)",
         "reproduce (wave 0) || obs1 rows + step outputs (others)", "src, planes patch, agent bitmap + prefix (last wave), hash clear (+ barrier)", "(scan order 2: merged into the interval before, round 4)", "refill or order 2 + gene hash", "planes 2 (+ barrier)",
         "policy lists (wave 0) || obs2 rows -> memory + LDS mirror", "recycle world (compaction, clears, 2 barriers)"]
IDX = [112, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69]
SUB = [60, 2, 3, 35, 36, 37, 4, 5, 6, 61]   # inside Environment.step
SUBN = ["act + attack + prep", "conflict loop (+ the loads of eat / vanish)", "precompute draws (round 4: no barrier here)", "(no interval: merged)", "commit: eat + clear + place + death + hash", "(mark 4)", "rewards", "food count + bitmap", "food placement .. end"]
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args = __import__("argparse").Namespace(worlds=R, workload=os.environ.get("RL_AB_WORKLOAD", "c4"), seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
stamps = torch.zeros(128, dtype=torch.int64, device="cuda:0")
lib = _lib.lib()
if os.environ.get("RL_PROF_TRACK"):
    dw.enable_tracking(True)   # the TRAIN instantiation with the Tracker pass on wave 1
dw.run(50, 70, 100)
if os.environ.get("RL_PROF_ABLATE"):   # tuning: skip sections of the tick in the stamped launches (results wrong; rl_world_dev.h RL_ABL bits)
    lib.rl_debug_set_ablate.argtypes = [C.c_int]
    lib.rl_debug_set_ablate(int(os.environ["RL_PROF_ABLATE"]))
acc = []
iv = []
sub = []
sub2 = []
SUB2 = [62, 14, 41, 42, 63]   # inside the reproduce interval (wave 0)
for t in range(40):
    _lib.check(lib.rl_bind_phase_profile(dw.handle, C.c_void_p(stamps.data_ptr()), (7 * t) % R), "bind")
    stamps.zero_()
    dw.run(int(os.environ.get("RL_PROFILE_TICKS", "0")) or (20 + t % 5), 70, 100)   # the stamps of the LAST tick remain (RL_PROFILE_TICKS=1: a launch's FIRST tick)
    torch.cuda.synchronize()
    st = stamps.cpu().numpy()[IDX]
    raw_all = stamps.cpu().numpy().astype(np.float64)
    if raw_all[96] and raw_all[67]:
        iv.append([raw_all[96] - raw_all[67], raw_all[97] - raw_all[96], raw_all[98] - raw_all[67], raw_all[99] - raw_all[67], raw_all[68] - raw_all[67]])
    if st.all() and (np.diff(st) > 0).all():
        acc.append(np.diff(st))
        sub.append(np.diff(stamps.cpu().numpy()[SUB]))
        r2 = stamps.cpu().numpy()[SUB2]
        if r2.all(): sub2.append(np.diff(r2))
m = np.mean(acc, axis=0)
print("rl_run tick half, thread 0 of the sampled world, mean of %d launches, total %.0f cycles" % (len(acc), m.sum()))
for n, v in zip(NAMES, m):
    print("   %-70s %8.0f  %5.1f%%" % (n, v, 100 * v / m.sum()))
ms = np.mean(sub, axis=0)
print("inside Environment.step:")
for n, v in zip(SUBN, ms):
    print("      %-60s %8.0f" % (n, v))
if sub2:
    print("inside reproduce (wave 0): gates + parents | placements | produce | newborns, dead -> food, barrier:", np.mean(sub2, axis=0).round(0))

if iv:
    print("the lists || obs2 interval (cycles from its start): wave 0 lists done %.0f, + schedule %.0f | wave 1 rows done %.0f, last wave rows done %.0f | thread 0 leaves %.0f" % tuple(np.mean(iv, axis=0)))
