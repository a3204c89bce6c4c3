import ctypes as C, os, sys
os.environ["RL_PHASE_PROFILE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reinlife_amd import _lib
args = __import__("argparse").Namespace(worlds=256, workload="c4", seed=1)
dw = bench.make_worlds(args, 0, "cuda:0")
stamps = torch.zeros(128, dtype=torch.int64, device="cuda:0")
lib = _lib.lib()
dw.run(300, 70, 100)
IDX = [100, 110, 101, 102, 103, 104, 105, 106, 107, 108, 109, 113, 114]
NAMES = ["entry->tile known", "ring prologue", "pass 1 max", "pass 2 splits (B1)", "input MFMAs", "epilogue 1 + pmax", "barrier 1", "split -> ex", "barrier 2", "hidden MFMAs", "epilogue 2 + scale", "head"]
acc = []
for t in range(60):
    _lib.check(lib.rl_bind_phase_profile(dw.handle, C.c_void_p(stamps.data_ptr()), (37 * t + 5) % 256), "bind")
    stamps.zero_(); dw.run(20, 70, 100); torch.cuda.synchronize()
    raw = stamps.cpu().numpy().astype(np.float64)
    st = raw[IDX]
    if st.all() and (np.diff(st) >= 0).all(): acc.append(np.diff(st))
print(len(acc), "samples")
if acc:
    m = np.mean(acc, axis=0)
    for n, v in zip(NAMES, m): print("  %-28s %7.0f" % (n, v))
    print("  total", m.sum())
