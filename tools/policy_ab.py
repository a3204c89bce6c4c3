"""Dense PERD3QN policy launches at several tile counts, for A/B of library builds (REINLIFE_HIP_LIB=...)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reinlife_amd import _lib
from reinlife_amd.worlds import pack_brain_weights, policy_forward
kind = _lib.PERD3QN
packed = pack_brain_weights(kind, bench.brain_weights("PERD3QN", 1))
res = []
for tiles in (2, 64, 256, 680, 1024, 2048, 4096, 10880):
    n = tiles * 32
    obs = torch.randn(n + 1, 153, device="cuda:0")[:n]
    out = torch.empty(n, 8, device="cuda:0")
    for _ in range(5):
        policy_forward(kind, packed, obs, out)
    evs = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); policy_forward(kind, packed, obs, out); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    res.append("%d: %.1f" % (tiles, float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3))
print(os.environ.get("TAG", ""), "tiles: us  ", "  ".join(res))
