#!/usr/bin/env python3
"""Where the register allocator put its scratch traffic, per k_run instantiation (build container, no GPU).

    python tools/spillcheck.py [rl_run.hip] [filter]

Compiles the translation unit to gfx950 assembly and, per kernel matching the filter, counts MFMA
instructions, scratch loads / stores, and the scratch loads that sit within 60 instructions of an MFMA
(reloads inside the policy tile: the thing DESIGN 5.9 says to rule out before timing a change).
"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "reinlife_amd" / "csrc" / (sys.argv[1] if len(sys.argv) > 1 else "rl_run.hip")
filt = sys.argv[2] if len(sys.argv) > 2 else "k_run<512"
out = Path("/tmp/spillcheck.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                "-Wno-unused-function", "--cuda-device-only", "-S", str(src), "-o", str(out)],
               check=True, stderr=subprocess.DEVNULL)
funcs, cur = {}, None
for line in out.read_text().split("\n"):
    m = re.match(r"^(_Z\S+):", line)
    if m:
        cur = m.group(1)
        funcs[cur] = []
    elif line.startswith(".Lfunc_end"):
        cur = None
    elif cur:
        funcs[cur].append(line)
names = list(funcs)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for name, d in zip(names, dem):
    if filt not in d:
        continue
    lines = funcs[name]
    mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
    sl = [i for i, l in enumerate(lines) if "scratch_load" in l]
    ss = [i for i, l in enumerate(lines) if "scratch_store" in l]
    near = [i for i in sl if any(abs(i - j) < 60 for j in mf)]
    short = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0]
    print(f"{short:40s} mfma {len(mf):4d}  scratch_load {len(sl):3d}  scratch_store {len(ss):3d}  loads near an MFMA {len(near)}")
