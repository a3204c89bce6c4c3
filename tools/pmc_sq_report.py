"""Per-kernel averages of the counters tools/pmc_sq.sh collected (rocpd sqlite), with a few derived ratios."""
import glob
import os
import sqlite3
import sys

out = sys.argv[1]
per = {}     # kernel -> counter -> (mean per dispatch, n)
dur = {}
for db in glob.glob(os.path.join(out, "*", "*", "*_results.db")):
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
    if "counters_collection" not in tabs:
        continue
    cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    valcol = "value" if "value" in cols else "counter_value"
    for name, c, v, n in cur.execute("select %s, counter_name, avg(%s), count(*) from counters_collection group by %s, counter_name" % (namecol, valcol, namecol)):
        per.setdefault(name, {})[c] = (v, n)
    if "kernels" in tabs:
        kc = [r[1] for r in cur.execute("pragma table_info('kernels')")]
        if "duration" in kc:
            for name, d, n in cur.execute("select name, avg(duration), count(*) from kernels group by name"):
                dur.setdefault(name, []).append((d, n))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:70]


for name in sorted(per):
    if not ("k_policy" in name or "k_world" in name or "k_bucket" in name or "k_run" in name):
        continue
    c = {k: v[0] for k, v in per[name].items()}
    print("== %s   (dispatches per pass: %d)" % (short(name), max(v[1] for v in per[name].values())))
    if name in dur:
        print("   avg duration under the profiler: %.2f us" % (sum(d for d, _ in dur[name]) / len(dur[name]) / 1e3))
    for k in sorted(c):
        print("   %-34s %16.0f" % (k, c[k]))
    g = c.get
    wc = g("SQ_WAVE_CYCLES")
    if wc:
        print("   -- derived (SQ_* cycle counters are quad-cycles summed over waves; MFMA_BUSY in cycles summed over SIMDs)")
        for lab, key in (("wave parked (waitcnt/barrier)", "SQ_WAIT_ANY"), ("issue stall", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY")):
            if g(key) is not None:
                print("   %-34s %15.1f %%" % (lab + " / wave cycles", 100.0 * g(key) / wc))
        if g("SQ_WAVES"):
            print("   %-34s %16.0f" % ("wave cycles (x4) per wave", 4.0 * wc / g("SQ_WAVES")))
    if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        # SQ_BUSY_CYCLES: per-SE busy; MFMA busy cycles summed over SIMDs -> divide by SIMDs (1024) x kernel cycles (GRBM_GUI_ACTIVE)
        ga = g("GRBM_GUI_ACTIVE")
        if ga:
            ga = ga / 8.0   # summed over the 8 XCDs
            print("   %-34s %15.1f %%" % ("MFMA pipe busy (of 1024 SIMDs x GUI_ACTIVE/8)", 100.0 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * ga)))
            if g("SQ_ACTIVE_INST_VALU") is not None:
                print("   %-34s %15.1f %%" % ("VALU issuing (x4 / 1024 SIMDs x GUI_ACTIVE/8)", 100.0 * 4.0 * g("SQ_ACTIVE_INST_VALU") / (1024.0 * ga)))
            if g("SQ_LEVEL_WAVES") is not None:
                print("   %-34s %16.2f" % ("mean resident waves per SIMD", g("SQ_LEVEL_WAVES") / ga / 1024.0))
    if g("SQ_LDS_IDX_ACTIVE"):
        print("   %-34s %15.1f %%" % ("LDS bank-conflict cycles / LDS active", 100.0 * (g("SQ_LDS_BANK_CONFLICT") or 0) / g("SQ_LDS_IDX_ACTIVE")))
    tot = sum(g(k) or 0 for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_FLAT"))
    if tot and g("SQ_WAVES"):
        print("   %-34s %16.0f  (VALU %d incl. MFMA %d, SALU %d, SMEM %d, VMEM %d, LDS %d)" % (
            "instructions per wave", tot / g("SQ_WAVES"), (g("SQ_INSTS_VALU") or 0) / g("SQ_WAVES"), (g("SQ_INSTS_MFMA") or 0) / g("SQ_WAVES"),
            (g("SQ_INSTS_SALU") or 0) / g("SQ_WAVES"), (g("SQ_INSTS_SMEM") or 0) / g("SQ_WAVES"), (g("SQ_INSTS_VMEM") or 0) / g("SQ_WAVES"),
            (g("SQ_INSTS_LDS") or 0) / g("SQ_WAVES")))
    print()
