"""Reads the rocpd databases of tools/pmc_traffic.sh, prints and stores per-kernel HBM bytes per launch."""
import glob
import json
import os
import sqlite3
import sys

out = sys.argv[1]
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = glob.glob(os.path.join(out, counter, "*", "*_results.db"))
    if not dbs:
        print("no database for", counter)
        continue
    db = sqlite3.connect(dbs[0])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
    print(counter, "columns:", cols)
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    valcol = "value" if "value" in cols else "counter_value"
    rows = cur.execute("select %s, counter_name, avg(%s), count(*), sum(%s) from counters_collection group by %s, counter_name" % (namecol, valcol, valcol, namecol)).fetchall()
    for name, cname, val, n, tot in rows:
        res.setdefault(name, {})[cname] = (val, n, tot)
calib = None
lines = []
for name, d in sorted(res.items()):
    f = d.get("FETCH_SIZE", (0, 0))
    w = d.get("WRITE_SIZE", (0, 0))
    lines.append("%-90s FETCH_SIZE(KB)=%14.1f  WRITE_SIZE(KB)=%14.1f  n=%d" % (name[:90], f[0], w[0], max(f[1], w[1])))
    if "copybuffer" in name.lower() and f[0] > 100000:
        calib = (f[2], w[2])  # totals over all copy dispatches (3 x 512 MiB + negligible small copies)
print("\n".join(lines))
summary = {"raw_mean_KB_per_dispatch": {k[:100]: {c: v[0] for c, v in d.items()} for k, d in res.items() if "k_" in k or "copyBuffer" in k}}
if calib:
    known_kb = 3 * 512 * 1024.0
    summary["calibration"] = {"copy_bytes_each_way": 3 * 512 * 1024 * 1024, "fetch_factor": known_kb / calib[0], "write_factor": known_kb / calib[1],
                              "note": "wide (16 B/lane) streaming copy; MI355X_MICROARCH.md: FETCH_SIZE counts 1/2 of such reads"}
    print("calibration (512 MiB device copy): FETCH_SIZE x%.3f, WRITE_SIZE x%.3f give true bytes" % (known_kb / calib[0], known_kb / calib[1]))
for name, d in res.items():
    if "k_world" in name:
        ff = summary.get("calibration", {}).get("fetch_factor", 2.0)
        wf = summary.get("calibration", {}).get("write_factor", 1.0)
        rd = d.get("FETCH_SIZE", (0, 0))[0] * 1024 * ff
        wr = d.get("WRITE_SIZE", (0, 0))[0] * 1024 * wf
        summary["tick_kernel"] = name
        summary["hbm_read_bytes_per_launch"] = rd
        summary["hbm_write_bytes_per_launch"] = wr
        summary["hbm_bytes_per_launch"] = rd + wr
        print("tick kernel: read %.2f MB + write %.2f MB = %.2f MB per launch (calibrated)" % (rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
for name, d in res.items():
    if "k_policy" in name:
        ff = summary.get("calibration", {}).get("fetch_factor", 2.0)
        summary["policy_kernel"] = name
        summary["policy_hbm_bytes_per_launch"] = d.get("FETCH_SIZE", (0, 0))[0] * 1024 * ff + d.get("WRITE_SIZE", (0, 0))[0] * 1024
ticks_per_launch = 1
for log in ("FETCH_SIZE.log", "WRITE_SIZE.log"):
    try:
        for line in open(os.path.join(out, log)):
            if line.startswith("agent_steps_per_tick"):
                summary["agent_steps_per_launch"] = float(line.split()[1])
                summary["agent_steps_per_tick"] = float(line.split()[1])
            if line.startswith("ticks_per_launch"):
                ticks_per_launch = int(line.split()[1])
    except OSError:
        pass
for name, d in res.items():
    if "k_run" in name:   # the multi-tick launch: bytes per TICK
        ff = summary.get("calibration", {}).get("fetch_factor", 2.0)
        wf = summary.get("calibration", {}).get("write_factor", 1.0)
        rd = d.get("FETCH_SIZE", (0, 0))[0] * 1024 * ff / ticks_per_launch
        wr = d.get("WRITE_SIZE", (0, 0))[0] * 1024 * wf / ticks_per_launch
        summary["run_kernel"] = name
        summary["ticks_per_launch"] = ticks_per_launch
        summary["hbm_read_bytes_per_tick"] = rd
        summary["hbm_write_bytes_per_tick"] = wr
        summary["hbm_bytes_per_tick"] = rd + wr
        print("multi-tick launch: read %.2f MB + write %.2f MB = %.2f MB per tick (calibrated, %d ticks per launch)" % (rd / 1e6, wr / 1e6, (rd + wr) / 1e6, ticks_per_launch))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reinlife_amd import build as _build  # noqa: E402
summary["kernel_src_sha16"] = _build.source_hash()   # bench.py reports `traffic` only while the kernel sources still hash to this
json.dump(summary, open(os.path.join(out, "tick_traffic.json"), "w"), indent=1)
