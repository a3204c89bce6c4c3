#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace [+ pmc]) as text: per-kernel calls / total / avg / min /
max duration, registers, LDS; and per-kernel mean PMC counter values when present.

    python tools/rocpd_summary.py <results.db> [--skip-first N]
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x, "
                       "dispatch_id from kernels order by start").fetchall()
    stats = {}
    for name, dur, vg, ag, sg, lds, gx, wx, did in rows:
        s = stats.setdefault(name, dict(d=[], vg=vg, ag=ag, sg=sg, lds=lds, gx=gx, wx=wx))
        s["d"].append(dur)
    total = sum(sum(s["d"][skip:]) for s in stats.values()) or 1
    print("%-72s %7s %12s %10s %10s %10s %6s %5s %5s %5s %7s %9s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct",
                                                                   "vgpr", "agpr", "sgpr", "lds_B", "grid/wg"))
    for name, s in sorted(stats.items(), key=lambda kv: -sum(kv[1]["d"])):
        d = s["d"][skip:] or s["d"]
        short = name if len(name) <= 72 else name[:69] + "..."
        print("%-72s %7d %12.1f %10.2f %10.2f %10.2f %6.1f %5s %5s %5s %7s %9s" % (
            short, len(d), sum(d) / 1e3, sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, 100.0 * sum(d) / total,
            s["vg"], s["ag"], s["sg"], s["lds"], "%d/%d" % (s["gx"] // max(1, s["wx"]), s["wx"])))
    for name, s in sorted(stats.items(), key=lambda kv: -sum(kv[1]["d"]))[:3]:   # the few calls of the big kernels, one by one (start order)
        d = s["d"][skip:] or s["d"]
        if 1 < len(d) <= 12:
            print("calls of %s in start order (us): %s" % (name if len(name) <= 72 else name[:69] + "...", "  ".join("%.2f" % (x / 1e3) for x in d)))
    try:
        pmc = cur.execute("select k.name, p.counter_name, avg(p.counter_value), count(*) from pmc_events p join kernels k "
                          "on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name").fetchall()
    except sqlite3.Error:
        try:
            pmc = cur.execute("select name, counter_name, avg(value), count(*) from counters_collection group by name, counter_name").fetchall()
        except sqlite3.Error:
            pmc = []
    if pmc:
        print("\nPMC counters (mean per dispatch)")
        for name, cname, val, n in pmc:
            short = name if len(name) <= 72 else name[:69] + "..."
            print("%-72s %-28s %16.1f  (n=%d)" % (short, cname, val, n))


if __name__ == "__main__":
    main()
