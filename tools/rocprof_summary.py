#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output (ROCm 7.2 default) into the text summaries kept under
profiles/.   usage: rocprof_summary.py stats <results.db>   |   pmc <results.db> [...]
|   timeline <results.db> <anchor kernel substring>   (one steady-state period between two launches of the anchor)"""
import json
import sqlite3
import sys


def stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name "
                     "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# kernel-trace stats from {db} (durations in ns)")
    print("name | calls | total_ns | avg_ns | min_ns | max_ns | pct | vgpr | lds_bytes | grid_x | wg_x")
    for r in rows:
        print(f"{r[0]} | {r[1]} | {r[2]} | {r[3]:.0f} | {r[4]} | {r[5]} | {100*r[2]/tot:.2f} | {r[6]} | {r[7]} | {r[8]} | {r[9]}")


def timeline(db, anchor):
    """The dispatches between two consecutive launches of the anchor kernel (the median-length period of the
    second half of the run): start offset, duration and the idle gap before each, in ns."""
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if anchor in r[0]]
    marks = marks[len(marks) // 2:]
    periods = sorted((rows[b][1] - rows[a][1], a, b) for a, b in zip(marks, marks[1:]))
    if not periods:
        print("no period found"); return
    plen, a, b = periods[len(periods) // 2]
    print(f"# one period of '{anchor}' from {db}: {plen} ns, {b - a} dispatches (median of {len(periods)} periods; "
          f"min {periods[0][0]}, max {periods[-1][0]})")
    print("offset_ns | dur_ns | gap_before_ns | name")
    t0, prev_end, busy = rows[a][1], None, 0
    for r in rows[a:b]:
        gap = 0 if prev_end is None else r[1] - prev_end
        print(f"{r[1] - t0} | {r[2] - r[1]} | {gap} | {r[0][:100]}")
        prev_end = r[2]; busy += r[2] - r[1]
    print(f"# busy {busy} ns of {plen} ns ({100 * busy / plen:.1f} %)")


def pmc(dbs):
    out = {}
    for db in dbs:
        c = sqlite3.connect(db)
        rows = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                         "from counters_collection group by kernel_name, counter_name").fetchall()
        print(f"# PMC per-dispatch values from {db}")
        print("kernel | counter | dispatches | avg | min | max")
        for r in rows:
            print(f"{r[0]} | {r[1]} | {r[2]} | {r[3]:.3f} | {r[4]:.3f} | {r[5]:.3f}")
            out.setdefault(r[0], {})[r[1]] = {"avg": r[3], "min": r[4], "max": r[5], "n": r[2]}
    return out


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3])
    else:
        res = pmc(sys.argv[2:])
        print("# json")
        print(json.dumps(res))
