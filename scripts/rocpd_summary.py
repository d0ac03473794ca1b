#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs: per-kernel time stats and per-kernel PMC means.

usage: rocpd_summary.py stats.db [pmc_fetch.db [pmc_write.db]]  > profiles/rNN_rocprof_summary.md
"""
import sqlite3
import sys


def short(name):
    name = name.replace("fdjac::", "").replace("void ", "")
    return name.split("(")[0]


def stats(db):
    cur = sqlite3.connect(db).cursor()
    cur.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration), "
                "max(grid_x), max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc")
    rows = cur.fetchall()
    tot = sum(r[5] for r in rows) or 1
    print("| kernel | calls | avg us | min us | max us | total us | % | grid_x | vgpr | sgpr | lds |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %d | %.2f | %.2f | %.2f | %.1f | %.1f | %d | %d | %d | %d |" % (
            short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[5] / tot, r[6], r[7], r[8], r[9]))


def pmc(db):
    cur = sqlite3.connect(db).cursor()
    cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                "from counters_collection group by kernel_name, counter_name order by avg(value)*count(*) desc")
    print("| kernel | counter | dispatches | mean | min | max | avg us (profiled) |")
    print("|---|---|---|---|---|---|---|")
    out = {}
    for r in cur.fetchall():
        print("| %s | %s | %d | %.1f | %.1f | %.1f | %.2f |" % (short(r[0]), r[1], r[2], r[3], r[4], r[5], r[6] / 1e3))
        out[(short(r[0]), r[1])] = r[3]
    return out


if __name__ == "__main__":
    print("## kernel time (rocprofv3 --kernel-trace --stats)\n")
    stats(sys.argv[1])
    for db in sys.argv[2:]:
        print("\n## PMC (%s), values in KB as rocprofv3 reports them (uncorrected)\n" % db)
        pmc(db)
