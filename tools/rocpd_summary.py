#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (``*_results.db``) as text: per-kernel dispatch stats
(the equivalent of ``--stats``' kernel table) and, if the run collected PMC counters, per-kernel
counter averages.  Usage: rocpd_summary.py results.db [> profiles/xyz.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    print("# rocprofv3 kernel summary of %s" % path.split("/")[-1])
    print("# durations in microseconds")
    print("%-6s %12s %12s %12s %12s %6s %6s %8s  %s" % ("calls", "total_us", "avg_us", "min_us", "max_us", "vgpr",
                                                         "sgpr", "lds_B", "kernel"))
    q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
         "max(sgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc")
    for name, n, tot, avg, mn, mx, vg, sg, lds in db.execute(q):
        print("%-6d %12.1f %12.2f %12.2f %12.2f %6d %6d %8d  %s" % (n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, vg, sg,
                                                                     lds, name.split("(")[0]))
    try:
        rows = list(db.execute("select name, counter_name, count(*), avg(counter_value), min(counter_value), "
                               "max(counter_value), avg(duration) from pmc_events group by name, counter_name "
                               "order by name"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("\n# PMC counters per dispatch (FETCH_SIZE / WRITE_SIZE are in KiB as reported by rocprofv3)")
        print("%-14s %6s %16s %16s %16s %12s  %s" % ("counter", "n", "avg", "min", "max", "avg_us", "kernel"))
        for name, cn, n, avg, mn, mx, dur in rows:
            print("%-14s %6d %16.1f %16.1f %16.1f %12.2f  %s" % (cn, n, avg, mn, mx, dur / 1e3, name.split("(")[0]))


if __name__ == "__main__":
    main(sys.argv[1])
