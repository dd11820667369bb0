#!/usr/bin/env python
"""Dump (start_us, end_us, stream/queue if present, kernel) of every dispatch in a rocprofv3 rocpd database, relative to
the first one: `rocpd_timeline.py results.db > timeline.txt`.  For reading overlap between streams off a trace."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
want = [c for c in ("start", "end", "queue_id", "stream_id", "name") if c in cols]
rows = list(db.execute("select %s from kernels order by start" % ", ".join(want)))
t0 = rows[0][0]
print("# columns:", ", ".join(want), " (start / end in microseconds from the first dispatch)")
for r in rows:
    d = dict(zip(want, r))
    print("%10.1f %10.1f %s %s" % ((d["start"] - t0) / 1e3, (d["end"] - t0) / 1e3,
                                   " ".join(str(d[k]) for k in ("queue_id", "stream_id") if k in d), d["name"].split("(")[0][:60]))
