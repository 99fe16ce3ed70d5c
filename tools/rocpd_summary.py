#!/usr/bin/env python3
"""Dump the kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV text."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
print("Name,Calls,TotalDurationUs,AverageUs,Percentage")
for name, calls, tot, avg, pct in cur.fetchall():
    print('"%s",%d,%.3f,%.3f,%.4f' % (name[:120], calls, tot, avg, pct))
