#!/usr/bin/env python3
"""Average PMC counter value per kernel symbol from a rocprofv3 --pmc rocpd database.
usage: pmc_summary.py results.db [filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else "spgan"
for name, ctr, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
    if flt in name:
        print("%-100s %-12s launches %3d  avg %14.2f" % (name.replace("(anonymous namespace)::", "")[:100], ctr, n, avg))
try:      # kernel durations of the same (profiled) run, for clock estimates: cycles / duration
    for name, n, avg in c.execute("select name, count(*), avg(duration) from kernels group by name"):
        if flt in name:
            print("%-100s %-12s launches %3d  avg %14.2f" % (name.replace("(anonymous namespace)::", "")[:100], "duration_ns", n, avg))
except sqlite3.Error:
    pass
