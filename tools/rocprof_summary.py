#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into a per-kernel table (text).
usage: rocprof_summary.py results.db [steps] [ntop] > profiles/xxx.txt
The ntop (default 3) most expensive kernels are additionally broken down by launch grid (= by problem shape)."""
import re, sqlite3, sys

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n) if len(n) > 90 else n
    return n[:110]

db = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary; durations in microseconds; %d kernels; total %.1f us (%.3f ms per step over %g steps)" % (len(rows), tot, tot / steps / 1e3, steps))
print("%-112s %8s %12s %10s %7s %12s" % ("kernel", "calls", "total_us", "avg_us", "pct", "us_per_step"))
for n, calls, td, avg, pct in rows:
    print("%-112s %8d %12.1f %10.2f %6.2f%% %12.1f" % (short(n), calls, td, avg, pct, td / steps))

ntop = int(sys.argv[3]) if len(sys.argv) > 3 else 3
print()
print("# per-grid breakdown of the %d most expensive kernels (a kernel symbol serves several problem shapes; grid = workgroups x 256 threads)" % ntop)
for n, calls, td, avg, pct in rows[:ntop]:
    print("%s" % short(n))
    for gx, cnt, tot, av in c.execute("select grid_x, count(*), sum(duration), avg(duration) from kernels where name = ? group by grid_x order by sum(duration) desc", (n,)):
        print("    grid_x %9d  calls %6d  total_us %12.1f  avg_us %10.2f  us_per_step %10.1f" % (gx, cnt, tot / 1e3, av / 1e3, tot / 1e3 / steps))
