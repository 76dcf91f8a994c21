#!/usr/bin/env python3
"""Ordered kernel sequence of ONE replayed train step from a rocprofv3 --kernel-trace database: name, duration, gap to the
previous kernel -- to see the dependency chains of small launches.  usage: step_sequence.py results.db [which_step]"""
import re, sqlite3, sys
db = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
adam = [i for i, r in enumerate(rows) if "adam_dev_kernel" in r[0]]
ends = adam[1::2]                                   # the generator's Adam closes a step
lo, hi = ends[which - 1] + 1, ends[which] + 1
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:70]
tot = 0.0
print("# %d kernels, step span %.3f ms" % (hi - lo, (rows[hi - 1][2] - rows[lo][1]) / 1e6))
for i in range(lo, hi):
    n, s, e, g = rows[i]
    gap = (s - rows[i - 1][2]) / 1e3
    tot += (e - s) / 1e3
    print("%4d %-72s grid %8d  %8.2f us  gap %6.2f  cum %9.1f" % (i - lo, short(n), g, (e - s) / 1e3, gap, tot))
