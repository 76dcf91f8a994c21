#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace database (rocpd): how much of the step is launch gaps.
usage: gap_analysis.py results.db [lo] [hi]  -- analyses the kernels between the fractions lo and hi of the trace (steady-state steps)."""
import sqlite3, sys
db = sys.argv[1]; skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5; hi_f = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
rows = rows[int(len(rows) * skip):int(len(rows) * hi_f)]
busy = sum(e - s for _, s, e, _ in rows)
span = rows[-1][2] - rows[0][1]
gaps = [max(0, rows[i + 1][1] - rows[i][2]) for i in range(len(rows) - 1)]
print("kernels %d  span %.3f ms  busy %.3f ms (%.1f%%)  idle %.3f ms" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 64), (64, 1e9)):
    sel = [g for g in gaps if lo * 1e3 <= g < hi * 1e3]
    print("  gaps %4g-%-6g us: %6d  total %.3f ms" % (lo, hi, len(sel), sum(sel) / 1e6))
# which kernels are followed by the largest total idle time
from collections import defaultdict
after = defaultdict(lambda: [0, 0.0])
for i, g in enumerate(gaps):
    k = rows[i][0][:70]
    after[k][0] += 1; after[k][1] += g
print("idle time by preceding kernel (top 15):")
for k, (n, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:15]:
    print("  %-72s n %6d  idle %.3f ms  avg %.2f us" % (k, n, t / 1e6, t / n / 1e3))
steps = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
agg = defaultdict(lambda: [0, 0.0])
for n, s, e, g in rows:
    k = n.replace("(anonymous namespace)::", "").replace("void ", "")[:80]
    agg[k][0] += 1; agg[k][1] += e - s
print("per-kernel time in the window (per step over %g steps):" % steps)
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-82s n/step %6.1f  us/step %8.1f  avg %7.2f us" % (k, n / steps, t / 1e3 / steps, t / n / 1e3))
