#!/usr/bin/env python3
"""Kernels of a rocprofv3 --kernel-trace database that run long on few workgroups (latency-bound launches that leave most of the
256 CUs idle).  usage: low_parallelism.py results.db [steps]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
wg = "workgroup_x" if "workgroup_x" in cols else None
q = "select name, grid_x, %s, count(*), avg(duration), sum(duration) from kernels group by name, grid_x order by sum(duration) desc" % (wg or "256")
print("%-70s %10s %6s %8s %9s %10s" % ("kernel", "grid_thr", "wgs", "n/step", "avg_us", "us/step"))
for name, gx, wgx, n, avg, tot in c.execute(q):
    wgs = gx // max(int(wgx or 256), 1)
    if wgs <= 1024 and avg / 1e3 >= 7.0:
        print("%-70s %10d %6d %8.1f %9.2f %10.1f" % (name.replace("(anonymous namespace)::", "").replace("void ", "")[:70], gx, wgs, n / steps, avg / 1e3, tot / 1e3 / steps))
