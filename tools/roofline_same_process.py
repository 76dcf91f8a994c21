#!/usr/bin/env python3
"""One process, one box: bench.py's own figures for the dominant kernel (device stamps inside the replayed graph; HIP events around
eagerly issued launches) next to rocprofv3's per-grid averages of the SAME run (round-4 review item 3).
usage: roofline_same_process.py bench_line.json results.db > profiles/r05_roofline_same_process.json"""
import json, sqlite3, sys

line = None
for l in open(sys.argv[1]):
    if l.startswith("{"):
        line = json.loads(l)
r = line["roofline"]
c = sqlite3.connect(sys.argv[2])
name = [n for (n,) in c.execute("select distinct name from kernels") if "gemm_nt_wide_kernel<1, 0, 0>" in n][0]
rows = {int(g): (int(n), float(a) / 1e3) for g, n, a in c.execute("select grid_x, count(*), avg(duration) from kernels where name = ? group by grid_x", (name,))}
peak = r["peak"]
out = {"ms_per_step_under_rocprof": line["ms_per_step"], "peak_tflops": peak, "shapes": {}}
fl_tot = us_tot = 0.0
for key, v in r["per_shape"].items():
    m = int(key.split("=")[1])
    grid = (m // 256) * 4 * 512
    calls, avg_us = rows.get(grid, (0, float("nan")))
    fl = 2.0 * m * 1024 * 256
    ev = r.get("per_shape_eager_events", {}).get(key, {})
    out["shapes"][key] = {"in_graph_stamps_us": round(v["avg_launch_ms"] * 1e3, 2), "in_graph_frac": v["frac"],
                          "eager_events_us": round(ev.get("avg_launch_ms", float("nan")) * 1e3, 2), "eager_events_frac": ev.get("frac"),
                          "rocprof_avg_us_all_instances": round(avg_us, 2), "rocprof_calls": calls,
                          "rocprof_frac": round(fl / (avg_us * 1e-6) / 1e12 / peak, 4),
                          "in_graph_over_rocprof": round(v["avg_launch_ms"] * 1e3 / avg_us, 4)}
    fl_tot += fl; us_tot += avg_us
out["frac_in_graph"] = r["frac"]; out["frac_eager_events"] = r.get("frac_eager_events"); out["frac_rocprof"] = round(fl_tot / (us_tot * 1e-6) / 1e12 / peak, 4)
out["note"] = ("same process under rocprofv3 --kernel-trace: rocprof averages ALL instances of a grid (replayed, priming and the 4 eager accounting "
               "steps); the stamps cover the %d timed replays only" % line["steps"])
print(json.dumps(out, indent=1))
