#!/usr/bin/env python3
"""Whole-step HBM traffic and time-weighted MFMA utilisation from rocprofv3 --pmc passes over tools/pmc_step.py.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d f -o r -- python tools/pmc_step.py      (separate passes: the counters do not fit one)
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d w -o r -- python tools/pmc_step.py
    rocprofv3 --pmc MfmaUtil   --kernel-trace -d m -o r -- python tools/pmc_step.py
    python tools/pmc_step_total.py f/…/r_results.db w/…/r_results.db m/…/r_results.db > profiles/r02_pmc_step.json

FETCH_SIZE is doubled (MI355X_MICROARCH.md: on gfx950 it reports half the bytes of wide coalesced reads); units are KB as rocprofv3
reports them.  Steps are counted by the optimiser kernels (two Adam launches per step)."""
import json, re, sqlite3, sys
from collections import defaultdict


def load(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, value, grid_size, end - start from counters_collection where counter_name = ?", (counter,)).fetchall()
    try:
        dur = {r[0]: (r[1], r[2]) for r in c.execute("select name, sum(end - start), count(*) from kernels group by name")}
    except sqlite3.Error:
        dur = {}
    return rows, dur


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:80]


def main():
    fdb, wdb, mdb = sys.argv[1:4]
    fetch, fdur = load(fdb, "FETCH_SIZE")
    write, _ = load(wdb, "WRITE_SIZE")
    mfma, mdur = load(mdb, "MfmaUtil")
    steps = sum(1 for r in fetch if "adam" in r[0] and "prep" not in r[0]) / 2.0
    per = defaultdict(lambda: [0, 0.0, 0.0])
    for n, v, g, _ in fetch:
        d = per[short(n)]; d[0] += 1; d[1] += 2.0 * v * 1024.0          # KB -> bytes, gfx950 x2
    for n, v, g, _ in write:
        per[short(n)][2] += v * 1024.0
    tot_f = sum(v[1] for v in per.values()) / steps
    tot_w = sum(v[2] for v in per.values()) / steps
    kern = []
    for k, (cnt, f, w) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        ns, calls = fdur.get(next((n for n in fdur if short(n) == k), None), (0, 0))
        kern.append({"kernel": k, "launches_per_step": round(cnt / steps, 1), "fetch_MB_per_step": round(f / steps / 1e6, 1),
                     "write_MB_per_step": round(w / steps / 1e6, 1),
                     "TB_per_s": round((f + w) / max(ns, 1) / 1e3, 2) if ns else None})
    # MfmaUtil weighted by the duration of every single dispatch, over the kernels that use the matrix cores
    mu = defaultdict(lambda: [0.0, 0.0, 0])
    for n, v, g, ns in mfma:
        k = short(n)
        if any(t in k for t in ("gemm_nt_kernel", "gemm_nt_wide_kernel", "gemm_tn_kernel", "knn_mfma")):
            mu[k][0] += v * ns; mu[k][1] += ns; mu[k][2] += 1
    tw_num = sum(v[0] for v in mu.values()); tw_den = sum(v[1] for v in mu.values())
    mlist = [{"kernel": k, "MfmaUtil_pct_time_weighted": round(v[0] / max(v[1], 1), 1), "ms_per_step": round(v[1] / steps / 1e6, 3),
              "launches_per_step": round(v[2] / steps, 1)} for k, v in mu.items()]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / MfmaUtil (three separate passes, each with --kernel-trace) -- python tools/pmc_step.py; "
                     "FETCH_SIZE doubled per the gfx950 correction; %g eager WGAN-GP train steps (B=32, N=2048) in the trace" % steps,
           "hbm_bytes_per_step": {"fetch_GB": round(tot_f / 1e9, 3), "write_GB": round(tot_w / 1e9, 3), "total_GB": round((tot_f + tot_w) / 1e9, 3),
                                  "algorithmic_GB_SURVEY_8d": 3.5, "ratio_to_algorithmic": round((tot_f + tot_w) / 3.5e9, 2)},
           "mfma_util_time_weighted_pct": round(tw_num / max(tw_den, 1), 1),
           "mfma_kernel_ms_per_step_under_pmc": round(tw_den / steps / 1e6, 3),
           "mfma_kernels": sorted(mlist, key=lambda r: -r["ms_per_step"]),
           "kernels_by_traffic": kern[:40]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
