#!/usr/bin/env python3
"""profiles/<tag>_pmc_gemm_nt.json (what bench.py reads for `roofline.traffic`) from the two per-kernel summaries
tools/pmc_summary.py printed for the FETCH_SIZE and the WRITE_SIZE pass over tools/pmc_gemm.py.
    python tools/pmc_gemm_json.py <tag>_pmc_gemm_FETCH_SIZE.txt <tag>_pmc_gemm_WRITE_SIZE.txt <tag> > <tag>_pmc_gemm_nt.json"""
import json
import re
import sys

KERNELS = {
    "gemm_nt_wide_kernel<1, 0, 0>": ("gemm_nt D.fc2.0 M=65536 N=1024 K=256 (affine prologue + statistics + pooling partials, output not stored)",
                                     80740352, "A 67.1 MB + W 1.05 MB read once = 68.2 MB; own statistics/pooling records 12.6 MB written"),
    "gemm_nt_kernel<0, 0, 0, 1, 1, 0, 0>": ("gemm_nt conv_out M=65536 N=128 K=1280", 369754112, "A 335.5 MB + W 0.66 MB read, Y 33.6 MB written"),
}


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"void (.+?)\(spgan_gemm_nt_args\)\s+(\w+)\s+launches\s+(\d+)\s+avg\s+([\d.]+)", line.strip())
        if m and m.group(2) != "duration_ns":      # (pmc_summary.py also prints the kernels' average durations: not a counter)
            out[m.group(1)] = (int(m.group(3)), float(m.group(4)))
    return out


def main():
    fetch, write, tag = parse(sys.argv[1]), parse(sys.argv[2]), sys.argv[3]
    res = {"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace / rocprofv3 --pmc WRITE_SIZE --kernel-trace (separate passes) -- python tools/pmc_gemm.py, "
                     "MI355X (profiles/%s_pmc_gemm_FETCH_SIZE.txt, %s_pmc_gemm_WRITE_SIZE.txt: the per-kernel averages tools/pmc_summary.py printed)" % (tag, tag),
           "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B for wide coalesced streams "
                    "(MI355X_MICROARCH.md, HBM section) -> doubled below; WRITE_SIZE used as reported",
           "kernels": {}}
    for sym, (key, algo, note) in KERNELS.items():
        if sym in fetch and sym in write:
            hbm = int(round((2.0 * fetch[sym][1] + write[sym][1]) * 1024))
            res["kernels"][key] = {"kernel_symbol": sym, "FETCH_SIZE_KiB_avg_of_%d" % fetch[sym][0]: fetch[sym][1],
                                   "WRITE_SIZE_KiB_avg_of_%d" % write[sym][0]: write[sym][1], "hbm_bytes_per_launch_corrected": hbm,
                                   "algorithmic_bytes_per_launch": algo, "ratio": round(hbm / algo, 3), "note": note}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
