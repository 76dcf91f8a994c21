#!/usr/bin/env python3
"""Per-kernel VGPR/AGPR/scratch/occupancy/LDS table from hipcc -Rpass-analysis=kernel-resource-usage (development aid).
usage: kernel_resources.py sp-gan_amd/csrc/gemm.hip [filter]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + ROOT + "/include", "-I" + ROOT + "/sp-gan_amd/csrc",
                      "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1).split(" ")[0], m.group(2)
    if k == "Function":
        cur = {"name": subprocess.run(["/usr/bin/c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print("%-90s %5s %5s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["name"]); n = re.sub(r"\(.*$", "", n)
    if flt in n:
        print("%-90s %5s %5s %7s %4s %7s" % (n[:90], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
