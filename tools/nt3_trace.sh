#!/bin/bash
# Kernel-only durations (rocprofv3 --kernel-trace --stats) of the split-bf16 256-row-tile gemm_nt at the step's shapes, next to the fp32 route.
# usage (GPU box): bash tools/nt3_trace.sh > gpurun_out/r06_nt3_trace.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/nt3_tr.py <<PY
import sys, torch
sys.path.insert(0, "$R/sp-gan_amd")
from spgan import ops
dev = torch.device("cuda", 0); torch.manual_seed(0)
img = {}
def prov(W):
    k = (W.data_ptr(), tuple(W.shape))
    if k not in img: img[k] = ops.split_image(W)
    return img[k]
big = torch.empty(64 * 1024 * 1024, device=dev); big2 = torch.empty_like(big)
for mode, pr in (("f32", None), ("bf16x3", None), ("bf16x3", prov)):
    ops.set_mfma_operands(mode); ops.w_image_provider = pr
    for (M, N, K, pool) in ((65536, 1024, 256, 1), (196608, 1024, 256, 1), (65536, 128, 1280, 0), (65536, 1280, 128, 0), (131072, 256, 128, 0), (65536, 256, 256, 0)):
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
        sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.3
        gamma, beta = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
        for _ in range(8):
            big2.copy_(big)
            if pool: ops.gemm_bn_pool(A, W, b, (gamma, beta, None, None), 2048, 0.2, pro=(sc, sh, 0.2))
            else: ops.gemm_nt(A, W, b)
        torch.cuda.synchronize()
PY
rm -rf /tmp/nt3_tr; rocprofv3 --kernel-trace --stats -d /tmp/nt3_tr -o r -- python /tmp/nt3_tr.py > /tmp/nt3_tr.log 2>&1
DB=$(find /tmp/nt3_tr -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
print("# kernel-only durations, 8 launches each behind a 256 MB copy; order of shapes per mode: see tools/nt3_trace.sh")
for n, gx, wg, cnt, av, mn in c.execute("select name, grid_x, workgroup_x, count(*), avg(duration), min(duration) from kernels where name like '%gemm_nt%' group by name, grid_x order by name, grid_x"):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*$", "", n)
    print("%-60s grid %8d x %3d  calls %3d  avg %8.1f us  min %8.1f us" % (n[:60], gx // max(wg, 1), wg, cnt, av / 1e3, mn / 1e3))
PY
