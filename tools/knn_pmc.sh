#!/bin/bash
# PMC passes over the feature-space kNN launches at the train step's shape (B=32, N=2048, C=64, k=10): the single-launch kernel of
# csrc/graph.hip and the tile-image + pipelined pair of csrc/knn_pipe.hip.  One rocprofv3 run per counter group (--pmc with
# --kernel-trace only).  usage (GPU box): bash tools/knn_pmc.sh > gpurun_out/r04_knn_pmc.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/knn_drive.py <<PY
import sys, torch
sys.path.insert(0, "$R/sp-gan_amd")
from spgan import ops, fixture_rng as fr
B, N, k = 32, 2048, 10
x = fr.normal("knn.ab", (B * N, 64), 0.5).cuda()
for pipe in (False, True):
    ops.KNN_PIPELINED[0] = pipe
    for _ in range(6):
        ops.knn(x, B, N, k, 0)
torch.cuda.synchronize()
PY
echo "# kNN launches, B=32 N=2048 C=64 k=10: average counter value per launch (rocprofv3 --pmc, one pass per group)"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  d=/tmp/kp_$(echo $grp | tr ' ' '_'); rm -rf $d
  rocprofv3 --pmc $grp --kernel-trace -d $d -o r -- python /tmp/knn_drive.py > /tmp/kp.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_summary.py $db knn_ ; else echo "# $grp: no database ($(tail -1 /tmp/kp.log))"; fi
done
