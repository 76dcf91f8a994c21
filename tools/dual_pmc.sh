#!/bin/bash
# SQ counters of the gemm_dual launches at the Discriminator's shapes (M = 65,536): fc2.0 collapsed (Na = 256, 256 inputs, activation operand),
# mlps.6 (256 -> 128, lazy operand), mlps.3 (128 -> 64).  One rocprofv3 --pmc run per counter pair (--kernel-trace only).
# usage (GPU box): bash tools/dual_pmc.sh > gpurun_out/r04_dual_pmc.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/dual_drive.py <<PY
import sys, torch
sys.path.insert(0, "$R/sp-gan_amd")
from spgan import ops
dev = torch.device("cuda", 0); torch.manual_seed(0); M = 65536
def run(Na, Nb, act):
    g, y = torch.randn(M, Na, device=dev), torch.randn(M, Na, device=dev)
    W = torch.randn(Na, Nb, device=dev) * 0.1
    sc, sh, mu, iv = (torch.randn(Nb, device=dev) for _ in range(4))
    yref = torch.randn(M, Nb, device=dev)
    if act:
        dy = ops.ActOperand(y, torch.rand(Na, device=dev) + 0.5, torch.randn(Na, device=dev) * 0.1, 0.01)
        yref = y
    else:
        dy = ops.Affine2(g, y, torch.randn(3, Na, device=dev))
    for _ in range(6):
        ops.gemm_dual(dy, W, yref, sc, sh, mu, iv, 0.01); ops.flush_tn()
run(256, 256, True); run(256, 128, False); run(128, 64, False)
torch.cuda.synchronize()
PY
echo "# gemm_dual launches, M = 65536: (Na,Nb) = (256,256) activation operand / (256,128) lazy / (128,64) lazy: average counter value per launch"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_SALU SQ_INST_CYCLES_SALU"; do
  d=/tmp/dp_$(echo $grp | tr ' ' '_'); rm -rf $d
  rocprofv3 --pmc $grp --kernel-trace -d $d -o r -- python /tmp/dual_drive.py > /tmp/dp.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_summary.py $db gemm_dual ; else echo "# $grp: no database ($(tail -1 /tmp/dp.log))"; fi
done
