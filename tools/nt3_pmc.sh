#!/bin/bash
# SQ counters of the split-bf16 gemm_nt launch at D.fc2.0 (M = 65,536, N = 1024, K = 256: BatchNorm + LeakyReLU prologue, statistics +
# pooling epilogue, output not stored) and at conv_out (65536 x 128 x 1280, plain).  One rocprofv3 --pmc run per counter pair.
# usage (GPU box): bash tools/nt3_pmc.sh > gpurun_out/r06_nt3_pmc.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/nt3_drive.py <<PY
import sys, torch
sys.path.insert(0, "$R/sp-gan_amd")
from spgan import ops
dev = torch.device("cuda", 0); torch.manual_seed(0); M = 65536
ops.set_mfma_operands("bf16x3")
img = {}
def prov(W):
    k = (W.data_ptr(), tuple(W.shape))
    if k not in img: img[k] = ops.split_image(W)
    return img[k]
ops.w_image_provider = prov
A = torch.randn(M, 256, device=dev); W = torch.randn(1024, 256, device=dev) * 0.1; b = torch.randn(1024, device=dev)
sc = torch.rand(256, device=dev) + 0.5; sh = torch.randn(256, device=dev) * 0.3
gamma, beta = torch.rand(1024, device=dev) + 0.5, torch.randn(1024, device=dev)
A2 = torch.randn(M, 1280, device=dev); W2 = torch.randn(128, 1280, device=dev) * 0.1
for _ in range(6):
    ops.gemm_bn_pool(A, W, b, (gamma, beta, None, None), 2048, 0.2, pro=(sc, sh, 0.2))
    ops.gemm_nt(A2, W2, None)
torch.cuda.synchronize()
PY
echo "# gemm_nt_wide3 launches, M = 65536: <1,0,4> = D.fc2.0 (N 1024, K 256), <0,0,2> = conv_out (N 128, K 1280): average counter value per launch"
for grp in "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  d=/tmp/np_$(echo $grp | tr ' ' '_'); rm -rf $d
  rocprofv3 --pmc $grp --kernel-trace -d $d -o r -- python /tmp/nt3_drive.py > /tmp/np.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_summary.py $db gemm_nt_wide3 ; else echo "# $grp: no database ($(tail -1 /tmp/np.log))"; fi
done
