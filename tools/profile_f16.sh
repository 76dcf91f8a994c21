#!/bin/bash
# Kernel table of the "f16" operand mode (BASELINE configs[4]) on the GPU box -> gpurun_out/<tag>_bench_f16_kernel_stats.txt
# usage (through gpurun): bash tools/profile_f16.sh r03
set -u
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_h; rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o r -- python $R/bench.py --mfma f16 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > /tmp/bench_h.log 2>&1
tail -1 /tmp/bench_h.log | cut -c1-240
DB=$(find /tmp/prof_h -name "*.db" | head -1)
[ -n "$DB" ] || { echo "no rocprofv3 database"; exit 1; }
python $R/tools/rocprof_summary.py $DB 31 8 > $O/${TAG}_bench_f16_kernel_stats.txt      # 31 steps: 4 priming + 3 warm-up + 20 timed + 4 eager accounting steps (+ the keep-busy launches: the Cijk_ row)
head -50 $O/${TAG}_bench_f16_kernel_stats.txt | cut -c1-200
