#!/bin/bash
# Collects the measurement artefacts of a round on the GPU box into gpurun_out/ (copy what shall be judged to profiles/).
# usage (through gpurun): bash tools/collect_profiles.sh r03
set -u
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python $R/bench.py --config c5 --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_f16_operands.json 2>> $O/${TAG}_bench.err      # c5 = --mfma f16
python $R/bench.py --config c4 --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_c4.json 2>> $O/${TAG}_bench.err                # N = 4096, per-GPU batch 16
python $R/bench.py --mfma bf16x3 --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_bf16x3.json 2>> $O/${TAG}_bench.err
rm -rf /tmp/prof_s; rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o r -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > /tmp/bench_s.log 2>&1
DB=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB 31 > $O/${TAG}_bench_kernel_stats.txt
python $R/tools/roofline_same_process.py /tmp/bench_s.log $DB > $O/${TAG}_roofline_same_process.json 2>> $O/${TAG}_bench.err      # stamps / events / rocprof of ONE process      # 31 steps: 4 priming + 3 warm-up + 20 timed + 4 eager accounting steps (+ the keep-busy launches: the Cijk_ row)
python $R/tools/gap_analysis.py $DB 0.35 0.7 > $O/${TAG}_bench_graph_replay_window.txt
python $R/tools/step_sequence.py $DB -8 > $O/${TAG}_step_sequence.txt
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do rm -rf /tmp/p_$c; rocprofv3 --pmc $c --kernel-trace -d /tmp/p_$c -o r -- python $R/tools/pmc_step.py > /tmp/log_$c 2>&1; done
python $R/tools/pmc_step_total.py $(find /tmp/p_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/p_WRITE_SIZE -name "*.db" | head -1) $(find /tmp/p_MfmaUtil -name "*.db" | head -1) > $O/${TAG}_pmc_step.json
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do rm -rf /tmp/p4_$c; rocprofv3 --pmc $c --kernel-trace -d /tmp/p4_$c -o r -- python $R/tools/pmc_step.py --config c4 > /tmp/log4_$c 2>&1; done     # the same at the C4 per-GPU shape
python $R/tools/pmc_step_total.py $(find /tmp/p4_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/p4_WRITE_SIZE -name "*.db" | head -1) $(find /tmp/p4_MfmaUtil -name "*.db" | head -1) > $O/${TAG}_pmc_step_c4.json
for m in f16 bf16x3; do      # the same for the two 16-bit operand modes at C2
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do rm -rf /tmp/pm_$c; rocprofv3 --pmc $c --kernel-trace -d /tmp/pm_$c -o r -- python $R/tools/pmc_step.py --mfma $m > /tmp/logm_$c 2>&1; done
python $R/tools/pmc_step_total.py $(find /tmp/pm_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*.db" | head -1) $(find /tmp/pm_MfmaUtil -name "*.db" | head -1) > $O/${TAG}_pmc_step_$m.json
done
for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/g_$c; rocprofv3 --pmc $c --kernel-trace -d /tmp/g_$c -o r -- python $R/tools/pmc_gemm.py > /tmp/glog_$c 2>&1; python $R/tools/pmc_summary.py $(find /tmp/g_$c -name "*.db" | head -1) gemm_nt > $O/${TAG}_pmc_gemm_$c.txt; done
python $R/tools/pmc_gemm_json.py $O/${TAG}_pmc_gemm_FETCH_SIZE.txt $O/${TAG}_pmc_gemm_WRITE_SIZE.txt ${TAG} > $O/${TAG}_pmc_gemm_nt.json
python $R/tools/mfma_shapes.py > $O/${TAG}_mfma_shapes.txt 2>/dev/null
python $R/tools/mfma_shapes.py --mfma f16 > $O/${TAG}_mfma_shapes_f16.txt 2>/dev/null
python $R/tools/mfma_shapes.py --mfma bf16x3 > $O/${TAG}_mfma_shapes_bf16x3.txt 2>/dev/null
# split-bf16 gemm_nt on 256-row tiles (csrc/gemm_wide3.hip): microbench against the other routes, K sweep (slope / intercept), kernel-only
# durations, SQ counters, operand-data (power) probe, accuracy / bias probe, arg-max flip probe
(cd /tmp && python $R/tools/nt3_bench.py) > $O/${TAG}_nt3_bench.txt 2>/dev/null
python $R/tools/nt3_ksweep.py > $O/${TAG}_nt3_ksweep.txt 2>/dev/null
bash $R/tools/nt3_trace.sh > $O/${TAG}_nt3_trace.txt 2>/dev/null
python $R/tools/tn3_bench.py > $O/${TAG}_tn3_bench.txt 2>/dev/null                   # split-bf16 weight gradients (csrc/gemm_tn_wide3.hip) against the fp32 kernel
(cd /tmp && python $R/tools/nt16_bench.py) > $O/${TAG}_nt16_bench.txt 2>/dev/null     # fp16 operands: row-pipelined 256 x 256 tiles (csrc/gemm_wide16.hip) against gemm_wide.hip's
python $R/bench.py --mfma bf16x3 --config c4 --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_c4_bf16x3.json 2>/dev/null      # the C4 per-GPU shape in the split mode
SPGAN_TN_SPLIT=0 python $R/bench.py --mfma bf16x3 --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_bf16x3_tn_f32.json 2>/dev/null      # A/B: weight gradients on the fp32 kernel
SPGAN_NT_WIDE16=0 python $R/bench.py --config c5 --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_f16_operands_old_wide.json 2>/dev/null   # A/B: gemm_wide.hip's fp16 form
bash $R/tools/nt3_pmc.sh > $O/${TAG}_nt3_pmc.txt 2>/dev/null
python $R/tools/exp/nt3_power_probe.py > $O/${TAG}_nt3_power_probe.txt 2>/dev/null
python $R/tools/exp/nt3_accuracy_probe.py > $O/${TAG}_nt3_accuracy_probe.txt 2>/dev/null
python $R/tools/exp/argmax_flip_probe.py > $O/${TAG}_argmax_flip_probe.txt 2>/dev/null
rm -rf /tmp/prof_b3; rocprofv3 --kernel-trace --stats -d /tmp/prof_b3 -o r -- python $R/bench.py --mfma bf16x3 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > /tmp/bench_b3.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/prof_b3 -name "*.db" | head -1) 31 > $O/${TAG}_bench_bf16x3_kernel_stats.txt
python $R/tools/step_sequence.py $(find /tmp/prof_b3 -name "*.db" | head -1) -8 > $O/${TAG}_step_sequence_bf16x3.txt
(python $R/tools/knn_ab.py 2048 32; python $R/tools/knn_ab.py 4096 16) > $O/${TAG}_knn_ab.txt 2>&1
python $R/tools/dual_bench.py > $O/${TAG}_dual_bench.txt 2>/dev/null
rm -rf /tmp/prof_c4; rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o r -- python $R/bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > /tmp/bench_c4.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/prof_c4 -name "*.db" | head -1) 31 > $O/${TAG}_bench_c4_kernel_stats.txt
bash $R/tools/profile_f16.sh $TAG > /dev/null 2>&1      # kernel table of the "f16" operand mode
python $R/tools/step_sequence.py $(find /tmp/prof_h -name "*.db" | head -1) -8 > $O/${TAG}_step_sequence_f16.txt      # ... and its ordered step sequence
python $R/tools/exp/wide_k_sweep.py > $O/${TAG}_wide_k_sweep.txt 2>/dev/null
python $R/tools/exp/mid_k_sweep.py > $O/${TAG}_mid_k_sweep.txt 2>/dev/null
bash $R/tools/knn_pmc.sh > $O/${TAG}_knn_pmc.txt 2>&1                      # PMC passes over the kNN launches (both routes)
python $R/tools/torch_ops_in_step.py > $O/${TAG}_torch_ops_in_step.txt 2>/dev/null
echo collected; ls -la $O | tail -20
