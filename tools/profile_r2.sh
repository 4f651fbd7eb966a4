#!/bin/bash
# Round-2 profile run on the GPU box: inference bench line, kernel-trace stats of the same command, counter passes of
# the fused residual-pair kernel (separate --pmc runs).
set -o pipefail
tag=${1:-r2_a}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train > $out/trace.log 2>&1 || exit 2
python3 tools/rocpd_summary.py stats $(find $out/trace -name "*.db" | head -1) > $out/infer_kernel_stats.csv || exit 21
head -12 $out/infer_kernel_stats.csv
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $out/pmc1 -o p -- python3 tools/prof_pair.py > $out/pmc1.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA -d $out/pmc2 -o p -- python3 tools/prof_pair.py > $out/pmc2.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc3 -o p -- python3 tools/prof_pair.py > $out/pmc3.log 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/pmc4 -o p -- python3 tools/prof_pair.py > $out/pmc4.log 2>&1 || exit 6
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $out/pmc5 -o p -- python3 tools/prof_pair.py > $out/pmc5.log 2>&1 || exit 7
for i in 1 2 3 4 5; do python3 tools/rocpd_summary.py pmc $(find $out/pmc$i -name "*.db" | head -1) pair; done > $out/pair_pmc.csv
cat $out/pair_pmc.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_train -o bench -- python3 bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $out/trace_train.log 2>&1 || exit 8
python3 tools/rocpd_summary.py stats $(find $out/trace_train -name "*.db" | head -1) > $out/train_kernel_stats.csv || exit 81
head -30 $out/train_kernel_stats.csv
echo done
