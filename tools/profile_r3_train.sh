#!/bin/bash
# Kernel-trace statistics of the training workload (rocprofv3 --kernel-trace --stats) into gpurun_out/<tag>/.
set -o pipefail
tag=${1:-r3_x}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_train -o bench -- python3 bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $out/trace_train.log 2>&1 || exit 2
python3 tools/rocpd_summary.py stats $(find $out/trace_train -name "*.db" | head -1) > $out/train_kernel_stats.csv || exit 21
head -24 $out/train_kernel_stats.csv | cut -c1-150
rm -rf $out/trace_train
