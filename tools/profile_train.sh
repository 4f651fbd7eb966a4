#!/bin/bash
# Kernel-trace statistics of the training step on the GPU box: bash tools/profile_train.sh <tag>
set -o pipefail
tag=${1:-r2_train}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_train -o bench -- python3 bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $out/trace_train.log 2>&1 || exit 8
python3 tools/rocpd_summary.py stats $(find $out/trace_train -name "*.db" | head -1) > $out/train_kernel_stats.csv || exit 81
cut -c1-200 $out/train_kernel_stats.csv | head -${2:-16}
