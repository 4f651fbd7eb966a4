#!/bin/bash
# Kernel-trace statistics of the default inference bench line (rocprofv3 --kernel-trace --stats) into gpurun_out/<tag>/.
set -o pipefail
tag=${1:-r3_x}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-aux > $out/trace.log 2>&1 || exit 2
python3 tools/rocpd_summary.py stats $(find $out/trace -name "*.db" | head -1) > $out/infer_kernel_stats.csv || exit 21
head -14 $out/infer_kernel_stats.csv | cut -c1-160
rm -rf $out/trace
