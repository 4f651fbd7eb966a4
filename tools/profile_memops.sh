#!/bin/bash
# Kernel-trace statistics and HBM counter passes of the memory-bound operators on the GPU box:
#   bash tools/profile_memops.sh <tag>        -> gpurun_out/<tag>/memops_kernel_stats.csv, memops_pmc.csv
set -o pipefail
tag=${1:-r2_mem}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/trace -o m -- python3 tools/bench_memops.py > $out/trace.log 2>&1 || exit 2
python3 tools/rocpd_summary.py stats $(find $out/trace -name "*.db" | head -1) > $out/memops_kernel_stats.csv || exit 21
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc1 -o p -- python3 tools/bench_memops.py > $out/pmc1.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/pmc2 -o p -- python3 tools/bench_memops.py > $out/pmc2.log 2>&1 || exit 4
for i in 1 2; do python3 tools/rocpd_summary.py pmc $(find $out/pmc$i -name "*.db" | head -1) ""; done > $out/memops_pmc.csv
cut -c1-150 $out/memops_kernel_stats.csv | head -30
cat $out/trace.log | tail -n 16
