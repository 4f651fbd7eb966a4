#!/bin/bash
# Kernel-trace statistics of one of the auxiliary workloads on the GPU box: bash tools/profile_aux.sh <tag> <workload>
set -o pipefail
tag=${1:-r2_aux}; wl=${2:-temporal}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_$wl -o bench -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $out/trace_$wl.log 2>&1 || exit 8
python3 tools/rocpd_summary.py stats $(find $out/trace_$wl -name "*.db" | head -1) > $out/${wl}_kernel_stats.csv || exit 81
cut -c1-160 $out/${wl}_kernel_stats.csv | head -${3:-24}
