#!/bin/bash
# Profile run on the GPU box: bench line, kernel-trace stats of the same command, counter passes.
set -o pipefail
tag=${1:-r1_b}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || exit 1
tail -1 $out/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/trace.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $out/pmc1 -o p -- python3 tools/prof_one.py > $out/pmc1.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA -d $out/pmc2 -o p -- python3 tools/prof_one.py > $out/pmc2.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc3 -o p -- python3 tools/prof_one.py > $out/pmc3.log 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/pmc4 -o p -- python3 tools/prof_one.py > $out/pmc4.log 2>&1 || exit 6
timeout -k 10 400 python -u bench.py --workload train --steps 10 --warmup 3 > $out/bench_train.json 2> $out/bench_train.err || exit 7
tail -1 $out/bench_train.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace_train -o train -- python3 bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > $out/trace_train.log 2>&1 || exit 8
echo done
