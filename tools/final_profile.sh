#!/bin/bash
# Full profile set of the default network on the GPU box, summarised there (the rocpd databases stay behind):
# bench line, kernel-trace stats of the same command, traffic counter passes of the dominant kernel, training bench + stats.
set -o pipefail
tag=${1:-r1_g}
out=gpurun_out/$tag
tmp=/tmp/srhip_prof_$tag
mkdir -p $out $tmp
export TMPDIR=/tmp
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || exit 1
tail -1 $out/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $tmp/trace -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/trace.log 2>&1 || exit 2
python tools/rocpd_summary.py stats $(ls $tmp/trace/*/*.db $tmp/trace/*.db 2>/dev/null | head -1) > $out/bench_kernel_stats.csv || exit 3
echo trace-done
: > $out/conv64_pmc.csv
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c -d $tmp/pmc$i -o p -- python3 tools/prof_one.py > $out/pmc$i.log 2>&1 || exit 4
  python tools/rocpd_summary.py pmc $(ls $tmp/pmc$i/*/*.db $tmp/pmc$i/*.db 2>/dev/null | head -1) "Cfg<4, 2, 1, false, 0>" >> $out/conv64_pmc.csv || exit 5
  echo pmc-pass-$i-done
done
timeout -k 10 400 python -u bench.py --workload train --steps 10 --warmup 3 > $out/bench_train.json 2> $out/bench_train.err || exit 7
tail -1 $out/bench_train.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $tmp/trace_train -o train -- python3 bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > $out/trace_train.log 2>&1 || exit 8
python tools/rocpd_summary.py stats $(ls $tmp/trace_train/*/*.db $tmp/trace_train/*.db 2>/dev/null | head -1) > $out/train_kernel_stats.csv || exit 9
rm -rf $tmp
echo done
