#!/bin/bash
# Final evidence of the round on one GPU box: bench lines, kernel traces of every workload, counter passes of the new
# kernels.  bash tools/profile_r3_final.sh <tag> [part]   (part: bench | traces | pmc; default all)
set -o pipefail
tag=${1:-r3_z}; part=${2:-all}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ $part = all ] || [ $part = bench ]; then
    timeout -k 10 900 python3 bench.py > $out/bench.json 2> $out/bench.err || { echo bench failed; tail -5 $out/bench.err; exit 2; }
    cut -c1-300 $out/bench.json
fi
if [ $part = all ] || [ $part = traces ]; then
    bash tools/profile_r3_infer.sh $tag > $out/infer_head.txt || exit 3
    bash tools/profile_r3_train.sh $tag > $out/train_head.txt || exit 4
    bash tools/profile_aux.sh $tag temporal 12 > $out/temporal_head.txt || exit 5
    bash tools/profile_aux.sh $tag reprfn 12 > $out/reprfn_head.txt || exit 6
    rm -rf $out/trace_temporal $out/trace_reprfn
fi
if [ $part = all ] || [ $part = pmc ]; then
    pass() {  # name case ; counters in $CTRS
        timeout -s KILL 120 rocprofv3 --pmc $CTRS -d $out/$1 -o p -- python3 tools/prof_new_kernels.py $2 > $out/$1.log 2>&1 || { echo "pass $1 failed"; tail -3 $out/$1.log; exit 7; }
    }
    i=0
    for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
        i=$((i + 1))
        for c in chain chain_res esa pairbwd pairbits; do pass ${c}_$i $c; done
    done
    for c in chain chain_res esa pairbwd pairbits; do
        pat=pair; [ $c = chain ] && pat=chain; [ $c = chain_res ] && pat=chain; [ $c = esa ] && pat=esa_tail
        for i in 1 2 3 4; do python3 tools/rocpd_summary.py pmc $(find $out/${c}_$i -name "*.db" | head -1) $pat; done > $out/${c}_pmc.csv
        echo "== $c"; cut -d, -f2- $out/${c}_pmc.csv | sed 's/.*Params)",//' | grep -v "^counter"
        for i in 1 2 3 4; do rm -rf $out/${c}_$i; done
    done
fi
