#!/bin/bash
# Counter passes (each in a run of its own) of the head form of the pair kernel, the reduction-split concat layer and
# the implicit-GEMM convolution on a temporal-network layer shape: bash tools/profile_r3_pmc.sh <tag>
set -o pipefail
tag=${1:-r3_pmc}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
pass() {  # name script args... ; counters in $CTRS
    local name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc $CTRS -d $out/$name -o p -- python3 "$@" > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 3; }
}
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    pass head$i tools/prof_head.py
    pass concat$i tools/prof_concat.py
    pass gemm$i tools/prof_gemm.py wide
    pass gemmk1$i tools/prof_gemm.py k1
done
for i in 1 2 3 4; do python3 tools/rocpd_summary.py pmc $(find $out/head$i -name "*.db" | head -1) pair; done > $out/head_pmc.csv
for i in 1 2 3 4; do python3 tools/rocpd_summary.py pmc $(find $out/concat$i -name "*.db" | head -1) ksplit; done > $out/concat_pmc.csv
for i in 1 2 3 4; do python3 tools/rocpd_summary.py pmc $(find $out/gemm$i -name "*.db" | head -1) conv_gemm; done > $out/gemm_wide_pmc.csv
for i in 1 2 3 4; do python3 tools/rocpd_summary.py pmc $(find $out/gemmk1$i -name "*.db" | head -1) conv_gemm; done > $out/gemm_k1_pmc.csv
rm -rf $out/head? $out/concat? $out/gemm? $out/gemmk1?
for f in head concat gemm_wide gemm_k1; do echo "== $f"; cut -d, -f2- $out/${f}_pmc.csv | sed 's/.*Params)",//' | grep -v "^counter"; done
