#!/bin/bash
# Counter passes (each in a run of its own) of the fused residual-pair kernel at 2160x3840, inference form and the
# training form that also stores the intermediate map: bash tools/profile_r3_pair_pmc.sh <tag>
set -o pipefail
tag=${1:-r3_pair}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
pass() {  # name args... ; counters in $CTRS
    local name=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc $CTRS -d $out/$name -o p -- python3 tools/prof_pair.py "$@" > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 3; }
}
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    pass infer$i
    pass mid$i mid
done
for i in 1 2 3 4; do python3 tools/rocpd_summary.py pmc $(find $out/infer$i -name "*.db" | head -1) pair; done > $out/pair_pmc.csv
for i in 1 2 3 4; do python3 tools/rocpd_summary.py pmc $(find $out/mid$i -name "*.db" | head -1) pair; done > $out/pair_mid_pmc.csv
rm -rf $out/infer? $out/mid?
for f in pair pair_mid; do echo "== $f"; cut -d, -f2- $out/${f}_pmc.csv | sed 's/.*Params)",//' | grep -v "^counter"; done
