#!/bin/bash
# Counter passes of the implicit-GEMM convolution on one layer shape: bash tools/profile_gemm.sh <tag> <case>
set -o pipefail
tag=${1:-r2_gemm}; cs=${2:-wide}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 100 python3 tools/prof_gemm.py wide k1 deep s2 mid > $out/times.log 2>&1 || exit 1
cat $out/times.log | grep -v amdgpu.ids
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $out/pmc1 -o p -- python3 tools/prof_gemm.py $cs > $out/pmc1.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA -d $out/pmc2 -o p -- python3 tools/prof_gemm.py $cs > $out/pmc2.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $out/pmc3 -o p -- python3 tools/prof_gemm.py $cs > $out/pmc3.log 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc4 -o p -- python3 tools/prof_gemm.py $cs > $out/pmc4.log 2>&1 || exit 6
for i in 1 2 3 4; do python3 tools/rocpd_summary.py pmc $(find $out/pmc$i -name "*.db" | head -1) conv_gemm; done > $out/gemm_pmc.csv
cut -d, -f2- $out/gemm_pmc.csv | sed 's/.*GemmParams)",//'
