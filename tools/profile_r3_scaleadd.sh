# Counter passes (FETCH_SIZE, WRITE_SIZE, an SQ set; one run each) and a kernel trace of the one-pass scaled sum of the
# temporal training graph and its backward: bash tools/profile_r3_scaleadd.sh [tag]
set -o pipefail
out=gpurun_out/${1:-r3_bl}; mkdir -p $out; export TMPDIR=/tmp
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $CTRS -d $out/sa_$i -o p -- python3 tools/prof_new_kernels.py scaleadd > $out/sa_$i.log 2>&1 || { echo fail $i; tail -3 $out/sa_$i.log; exit 3; }
done
for i in 1 2 3; do python3 tools/rocpd_summary.py pmc $(find $out/sa_$i -name "*.db" | head -1) scale_add; done > $out/scaleadd_pmc.csv
for i in 1 2 3; do rm -rf $out/sa_$i; done
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $out/sa_trace -o t -- python3 tools/prof_new_kernels.py scaleadd > $out/sa_trace.log 2>&1 || { echo fail trace; exit 4; }
python3 tools/rocpd_summary.py stats $(find $out/sa_trace -name "*.db" | head -1) > $out/scaleadd_kernel_stats.csv 2>/dev/null || find $out/sa_trace -name "*kernel_stats.csv" -exec cp {} $out/scaleadd_kernel_stats.csv \;
rm -rf $out/sa_trace
cut -c1-60,100-400 $out/scaleadd_pmc.csv | grep -v "^kernel"; grep scale_add $out/scaleadd_kernel_stats.csv | cut -c1-200
