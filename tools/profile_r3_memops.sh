set -o pipefail
out=gpurun_out/r3_zz; mkdir -p $out; export TMPDIR=/tmp
for c in loss upbwd; do
  i=0
  for CTRS in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $CTRS -d $out/${c}_$i -o p -- python3 tools/prof_new_kernels.py $c > $out/${c}_$i.log 2>&1 || { echo fail $c $i; tail -3 $out/${c}_$i.log; exit 3; }
  done
  pat=ssim; [ $c = upbwd ] && pat=upsample_bwd
  for i in 1 2 3; do python3 tools/rocpd_summary.py pmc $(find $out/${c}_$i -name "*.db" | head -1) $pat; done > $out/${c}_pmc.csv
  for i in 1 2 3; do rm -rf $out/${c}_$i; done
  cut -c1-60,100-400 $out/${c}_pmc.csv | grep -v "^kernel"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/tr -o b -- python3 tools/prof_new_kernels.py loss > $out/tr.log 2>&1 && python3 tools/rocpd_summary.py stats $(find $out/tr -name "*.db" | head -1) ssim | cut -c1-150
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/tr2 -o b -- python3 tools/prof_new_kernels.py upbwd > $out/tr2.log 2>&1 && python3 tools/rocpd_summary.py stats $(find $out/tr2 -name "*.db" | head -1) upsample | cut -c1-150
rm -rf $out/tr $out/tr2
